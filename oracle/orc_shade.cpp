/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 * CPU restatement of the reference's light transport: RayTracer::VisualRay / LightRay and their
 * helpers (Shade.cpp:62-520), cameras (Cameras.h:8-67), light sample points (Lights.h:20-30,
 * Lights.cpp:33-37), propagators (TextureMaps.cpp:13-57), spectrum -> RGB (RadiometryColor.h:62-66).
 *
 * Stated policies where the reference's behaviour is undefined or order-dependent (DESIGN.md "Quirks"):
 *  Q2  MAT_NOSHADE early return (Shade.cpp:436-437) does not leak the depth counter.
 *  Q3  spectral metals (MAT_COPPER..MAT_TITANIUM): fFresnel is never assigned in the reference
 *      (Shade.cpp:446-466) but read at :491/:494; here it is 1.0 (any non-zero value gives the same
 *      radiance), and the per-hit Fresnel colour is kept per hit instead of being written into the
 *      shared Material (Shade.cpp:447).
 *  Q11 a hit with no surface (ps == 0) that is reached inside the walk is shaded as matte instead of
 *      dereferencing null (Shade.cpp:471-472, :331).
 */
#include "orc_shade.h"

namespace orc {

Tables g_t;

Rgb spectrum_srgb(const Spec &s)
{
    Rgb c = { 0.0f, 0.0f, 0.0f };
    for (int i = 0; i < NSPEC; i++) {
        double w = s.v[i] * (5.0 / 10.0);
        c.r = c.r + float(g_t.spec2rgb[i][0] * w);
        c.g = c.g + float(g_t.spec2rgb[i][1] * w);
        c.b = c.b + float(g_t.spec2rgb[i][2] * w);
    }
    return c;
}

Rgb spectrum_vrgb(const Spec &s)      /* RadiometryColor.cpp:971-983 */
{
    Rgb c = { 0.0f, 0.0f, 0.0f };
    for (int i = 0; i < NSPEC; i++) {
        c.r += s.v[i] * g_t.chan[0].v[i];
        c.g += s.v[i] * g_t.chan[1].v[i];
        c.b += s.v[i] * g_t.chan[2].v[i];
    }
    return c;
}

Ray camera_emit(const RtCamera &c, double vx, double vy, const Ctx &cx)
{
    Ray r;
    V3 org(c.org[0], c.org[1], c.org[2]);
    double x, y, d;
    switch (c.type) {
    case RT_CAM_PERSPECTIVE:                                    /* Cameras.h:11 */
        r.org = org;
        r.dir = V3((vx - 0.5) * 35.0 / c.focal, (vy - 0.5) * 35.0 / c.focal, -1.0);
        return r;
    case RT_CAM_ORTHOGRAPHIC:                                   /* Cameras.h:29-34 */
        x = (vx - 0.5) * 35.0 / c.focal;
        y = (vy - 0.5) * 35.0 / c.focal;
        r.org = V3(x, y, org.z);
        r.dir = V3(0.0, 0.0, -1.0);
        return r;
    default:
        break;
    }
    x = (vx - 0.5) * 35.0 / c.focal;                            /* Cameras.h:17-23 */
    y = (vy - 0.5) * 35.0 / c.focal;
    d = x * x + y * y + 4.0;
    r.org = org;
    r.dir = V3(4.0 * x / d, 4.0 * y / d, 1.0 - 8.0 / d);
    if (c.type == RT_CAM_APERTURE) {                            /* Cameras.h:57-66 */
        Ray r2;
        r.dir = normalize(r.dir);
        V3 vX(c.vX[0], c.vX[1], c.vX[2]), vY(c.vY[0], c.vY[1], c.vY[2]);
        r2.org = r.org + vX * cx.aux[0] + vY * cx.aux[1];
        r2.dir = at(r, c.focus_distance) - r2.org;
        return r2;
    }
    return r;
}

/* Shade.cpp:62-86 */
static void consistent_normal(const Ray &r, const Hit &h, V3 &vE, V3 &cvN, double &fNE)
{
    vE = -normalize(r.dir);
    if (h.perturbed) {
        cvN = normalize(h.np);
        fNE = dot(cvN, vE);
        double fNE2 = dot(h.n, vE);
        if (fNE * fNE2 < 0.0) {
            cvN = normalize(h.n);
            fNE = dot(cvN, vE);
        }
    } else {
        cvN = normalize(h.n);
        fNE = dot(cvN, vE);
    }
    if (fNE < 0.0) {
        cvN = -cvN;
        fNE = -fNE;
    }
}

/* Shade.cpp:90-113 */
static void fresnel_real(double fNE, double n1, double n2, double &F, double &fNT)
{
    double eta = n1 / n2;
    double crit = 1.0 - 1.0 / (eta * eta);
    if (crit < 0.0) crit = 0.0; else crit = sqrt(crit);
    if (fNE < crit) {
        F = 1.0;
    } else {
        fNT = sqrt(1.0 - eta * eta * (1.0 - fNE * fNE));
        if (n1 == n2) {
            F = 0.0;
        } else {
            double Rs = (n1 * fNE - n2 * fNT) / (n1 * fNE + n2 * fNT);
            double Rp = (n1 * fNT - n2 * fNE) / (n1 * fNT + n2 * fNE);
            F = 0.5 * (Rs * Rs + Rp * Rp);
        }
    }
}

/* Shade.cpp:117-139 */
static void fresnel_complex(double fNE, Cx n1, Cx n2, double &F, double &fNT)
{
    Cx eta = n1 / n2;
    double crit = 1.0 - 1.0 / eta.abs2();
    if (crit < 0.0) crit = 0.0; else crit = sqrt(crit);
    if (fNE < crit) {
        F = 1.0;
    } else {
        fNT = sqrt(1.0 - eta.abs2() * (1.0 - fNE * fNE));
        Cx Rs = (n1 * fNE - n2 * fNT) / (n1 * fNE + n2 * fNT);
        Cx Rp = (n1 * fNT - n2 * fNE) / (n1 * fNT + n2 * fNE);
        F = 0.5 * (Rs.abs2() + Rp.abs2());
        if (n1.re == n2.re && n1.im == n2.im) F = 0.0;
    }
}

/* Shade.cpp:141-153 */
static Spec fresnel_spectral(double fNE, double n1, const Spec &nn, const Spec &kk)
{
    Spec s;
    double F = 0.0, fNT = 0.0;
    for (int i = 0; i < NSPEC; i++) {
        fresnel_complex(fNE, Cx(n1, 0.0), Cx(nn.v[i], kk.v[i]), F, fNT);
        s.v[i] = float(F);
    }
    return s;
}

/* the three Fresnel routines on their own (TestFresnel, Test.cpp:1467-1515): FresnelLaw_Real for (n1, n2); Fresnel_Spectral of a
   spectral metal seen from a medium of index n1 when metal_luster names one (spec10), else FresnelLaw_Complex with real indices */
void fresnel_probe(double fNE, double n1, double n2, int metal_luster, double *F, double *fNT, float *spec10)
{
    *F = 0.0; *fNT = 0.0;
    fresnel_real(fNE, n1, n2, *F, *fNT);
    if (!spec10) return;
    if (metal_luster >= RT_MAT_COPPER && metal_luster <= RT_MAT_TITANIUM) {
        Spec s = fresnel_spectral(fNE, n1, g_t.metal_n[metal_luster], g_t.metal_k[metal_luster]);
        memcpy(spec10, s.v, sizeof(float) * NSPEC);
    } else {
        double Fc = 0.0, fNTc = 0.0;
        fresnel_complex(fNE, Cx(n1, 0.0), Cx(n2, 0.0), Fc, fNTc);
        for (int i = 0; i < NSPEC; i++) spec10[i] = float(Fc);
    }
}

static inline bool spectral_metal(int luster) { return luster >= RT_MAT_COPPER && luster <= RT_MAT_TITANIUM; }
static inline int roughness(const Hit &h) { return h.surf ? h.surf->i0 : RT_SURF_MATTE; }   /* Q11 */

/* Shade.cpp:155-234 */
static Spec brdf(const Hit &h, const Spec &color, const V3 &vL, const V3 &vE, const V3 &cvN, double F)
{
    const Spec white = spec_const(1.0f), black = spec_const(0.0f);
    double lambert = 1.0 / kPi, kS = 1.0, fBRDF;
    int logk = 0;
    V3 nspec = cvN;
    switch (roughness(h)) {
    case RT_SURF_MATTE:     return color * lambert;
    case RT_SURF_BURNISHED: logk = 3; break;
    case RT_SURF_SMOOTH:    logk = 5; kS = 1.0; break;
    case RT_SURF_WET:       nspec = normalize(h.n);     /* falls through */
    case RT_SURF_SHINY:     logk = 7; kS = 0.60; break;
    case RT_SURF_POLISHED:  logk = 9; kS = 0.35; break;
    }
    if (logk) {
        double fNL = dot(cvN, vL), fNLs = dot(nspec, vL), fNE = dot(nspec, vE);
        V3 vH = normalize(vE + vL);
        double fNH = dot(nspec, vH);
        int kappa = 1;
        double blinn = fNH;
        while (logk > 0) { blinn = blinn * blinn; kappa = kappa + kappa; logk--; }
        blinn = kS * (double(kappa) + 2.0) * blinn / (2.0 * kPi);
        double fEH = dot(vE, vH);
        double f1 = fabs(2.0 * fNH * fNE / fEH), f2 = fabs(2.0 * fNH * fNLs / fEH);
        double geo = f1 < f2 ? f1 : f2;         /* Min(1.0, Min(f1, f2)), Affine.h:94-101 */
        geo = 1.0 < geo ? 1.0 : geo;
        fBRDF = blinn * geo / (4.0 * fNL * fNE);
    } else
        fBRDF = 0.0;
    int luster = h.mat->i0;
    switch (luster) {
    case RT_MAT_PLASTIC:     return color * lambert + white * (F * fBRDF);
    case RT_MAT_METAL:       return color * ((1.0 - F) * fBRDF) + white * (F * fBRDF);
    case RT_MAT_TRANSPARENT: return white * (F * fBRDF);
    default:
        if (spectral_metal(luster)) return color * fBRDF;
    }
    return black;
}

/* Shade.cpp:236-256 */
static Spec ray_reflectance(const Hit &h, const Spec &color, double F)
{
    const Spec white = spec_const(1.0f);
    int luster = h.mat->i0;
    switch (luster) {
    case RT_MAT_PLASTIC:     return white * F;
    case RT_MAT_METAL:       return color * (1.0 - F) + white * F;
    case RT_MAT_TRANSPARENT: return white * F;
    default:
        if (spectral_metal(luster)) return color;
    }
    return spec_const(0.0f);
}

/* Shade.cpp:264-284 */
static void transform_normals(Hit &h, const Ctx &cx)
{
    size_t it = 0;
    Node *pa;
    for (pa = h.affine; pa; pa = pa->link) {
        while (it < h.tex.size() && h.tex[it].pa == pa) {
            apply_texture(h.tex[it].tex, h.tex[it].p, h, cx, h.tex[it].images);
            it++;
        }
        h.n = mul(h.n, pa->mi);
        if (h.perturbed)
            h.np = mul(h.np, pa->mi);
    }
    while (it < h.tex.size() && h.tex[it].pa == pa) {       /* pa == 0 here */
        apply_texture(h.tex[it].tex, h.tex[it].p, h, cx, h.tex[it].images);
        it++;
    }
}

void transform_normals_export(Hit &h, const Ctx &cx) { transform_normals(h, cx); }

struct Prop { Spec tr, ster; };

/* TextureMaps.cpp:13-57 */
static Prop propagate(const Scene &s, int id, const V3 &p0, const V3 &p1, const Node *pm)
{
    Prop p;
    if (id == RT_PROP_ATMOSPHERE) {
        /* NEW BEHAVIOUR (the reference's Atmosphere is an empty stub, TextureMaps.cpp:162-168): Rayleigh extinction with the
           constants of Rayleigh() (:123-135) -- K = 1e36 * 1e3 * 24 pi^3 * ((n^2-1)/(n^2+2))^2 / N over the wavelength in nm to
           the fourth, per kilometre; N from n = sqrt(1 + alpha N) (:117-121) with POLARIZABILITY_AIR (:107); the material's
           colour scales the coefficient per band; scene units are metres -- and airlight (1 - T) times the ambient radiance. */
        const double n2 = pm->refr * pm->refr;
        double K = 0.0;
        if (n2 > 1.0) {
            const double ll = (n2 - 1.0) / (n2 + 2.0), N = (n2 - 1.0) / 2.133e-29;
            K = 1.0e36 * 1.0e3 * (24.0 * kPi * kPi * kPi) * (ll * ll) / N;
        }
        const double km = norm(p1 - p0) / 1000.0;
        for (int i = 0; i < NSPEC; i++) {
            const double lam = 380.0 + 40.0 * i, l2 = lam * lam;
            double x = (double)pm->color.v[i] * K / (l2 * l2) * km, T;
            if (x > 87.0) T = 0.0; else T = exp(-x);
            p.tr.v[i] = float(T);
            float a = float(1.0 - T);
            p.ster.v[i] = s.has_ambient ? float(double(a) * double(s.ambient.v[i])) : 0.0f;
        }
        return p;
    }
    if (id == RT_PROP_FOGGY) {
        double t = norm(p1 - p0);
        for (int i = 0; i < NSPEC; i++) {
            double sc = pm->color.v[i] * t, T;
            if (sc > 87.0) T = 0.0; else T = exp(-sc);
            p.tr.v[i] = float(T);
            p.ster.v[i] = float((1.0 - T) * 0.1);
        }
        return p;
    }
    p.ster = spec_const(0.0f);
    double t = norm(p1 - p0) * 100.0;
    for (int i = 0; i < NSPEC; i++) {
        double mu = pm->color.v[i] * t, T;
        if (mu > 87.0) T = 0.0; else T = exp(-mu);
        p.tr.v[i] = float(T);
    }
    return p;
}

static inline V3 light_origin(const Light &l, const Ctx &cx)      /* Lights.h:29, RayTracer2020.h:162 */
{
    if (l.type == RT_LIGHT_AREA)
        return l.org + l.vA * cx.aux[2] + l.vB * cx.aux[3];
    return l.org;
}

/* Shade.cpp:301-359 */
Spec light_ray(Scene &s, const Ray &ray, const Light &l, Ctx &cx)
{
    const Spec black = spec_const(0.0f);
    double r2 = dot(ray.dir, ray.dir);
    Spec E = l.flux / (4.0 * kPi * r2);
    if (l.no_shadow)
        return E;
    HitList hl;
    cx.n_shadow++;
    node_hits(s.root, ray, cx, hl);
    size_t i;
    Node *pm;
    if (hl.empty() || hl[0].t > 0.0) { pm = &g_b.vacuum; i = 0; } else { pm = hl[0].mat; i = 1; }
    double t = 0.0;
    Spec tr = spec_const(1.0f);
    for (; i < hl.size() && hl[i].t < 1.0 - kEps; i++) {
        Hit &h = hl[i];
        transform_normals(h, cx);
        if (pm->i0 != RT_MAT_TRANSPARENT || h.mat->i0 != RT_MAT_TRANSPARENT || roughness(h) != RT_SURF_POLISHED)
            return black;
        if (pm->i2) {
            Prop p = propagate(s, pm->i2, at(ray, t), at(ray, h.t), pm);
            tr = tr * p.tr;
        }
        if (pm->refr != h.mat->refr) {
            V3 vE, cvN;
            double fNE, F = 0.0, fNT = 0.0;
            consistent_normal(ray, h, vE, cvN, fNE);
            fresnel_real(fNE, h.mat->refr, pm->refr, F, fNT);
            if (F == 1.0)
                return black;
            tr = tr * (1.0 - F);
        }
        t = h.t;
        pm = h.mat;
    }
    if (pm->i2 && pm->i0 == RT_MAT_TRANSPARENT) {
        Prop p = propagate(s, pm->i2, at(ray, t), at(ray, 1.0), pm);
        tr = tr * p.tr;
    }
    return E * tr;
}

/* Shade.cpp:361-391 */
static V3 reflect_dir(const V3 &cvN, const V3 &vE, const Hit &h, double fNE)
{
    V3 vR = cvN * 2.0 * fNE - vE;
    if (h.perturbed && dot(h.n, vR) < kEps) {
        V3 n2 = normalize(h.n);
        double fNE2 = dot(n2, vE);
        vR = n2 * 2.0 * fNE2 - vE;
    }
    return vR;
}
static V3 refract_dir(const V3 &cvN, const V3 &vE, const Hit &h, double eta, double fNE, double fNT)
{
    V3 vT = cvN * (eta * fNE - fNT) - vE * eta;
    if (h.perturbed && dot(h.n, vT) < kEps) {
        V3 n2 = normalize(h.n);
        double fNE2 = dot(n2, vE);
        vT = n2 * (eta * fNE2 - fNT) - vE * eta;
    }
    return vT;
}

/* Shade.cpp:401-520 */
Spec visual_ray(Scene &s, const Ray &ray, Ctx &cx)
{
    const Spec black = spec_const(0.0f);
    if (cx.depth >= 11)
        return black;
    HitList hl;
    if (cx.depth == 0) cx.drop_scratch();
    if (cx.depth == 0) cx.n_primary++; else cx.n_secondary++;
    node_hits(s.root, ray, cx, hl);
    cx.depth++;
    size_t i;
    Node *pm;
    if (hl.empty() || hl[0].t > 0.0) { pm = &g_b.vacuum; i = 0; } else { pm = hl[0].mat; i = 1; }
    double t = 0.0, fNE = 0.0, fNL, F = 0.0, fNT = 0.0;
    Spec tr = spec_const(1.0f), R = black;
    V3 vE, cvN;
    for (; i < hl.size(); i++) {
        Hit &h = hl[i];
        transform_normals(h, cx);
        if (pm->i0 == RT_MAT_NOSHADE) {         /* Q2 */
            --cx.depth;
            return pm->color;
        }
        if (pm->i2) {
            Prop p = propagate(s, pm->i2, at(ray, t), at(ray, h.t), pm);
            tr = tr * p.tr;
            R = R + p.ster;
        }
        if ((pm->refr != h.mat->refr) || (h.mat->i0 != RT_MAT_TRANSPARENT)) {
            consistent_normal(ray, h, vE, cvN, fNE);
            Spec color = h.mat->color;
            int luster = h.mat->i0;
            if (spectral_metal(luster)) {       /* Q3 */
                color = fresnel_spectral(fNE, pm->refr, g_t.metal_n[luster], g_t.metal_k[luster]);
                F = 1.0;
            } else {
                fresnel_real(fNE, pm->refr, h.mat->refr, F, fNT);
            }
            int rough = roughness(h);
            if (s.has_ambient && rough != RT_SURF_POLISHED) {
                Spec E = s.ambient * kPi;       /* AmbientLight::Illuminate, Lights.cpp:33-37 */
                R = R + color * E;
            }
            Ray sec;
            sec.org = at(ray, h.t);
            for (size_t li = 0; li < s.lights.size(); li++) {
                const Light &l = s.lights[li];
                sec.dir = light_origin(l, cx) - sec.org;
                if (dot(cvN, sec.dir) <= 0.0 && luster != RT_MAT_TRANSPARENT)
                    continue;
                V3 vL = normalize(sec.dir);
                fNL = dot(cvN, vL);
                Spec E = light_ray(s, sec, l, cx) * fNL;
                if (E == black)
                    continue;
                Spec refl = brdf(h, color, vL, vE, cvN, F);
                R = R + refl * E;
            }
            if (rough == RT_SURF_POLISHED) {
                if (F && cx.depth < 6) {
                    Spec w = ray_reflectance(h, color, F);
                    sec.dir = reflect_dir(cvN, vE, h, fNE);
                    R = R + visual_ray(s, sec, cx) * w;
                }
                if (luster == RT_MAT_TRANSPARENT && F < 1.0) {
                    double eta = pm->refr / h.mat->refr;
                    sec.dir = refract_dir(cvN, vE, h, eta, fNE, fNT);
                    R = R + visual_ray(s, sec, cx) * (1.0 - F);
                }
            }
            R = R * tr;
            t = h.t;
            pm = h.mat;
            break;
        } else {
            t = h.t;
            pm = h.mat;
        }
    }
    if (pm->i2) {
        Prop p = propagate(s, pm->i2, at(ray, t), at(ray, kInf), pm);
        R = R + tr * p.ster;
        tr = tr * p.tr;
    }
    --cx.depth;
    return R;
}

} // namespace orc
