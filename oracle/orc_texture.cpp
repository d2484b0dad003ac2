/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 * CPU restatement of the lattice-noise core (Texture.cpp:11-191) and the bump-map call-outs built on
 * it (TextureMaps.cpp:189-369).  All noise arithmetic is float, hashes are 32-bit integer.
 */
#include "orc_shade.h"

namespace orc {

/* Texture.cpp:11-16: multiply, then rotate right by the value itself (mod 32) */
static inline uint32_t lattice(uint32_t n)
{
    n = 1099087573u * n;
    uint32_t s = n & 31u;
    return s ? ((n >> s) | (n << (32u - s))) : n;
}

struct F4 { float x, y, z, w; };
static inline double dot4(const F4 &a, const F4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }   /* Texture.cpp:24-28 */

struct NoiseState {     /* the reference's file-static scratch (Texture.cpp:30-32), made local */
    F4       grad[9], coef[9];
    uint32_t flr[9];
};

static const float kHalf32 = 0.5f * 4294967296.0f;

/* Texture.cpp:34-59 */
static float spline(NoiseState &s, int dim, uint32_t lat)
{
    uint32_t f = s.flr[dim];
    uint32_t h0 = lattice((f - 1) ^ lat), h1 = lattice(f ^ lat), h2 = lattice((f + 1) ^ lat), h3 = lattice((f + 2) ^ lat);
    F4 k;
    if (dim == 1) {
        k.x = float(h0) - kHalf32; k.y = float(h1) - kHalf32; k.z = float(h2) - kHalf32; k.w = float(h3) - kHalf32;
    } else {
        k.x = spline(s, dim - 1, h0); k.y = spline(s, dim - 1, h1); k.z = spline(s, dim - 1, h2); k.w = spline(s, dim - 1, h3);
    }
    return float(dot4(s.coef[dim], k));
}

static void setup(NoiseState &s, const float *p, int ndim, float &gain, bool with_grad)
{
    gain = 1.0f;
    for (int i = 0; i < ndim; i++) {
        gain *= 1.27f;
        int d = ndim - i;
        int fl = int(p[i]) - (p[i] < 0.0f && p[i] != int(p[i]));
        float x = p[i] - fl;
        s.flr[d] = (uint32_t)fl;
        s.coef[d].x = 1.0f / 6.0f + x * (x * (0.5f - 1.0f / 6.0f * x) - 0.5f);
        s.coef[d].y = 4.0f / 6.0f + x * x * (0.5f * x - 1.0f);
        s.coef[d].z = 1.0f / 6.0f + 0.5f * x * (1.0f + x * (1.0f - x));
        s.coef[d].w = 1.0f / 6.0f * x * x * x;
        if (with_grad) {
            s.grad[d].x = -0.5f + x * (1.0f - 0.5f * x);
            s.grad[d].y = x * (1.5f * x - 2.0f);
            s.grad[d].z = 0.5f + x * (1.0f - 1.5f * x);
            s.grad[d].w = 0.5f * x * x;
        }
    }
}

/* Texture.cpp:98-123 (nPow == 1) */
float band_noise(const float *p, int ndim)
{
    NoiseState s;
    float gain;
    setup(s, p, ndim, gain, false);
    return gain * spline(s, ndim, 1099087573u) * (0.9f / 4294967296.0f) + 0.5f;
}

/* Texture.cpp:127-157 */
static float grad_spline(NoiseState &s, float *g, int dim, uint32_t lat, int stride)
{
    uint32_t f = s.flr[dim];
    uint32_t h0 = lattice((f - 1) ^ lat), h1 = lattice(f ^ lat), h2 = lattice((f + 1) ^ lat), h3 = lattice((f + 2) ^ lat);
    F4 k, gk[8];
    if (dim == 1) {
        k.x = float(h0) - kHalf32; k.y = float(h1) - kHalf32; k.z = float(h2) - kHalf32; k.w = float(h3) - kHalf32;
    } else {
        k.x = grad_spline(s, &gk[0].x, dim - 1, h0, 4);
        k.y = grad_spline(s, &gk[0].y, dim - 1, h1, 4);
        k.z = grad_spline(s, &gk[0].z, dim - 1, h2, 4);
        k.w = grad_spline(s, &gk[0].w, dim - 1, h3, 4);
    }
    int is = stride;
    for (int i = 1; i < dim; i++, is += stride)
        g[is] = float(dot4(s.coef[dim], gk[i - 1]));
    g[0] = float(dot4(s.grad[dim], k));
    return float(dot4(s.coef[dim], k));
}

/* Texture.cpp:159-191 (nPow == 1) */
float gradient_band_noise(float *g, const float *p, int ndim)
{
    NoiseState s;
    float gain;
    setup(s, p, ndim, gain, true);
    float v = gain * grad_spline(s, g, ndim, 123456789u, 1) * (0.9f / 4294967296.0f) + 0.5f;
    for (int i = 0; i < ndim; i++)
        g[i] *= gain * (0.9f / 4294967296.0f);
    return v;
}

/* GradientFractalNoise, Texture.cpp:243-297, with the defaults of Texture.h:19-21 (fractal dimension 1, lacunarity 2.1,
   no absolute value) and a whole number of levels (every caller passes 5 or 6, so the remainder branch never runs) */
float gradient_fractal_noise(float *grad, const float *p, int ndim, int levels)
{
    const float lac = 2.1f;
    float omega = powf(lac, -1.0f);
    float wl = 1.0f, scale = 1.0f, noise = 0.0f, gain = 0.0f, sc[3] = { 0, 0, 0 }, bg[3] = { 0, 0, 0 };
    for (int j = 0; j < ndim; j++) { sc[j] = p[j]; grad[j] = 0.0f; }
    for (int i = 0; i < levels; i++) {
        float fb = wl * (gradient_band_noise(bg, sc, ndim) - 0.5f);
        float sign = 1.0f;
        noise += fb;
        for (int j = 0; j < ndim; j++) {
            sc[j] *= lac;
            grad[j] += bg[j] * scale * wl * sign;
        }
        gain += wl;
        wl *= omega;
        scale *= lac;
    }
    gain = 1.0f / gain;
    for (int j = 0; j < ndim; j++) grad[j] *= gain;
    return noise * gain + 0.5f;
}


/* VHash3 + GradientVoronoi, Texture.cpp:318-332, :380-431: nearest feature point among the lattice cells the current
   best distance can reach; the gradient quirk for nPow > 2 (scaled by fmin only nPow - 2 times) is kept */
static inline void vhash3(int nx, int ny, int nz, float &x, float &y, float &z)
{
    uint32_t n;
    n = lattice((uint32_t)(nx ^ 1099087573));
    n = lattice((uint32_t)(ny ^ (int)n ^ 1099087573));
    n = lattice((uint32_t)(nz ^ (int)n ^ 1099087573));
    x = nx + float(n) * (1.0f / 4294967296.0f);
    n = lattice(n ^ 1099087573u);
    y = ny + float(n) * (1.0f / 4294967296.0f);
    n = lattice(n ^ 1099087573u);
    z = nz + float(n) * (1.0f / 4294967296.0f);
}
float gradient_voronoi(float *grad, const float *p, int npow)
{
    float x = p[0], y = p[1], z = p[2], hx, hy, hz;
    int nx = int(floorf(x)), ny = int(floorf(y)), nz = int(floorf(z));
    vhash3(nx, ny, nz, hx, hy, hz);
    double f = (x - hx) * (x - hx) + (y - hy) * (y - hy) + (z - hz) * (z - hz), fmin = f;
    grad[0] = hx; grad[1] = hy; grad[2] = hz;
    f = sqrt(fmin);
    int i0 = int(floor(x - f)), i1 = int(floor(x + f)), j0 = int(floor(y - f)), j1 = int(floor(y + f)), k0 = int(floor(z - f)), k1 = int(floor(z + f));
    for (int i = i0; i <= i1; i++)
        for (int j = j0; j <= j1; j++)
            for (int k = k0; k <= k1; k++)
                if (i != nx || j != ny || k != nz) {
                    vhash3(i, j, k, hx, hy, hz);
                    f = (x - hx) * (x - hx) + (y - hy) * (y - hy) + (z - hz) * (z - hz);
                    if (f < fmin) { fmin = f; grad[0] = hx; grad[1] = hy; grad[2] = hz; }
                }
    f = fmin;
    grad[0] = float(npow) * 2.0f * (x - grad[0]);
    grad[1] = float(npow) * 2.0f * (y - grad[1]);
    grad[2] = float(npow) * 2.0f * (z - grad[2]);
    for (int i = 1; i < npow; i++) {
        f *= fmin;
        if (i > 1) { grad[0] *= float(fmin); grad[1] *= float(fmin); grad[2] *= float(fmin); }
    }
    return float(f);
}

/* FractalNoise, Texture.cpp:200-241, as the solid textures call it: fractal dimension 1, lacunarity 1.9, a whole number
   of levels (the remainder branch never runs) */
static float fractal_noise(const float *p, int ndim, int levels, int absolute)
{
    const float lac = 1.9f, omega = powf(lac, -1.0f);
    float wl = 1.0f, noise = 0.0f, gain = 0.0f, sc[3] = { 0.0f, 0.0f, 0.0f };
    for (int j = 0; j < ndim; j++) sc[j] = p[j];
    for (int i = 0; i < levels; i++) {
        float fb = wl * (band_noise(sc, ndim) - 0.5f);
        if (absolute) fb = fabsf(fb);
        noise += fb;
        gain += wl;
        wl *= omega;
        for (int j = 0; j < ndim; j++) sc[j] *= lac;
    }
    return noise / gain + 0.5f;
}
float fractal_noise_export(const float *p, int ndim, int levels, int absolute) { return fractal_noise(p, ndim, levels, absolute); }
static double sawtooth_wave(double x)                   /* TextureMaps.cpp:546-555 */
{
    x = (0.5 / kPi) * x + 0.25;
    x = 2.0 * (x - floor(x));
    if (x < 1.0) return 2.0 * x - 1.0;
    return 3.0 - 2.0 * x;
}
static double wood_wave(double x)                       /* TextureMaps.cpp:590-604 */
{
    x = (0.5 / kPi) * x + 0.25;
    x = 2.0 * (x - floor(x));
    if (x < 1.0) return 0.0;
    else if (x < 1.5) { x = 2.0 * (x - 1.0); return 0.5 * x * x; }
    else { x = 2.0 * (x - 2.0); return -0.5 * x * x + 1.0; }
}
/* the solid colour textures, TextureMaps.cpp:518-683: colour (and for Spots the luster) of the material this hit enters;
   written to a private copy of the material (Q13, orc_solids.h) */
static void colour_texture(int tex, const V3 &p, Hit &h, const Ctx &cx)
{
    const double sc = tex == RT_TEX_WOOD ? 3.0 : 7.0;
    float q[3] = { float(p.x * sc), float(p.y * sc), float(p.z * sc) };
    Node *m = cx.clone_material(h.mat);
    h.mat = m;
    switch (tex) {
    case RT_TEX_VORONOI: {
        float g[3];
        double fBlend = gradient_voronoi(g, q, 1) / 1.6;        /* Voronoi(rgf): the same search, value only (Texture.cpp:334-378) */
        if (fBlend > 1.0) fBlend = 1.0;
        m->color = g_t.s_green * fBlend + g_t.s_white * (1.0 - fBlend);
        return;
    }
    case RT_TEX_AGATE: {
        double fBlend = band_noise(q, 3);
        fBlend = 0.5 * (sin(100.0 * fBlend) + 1.0);
        m->color = g_t.s_orange * fBlend + g_t.s_white * (1.0 - fBlend);
        return;
    }
    case RT_TEX_MARBLE: {
        Spec pink = g_t.s_white * 0.67 + g_t.s_red * 0.33;
        double d = sawtooth_wave(q[2] + 12.0 * fractal_noise(q, 3, 5, 1));
        d = d * d;
        m->color = pink * (1.0 - d);
        return;
    }
    case RT_TEX_CLOUDS: {
        double fBlend = 2.0 * fractal_noise(q, 3, 5, 0) - 1.0;
        if (fBlend < 0.0) fBlend = 0.0; else if (fBlend > 1.0) fBlend = 1.0;
        m->color = g_t.s_white * fBlend + g_t.s_blue * (1.0 - fBlend);
        return;
    }
    case RT_TEX_WOOD: {
        double fWarp = 3.0f * fractal_noise(q, 3, 2, 0);
        q[0] += float(fWarp);
        double fBlend = 0.5 + 0.5 * wood_wave(50.0 * sqrt(q[0] * q[0] + q[1] * q[1]));
        fBlend = 0.9 * fBlend + 0.1 * sawtooth_wave(250.0 * (1.0 + fBlend) * sqrt(q[0] * q[0] + q[1] * q[1]));
        fBlend = 1.0 - fBlend;
        m->color = (g_t.s_orange + g_t.s_red) * (0.25 * fBlend) + g_t.s_white * 0.03;
        return;
    }
    case RT_TEX_SPOTS: {
        double fBlend = band_noise(q, 3);
        if (fBlend > 0.5) m->i0 = RT_MAT_COPPER;
        else { m->i0 = RT_MAT_PLASTIC; m->color = spec_const(0.0f); }
        return;
    }
    case RT_TEX_VEINS: {
        double fBlend = fractal_noise(q, 3, 5, 0) - 0.5;
        if (fabs(fBlend) > 0.02) m->color = g_t.s_green * 0.125; else m->color = g_t.s_white;
        return;
    }
    }
}

/* TextureMaps.cpp:189-198 */
static void perturb(Hit &h, const V3 &d)
{
    if (h.perturbed) {
        h.np = h.np + d;
    } else {
        h.np = normalize(h.n) + d;
        h.perturbed++;
    }
}

/* ---- the planet image maps ---- */
static int img_wrap(int i, int len, int wrap)                   /* ML_Modulus, ML_Wrap: Image.cpp:50-83 */
{
    if (wrap == RT_WRAP_AROUND) { i = i % len; if (i < 0) i += len; return i; }
    if (wrap == RT_WRAP_MIRROR) { i = i % (2 * len); if (i < 0) i += 2 * len; if (i >= len) i = 2 * len - i - 1; return i; }
    return i < 0 ? 0 : (i >= len ? len - 1 : i);
}
static float img_get(const RtImageMap &im, int i, int j, int k)  /* Image::Get, Image.h:226-236 */
{
    if ((unsigned)i >= (unsigned)im.width || (unsigned)j >= (unsigned)im.height) { i = img_wrap(i, im.width, im.wrap_x); j = img_wrap(j, im.height, im.wrap_y); }
    return im.pixels[(size_t)i + (size_t)im.width * ((size_t)j + (size_t)im.height * (size_t)k)];
}
static double img_sample_lerp(const RtImageMap &im, double x, double y, int k = 0)      /* Image.cpp:1262-1283 */
{
    x -= 0.5; y -= 0.5;
    const int i1 = (int)(long long)floor(x), j1 = (int)(long long)floor(y);     /* unsigned(floor(x)) handed to Get(int ...) */
    const double dx = x - floor(x), dy = y - floor(y);
    double a = img_get(im, i1, j1, k), b = img_get(im, i1 + 1, j1, k), c = img_get(im, i1, j1 + 1, k), d = img_get(im, i1 + 1, j1 + 1, k);
    a = a + dx * (b - a);
    c = c + dx * (d - c);
    return double(a + dy * (c - a));
}
struct Rgb3 { float r, g, b; };
static Rgb3 img_get_rgb(const RtImageMap &im, int i, int j)       /* Image::GetRGB, Image.h:266-281 */
{
    Rgb3 c;
    if (im.channels < 3) { c.r = c.g = c.b = img_get(im, i, j, 0); return c; }
    c.r = img_get(im, i, j, 0); c.g = img_get(im, i, j, 1); c.b = img_get(im, i, j, 2);
    return c;
}
/* DisplayRGB arithmetic, Image.h:64-69 */
static Rgb3 rgb_add(const Rgb3 &a, const Rgb3 &b) { Rgb3 c = { float(double(a.r + b.r)), float(double(a.g + b.g)), float(double(a.b + b.b)) }; return c; }
static Rgb3 rgb_sub(const Rgb3 &a, const Rgb3 &b) { Rgb3 c = { float(double(a.r - b.r)), float(double(a.g - b.g)), float(double(a.b - b.b)) }; return c; }
static Rgb3 rgb_mul(const Rgb3 &a, double f) { Rgb3 c = { float(a.r * f), float(a.g * f), float(a.b * f) }; return c; }
static Rgb3 img_sample_lerp_rgb(const RtImageMap &im, double x, double y)                /* Image.cpp:1285-1307 */
{
    x -= 0.5; y -= 0.5;
    const int i1 = (int)(long long)floor(x), j1 = (int)(long long)floor(y);
    const double dx = x - floor(x), dy = y - floor(y);
    Rgb3 a = img_get_rgb(im, i1, j1), b = img_get_rgb(im, i1 + 1, j1), c = img_get_rgb(im, i1, j1 + 1), d = img_get_rgb(im, i1 + 1, j1 + 1);
    a = rgb_add(a, rgb_mul(rgb_sub(b, a), dx));
    c = rgb_add(c, rgb_mul(rgb_sub(d, c), dx));
    return rgb_add(a, rgb_mul(rgb_sub(c, a), dy));
}
static void img_sample_gradient(const RtImageMap &im, double x, double y, float g[2])  /* Image.h:187-189 */
{
    g[0] = float(img_sample_lerp(im, x + 0.5, y) - img_sample_lerp(im, x - 0.5, y));
    g[1] = float(img_sample_lerp(im, x, y + 0.5) - img_sample_lerp(im, x, y - 0.5));
}
static Spec rgb_to_spectrum_6500(const Rgb3 &rgb)                /* RGBtoSpectrum(rgb, 1), RadiometryColor.cpp:947-963 */
{
    Spec s;
    if (rgb.r < rgb.g && rgb.r < rgb.b) s = g_t.s_gwhite * rgb.r + g_t.s_green * (rgb.g - rgb.r) + g_t.s_blue * (rgb.b - rgb.r);
    else if (rgb.g < rgb.r && rgb.g < rgb.b) s = g_t.s_gwhite * rgb.g + g_t.s_red * (rgb.r - rgb.g) + g_t.s_blue * (rgb.b - rgb.g);
    else s = g_t.s_gwhite * rgb.b + g_t.s_red * (rgb.r - rgb.b) + g_t.s_green * (rgb.g - rgb.b);
    for (int i = 0; i < NSPEC; i++) s.v[i] = g_t.s_d6500.v[i] ? s.v[i] / g_t.s_d6500.v[i] : 0.0f;      /* Spectrum / Spectrum, RadiometryColor.h:34-42 */
    return s;
}
static void perturb(Hit &h, const V3 &d);
/* PlanetMapping, TextureMaps.cpp:955-1084, with its callers' images (:1098-1243).  The writes to the shared Material / Surface go to
   private copies, as for the solid colour textures (Q13).  Q15: without a bump map the reference hands three uninitialised
   doubles to PerturbNormal (:1055-1083, the sky dome): here the normal is left alone. */
static void planet_texture(int tex, const V3 &p, Hit &h, const Ctx &cx, const RtImageMap *images)
{
    static const int slots[7][3] = { { RT_IMG_EARTH, RT_IMG_EARTHBUMP, RT_IMG_EARTHWATER }, { RT_IMG_EARTH, RT_IMG_EARTHBUMP, -1 }, { RT_IMG_MARS, RT_IMG_MARSBUMP, -1 },
                                     { RT_IMG_MOON, RT_IMG_MOONBUMP, -1 }, { RT_IMG_VENUS, RT_IMG_VENUSBUMP, -1 }, { RT_IMG_MERCURY, RT_IMG_MERCURYBUMP, -1 }, { RT_IMG_SKY, -1, -1 } };
    if (!images) return;
    const int *sl = slots[tex - RT_TEX_EARTH];
    const RtImageMap *pimColor = &images[sl[0]], *pimBump = sl[1] >= 0 ? &images[sl[1]] : 0, *pimWater = sl[2] >= 0 ? &images[sl[2]] : 0;
    double fLat, fLong, w, x, y, z, r, fScale, fKmPerPixel, fSinLat, fCosLat, fSinLong, fCosLong;
    double xGrad, yGrad, zGrad, xNormal, yNormal, zNormal;
    float rgfGrad[2];
    int bPolar;
    const V3 cvN = normalize(h.n);
    x = cvN.x; y = cvN.y; z = cvN.z;
    fLat = asin(z);
    fLong = kPi - atan2(p.y, p.x);
    w = 8192.0;                                     /* SPHEREMAPWIDTH */
    fKmPerPixel = 2.0 * kPi * 6378.137 / w;
    if (fLong < 0.0) fLong += 2.0 * kPi;
    fSinLat = z;
    fCosLat = sqrt(1.0 - z * z);
    r = sqrt(x * x + y * y);
    fSinLong = y / r;
    fCosLong = x / r;
    if (fabs(fLat * 180.0 / kPi) > 32.70) {         /* MAPSELECT */
        double xPolar, yPolar, fScalePolar;
        if (fabs(fLat) < 0.5 * kPi - 0.001) fScalePolar = (0.5 * kPi - fabs(fLat)) / r;
        else fScalePolar = 1 / fabs(z);
        if (fLat > 0.0) {
            xPolar = +x / (1.0 + z) / kPi + 0.25;
            yPolar = +y / (1.0 + z) / kPi + 0.25;
            if (xPolar < 0.0 || xPolar >= 0.5) fScalePolar = 0.0;
        } else {
            xPolar = +x / (1.0 - z) / kPi + 0.75;
            yPolar = -y / (1.0 - z) / kPi + 0.25;
            if (xPolar < 0.5 || xPolar >= 1.0) fScalePolar = 0.0;
        }
        if (yPolar < 0.0 || yPolar >= 0.5) fScalePolar = 0.0;
        x = xPolar; y = yPolar; fScale = fScalePolar; bPolar = 1;
    } else {
        double xCyl = fLong / (2.0 * kPi), fScaleCyl;
        if (xCyl >= 1.0) xCyl -= 1.0; else if (xCyl < 0.0) xCyl += 1.0;
        const double yCyl = -log(tan(0.25 * kPi + 0.5 * fLat)) / (2.0 * kPi) + 0.75;
        if (yCyl >= 0.5 && yCyl < 1.0) fScaleCyl = 1.0 / r; else fScaleCyl = 0.0;
        x = xCyl; y = yCyl; fScale = fScaleCyl; bPolar = 0;
    }
    Node *m = cx.clone_material(h.mat);
    h.mat = m;
    if (pimWater && img_sample_lerp(*pimWater, w * x, w * y) > 0.0) {
        m->color = g_t.s_ocean;
        h.surf = cx.clone_material(h.surf ? h.surf : &g_b.matte); h.surf->i0 = RT_SURF_SHINY;
        img_sample_gradient(*pimWater, w * x, w * y, rgfGrad);
        xGrad = rgfGrad[0] * 256.0 * fScale / fKmPerPixel;
        yGrad = rgfGrad[1] * 256.0 * fScale / fKmPerPixel;
        zGrad = 1.0;
    } else {
        m->color = rgb_to_spectrum_6500(img_sample_lerp_rgb(*pimColor, w * x, w * y));
        if (pimBump) {
            h.surf = cx.clone_material(h.surf ? h.surf : &g_b.matte); h.surf->i0 = RT_SURF_SHINY;
            img_sample_gradient(*pimBump, w * x, w * y, rgfGrad);
            xGrad = rgfGrad[0] * 256.0 * fScale / fKmPerPixel;
            yGrad = rgfGrad[1] * 256.0 * fScale / fKmPerPixel;
            zGrad = 0.0;
        } else {
            m->i0 = RT_MAT_NOSHADE;
            return;                                 /* Q15 */
        }
    }
    if (bPolar) {
        x = +fCosLong * xGrad - -fSinLong * yGrad;
        y = -fSinLong * xGrad + +fCosLong * yGrad;
        z = zGrad;
        xNormal = fCosLat * z - fSinLat * x;
        yNormal = y;
        zNormal = fSinLat * z + fCosLat * x;
        x = fCosLong * xNormal - fSinLong * yNormal;
        y = fSinLong * xNormal + fCosLong * yNormal;
        z = zNormal;
    } else {
        y = xGrad;
        x = fCosLat * zGrad - fSinLat * yGrad;
        z = fSinLat * zGrad + fCosLat * yGrad;
        zNormal = z;
        xNormal = fCosLong * x - fSinLong * y;
        yNormal = fSinLong * x + fCosLong * y;
        z = zNormal; x = xNormal; y = yNormal;
    }
    perturb(h, V3(x, y, z));
}

static void bump3(const V3 &p, Hit &h, double scale)      /* Bumpy / MicroFacets / MicroFacets2: TextureMaps.cpp:209-249 */
{
    float q[3] = { float(p.x * scale), float(p.y * scale), float(p.z * scale) }, g[3];
    gradient_band_noise(g, q, 3);
    perturb(h, V3(g[0], g[1], g[2]) * 0.3);
}

extern const V3 kDodecahedronNormals[20];

void apply_texture(int tex, const V3 &p, Hit &h, const Ctx &cx, const RtImageMap *images)
{
    switch (tex) {
    case RT_TEX_EARTH: case RT_TEX_EARTH2: case RT_TEX_MARS: case RT_TEX_MOON: case RT_TEX_VENUS: case RT_TEX_MERCURY: case RT_TEX_SKY:
        planet_texture(tex, p, h, cx, images); return;
    case RT_TEX_VORONOI: case RT_TEX_AGATE: case RT_TEX_MARBLE: case RT_TEX_CLOUDS: case RT_TEX_WOOD: case RT_TEX_SPOTS: case RT_TEX_VEINS:
        colour_texture(tex, p, h, cx); return;
    case RT_TEX_SEASCAPE:     return;       /* TextureMaps.cpp:497-513: the loop computes points and discards them */
    case RT_TEX_BUMPY:        bump3(p, h, 50.0); return;
    case RT_TEX_MICROFACETS:  bump3(p, h, 500.0); return;
    case RT_TEX_MICROFACETS2: bump3(p, h, 5000.0); return;
    case RT_TEX_FIBERS: {                                   /* TextureMaps.cpp:338-349 */
        float q[3] = { float(p.x * 250.0), float(p.y * 250.0), 0.0f }, g[3];
        gradient_band_noise(g, q, 2);
        perturb(h, V3(g[0], g[1], 0.0) * 0.3);
        return;
    }
    case RT_TEX_BLISTERED: {                                /* TextureMaps.cpp:351-369 */
        float q[3] = { float(p.x * 25.0), float(p.y * 25.0), float(p.z * 25.0) }, g[3], f;
        f = gradient_band_noise(g, q, 3);
        if (f > 0.5) f = -0.5; else f = 0.0;
        perturb(h, V3(g[0], g[1], g[2]) * f);
        return;
    }
    case RT_TEX_FACETED20: {                                /* TextureMaps.cpp:281-305 */
        float q[3] = { float(p.x * 50.0), float(p.y * 50.0), float(p.z * 50.0) }, g[3];
        gradient_band_noise(g, q, 3);
        V3 cv(g[0], g[1], g[2]), best;
        double bestd = -1000000.0;
        for (int i = 0; i < 20; i++) {
            double d = dot(cv, kDodecahedronNormals[i]);
            if (d > bestd) { bestd = d; best = kDodecahedronNormals[i]; }
        }
        perturb(h, best * 0.15);
        return;
    }
    case RT_TEX_WRINKLED: {                                 /* TextureMaps.cpp:267-279 */
        float q[3] = { float(p.x * 50.0), float(p.y * 50.0), float(p.z * 50.0) }, g[3];
        gradient_fractal_noise(g, q, 3, 6);
        perturb(h, V3(g[0], g[1], g[2]) * 0.3);
        return;
    }
    case RT_TEX_DENTED: {                                   /* TextureMaps.cpp:307-328 */
        V3 p2 = p + V3(13.0, 17.0, 31.0);
        float q[3] = { float(p2.x * 40.0), float(p2.y * 40.0), float(p2.z * 40.0) }, g[3], f;
        f = gradient_fractal_noise(g, q, 3, 5);
        if (f < 0.5f) { h.np = h.n; return; }               /* cvPerturbed = cvNormal, nPerturbed untouched */
        f = 3.0f * f * f;
        perturb(h, V3(g[0], g[1], g[2]) * (1.0 * f));
        return;
    }
    case RT_TEX_HAMMERED: {                                 /* TextureMaps.cpp:371-387 */
        float q[3] = { float(p.x * 8.0), float(p.y * 8.0), float(p.z * 8.0) }, g[3];
        gradient_band_noise(g, q, 3);
        V3 cv(g[0], g[1], g[2]);
        gradient_fractal_noise(g, q, 3, 5);
        V3 cv2(g[0], g[1], g[2]);
        perturb(h, cv * 0.1 + cv2 * 0.1);
        return;
    }
    case RT_TEX_TECTONIC: {                                 /* TextureMaps.cpp:389-405 */
        float q[3] = { float(p.x * 3.0), float(p.y * 3.0), float(p.z * 3.0) }, g[3], f;
        f = gradient_fractal_noise(g, q, 3, 6);
        if (f < 0.5f) f = 0.0f; else f = -0.4f;
        perturb(h, V3(g[0], g[1], g[2]) * f);
        return;
    }
    case RT_TEX_VORONOIBUMP: case RT_TEX_VORONOIANTIBUMP: case RT_TEX_VORONOICRINKLED: {     /* TextureMaps.cpp:407-441 */
        float q[3] = { float(p.x * 7.0), float(p.y * 7.0), float(p.z * 7.0) }, g[3];
        gradient_voronoi(g, q, tex == RT_TEX_VORONOICRINKLED ? 4 : 1);
        if (tex == RT_TEX_VORONOIBUMP) perturb(h, V3(g[0], g[1], g[2]) * 0.3);
        else perturb(h, V3(-g[0], -g[1], -g[2]) * 0.3);
        return;
    }
    case RT_TEX_WAVY: {                                     /* TextureMaps.cpp:330-336: CoVector3(double) splats the scalar */
        double f = 0.3 * cos(p.z / (2.0 * kPi));
        perturb(h, V3(f, f, f));
        return;
    }
    case RT_TEX_QUASIMICROFACET: {                          /* TextureMaps.cpp:251-265 */
        V3 n = h.perturbed ? h.np : h.n;
        V3 cu = normalize(cross(n, V3(0.0, 1.0, 0.0)));
        V3 cvv = normalize(cross(n, cu));
        double u = cx.aux[4], v = cx.aux[5];
        u = u * u * u * u * u;
        v = v * v * v * v * v;
        perturb(h, (cu * u + cvv * v) * 0.25);
        return;
    }
    default:
        return;
    }
}

/* rgvDodecahedron, Solids.cpp:649-711 */
#define T_  (1.0 / kSqrt3)
#define P1_ (1.61803398874989484820458683436563811772030917980576 / kSqrt3)
#define P2_ ((1.61803398874989484820458683436563811772030917980576 - 1.0) / kSqrt3)
const V3 kDodecahedronNormals[20] = {
    V3(T_, T_, T_), V3(T_, T_, -T_), V3(T_, -T_, T_), V3(T_, -T_, -T_),
    V3(-T_, T_, T_), V3(-T_, T_, -T_), V3(-T_, -T_, T_), V3(-T_, -T_, -T_),
    V3(0.0, P2_, P1_), V3(0.0, P2_, -P1_), V3(0.0, -P2_, P1_), V3(0.0, -P2_, -P1_),
    V3(P1_, 0.0, P2_), V3(P1_, 0.0, -P2_), V3(-P1_, 0.0, P2_), V3(-P1_, 0.0, -P2_),
    V3(P2_, P1_, 0.0), V3(P2_, -P1_, 0.0), V3(-P2_, P1_, 0.0), V3(-P2_, -P1_, 0.0)
};

} // namespace orc
