/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 * C entry points used by tests/, smoke() and bench.py's cpu_baseline leg (through ctypes):
 * scene construction from an RtSceneDesc, the sampling tiles and reconstruction filters
 * (Sampling.cpp, Image.cpp), the integer PRNG and quasi-random permutation (MonteCarlo.h,
 * Patterns.cpp), plus probes that expose intermediate values for parity tests.
 *
 * PARITY PINNING: the reference (MSVC/Win32 only: <Windows.h>, <tchar.h>, 131 non-const temporaries)
 * cannot be compiled in this image without stand-in headers and source patches, so it is treated as
 * unbuildable.  This restatement is pinned by (a) the reference's checked-in outputs of
 * BuildQuasiIndex (tests/golden/quasi_index_tables.json) and (b) the image checksums the survey
 * recorded from the reference itself (tests/golden/survey_checksums.json).
 */
#include <stdlib.h>
#include <string>
#include <chrono>
#include "orc_shade.h"
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

/* ------------------------------------------------------------------------------------------------
 * tables.bin reader
 * ---------------------------------------------------------------------------------------------- */
static bool find_table(const std::vector<unsigned char> &blob, const char *name, int &dtype, int &count, size_t &off)
{
    if (blob.size() < 16 || memcmp(&blob[0], "RTTABLE1", 8)) return false;
    uint32_t n;
    memcpy(&n, &blob[8], 4);
    for (uint32_t i = 0; i < n; i++) {
        const unsigned char *e = &blob[16 + 56 * i];
        if (!strncmp((const char *)e, name, 40)) {
            uint32_t dt, cnt; uint64_t o;
            memcpy(&dt, e + 40, 4); memcpy(&cnt, e + 44, 4); memcpy(&o, e + 48, 8);
            dtype = (int)dt; count = (int)cnt; off = (size_t)o;
            return true;
        }
    }
    return false;
}

static bool load_spec(const std::vector<unsigned char> &blob, const char *name, Spec &s)
{
    int dt, cnt; size_t off;
    if (!find_table(blob, name, dt, cnt, off) || dt != 0 || cnt != NSPEC) return false;
    memcpy(s.v, &blob[off], sizeof(float) * NSPEC);
    return true;
}

extern "C" int orc_load_tables(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    std::vector<unsigned char> blob;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n);
    fclose(f);
    static const char *metal[11] = { 0, 0, 0, 0, "Copper", "Gold", "Iron", "Aluminum", "Osmium", "Silver", "Titanium" };
    for (int m = RT_MAT_COPPER; m <= RT_MAT_TITANIUM; m++) {
        std::string a = std::string("g_s") + metal[m] + "Refraction", b = std::string("g_s") + metal[m] + "Extinction";
        if (!load_spec(blob, a.c_str(), g_t.metal_n[m]) || !load_spec(blob, b.c_str(), g_t.metal_k[m])) return -2;
    }
    if (!load_spec(blob, "g_sRedChannel", g_t.chan[0]) || !load_spec(blob, "g_sGrnChannel", g_t.chan[1]) ||
        !load_spec(blob, "g_sBluChannel", g_t.chan[2])) return -3;
    int dt, cnt; size_t off;
    if (!find_table(blob, "SpectrumToRGB", dt, cnt, off) || dt != 0 || cnt != 30) return -4;
    memcpy(g_t.spec2rgb, &blob[off], sizeof(float) * 30);
    if (!find_table(blob, "g_rgvBlueNoise", dt, cnt, off) || dt != 1 || cnt != 2 * RT_BLUE_N) return -5;
    g_t.blue.resize(cnt);
    memcpy(&g_t.blue[0], &blob[off], sizeof(double) * cnt);
    if (!load_spec(blob, "g_sGreen", g_t.s_green) || !load_spec(blob, "g_sEqualWhite", g_t.s_white) || !load_spec(blob, "g_sOrange", g_t.s_orange) ||
        !load_spec(blob, "g_sRed", g_t.s_red) || !load_spec(blob, "g_sBlue", g_t.s_blue)) return -7;
    {   /* PlanetMapping's spectra: g_sWhite, g_sD6500 (RGBtoSpectrum) and s_sOcean (TextureMaps.cpp:1123) */
        Spec cyan;
        if (!load_spec(blob, "g_sWhite", g_t.s_gwhite) || !load_spec(blob, "g_sD6500", g_t.s_d6500) || !load_spec(blob, "g_sCyan", cyan)) return -8;
        g_t.s_ocean = g_t.s_white * 0.20 + cyan * 0.20 + g_t.s_blue * 0.30;
    }
    if (!find_table(blob, "DiskSamples128", dt, cnt, off) || dt != 1 || cnt != 256) return -6;
    g_t.disk128.resize(cnt);
    memcpy(&g_t.disk128[0], &blob[off], sizeof(double) * cnt);
    g_t.loaded = true;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Marsaglia multiply-with-carry PRNG, MonteCarlo.h:17-63, MonteCarlo.cpp:9-25
 * ---------------------------------------------------------------------------------------------- */
struct Mwc { uint32_t x, c; };
static inline uint32_t mwc_next(Mwc &s)
{
    uint64_t n = (uint64_t)1965537969u * s.x + s.c;
    s.x = (uint32_t)(n & 0xFFFFFFFFu);
    s.c = (uint32_t)(n >> 32);
    return s.x;
}
static inline double mwc_double(Mwc &s)
{
    double a = double(mwc_next(s));        /* MonteCarlo.h:38 -- operand order of the two draws is unspecified C++ */
    double b = double(mwc_next(s));        /* (g++ and MSVC both evaluate left to right here; stated policy) */
    return (a + b * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0);
}
static uint32_t mwc_less_than(Mwc &s, uint32_t mod)
{
    uint32_t rem = 0xFFFFFFFFu % mod, n;
    do { n = mwc_next(s); } while (n <= rem);
    return n % mod;
}

extern "C" void orc_random_unsigned(uint32_t *state /*x,c*/, int n, uint32_t *out)
{
    Mwc s = { state[0], state[1] };
    for (int i = 0; i < n; i++) out[i] = mwc_next(s);
    state[0] = s.x; state[1] = s.c;
}
extern "C" void orc_random_double(uint32_t *state, int n, double *out)
{
    Mwc s = { state[0], state[1] };
    for (int i = 0; i < n; i++) out[i] = mwc_double(s);
    state[0] = s.x; state[1] = s.c;
}
extern "C" void orc_random_shuffle(uint32_t *state, int *items, int n)      /* MonteCarlo.h:50-63 */
{
    Mwc s = { state[0], state[1] };
    for (int j = n - 1; j > 0; --j) {
        int i = (int)mwc_less_than(s, (uint32_t)(j + 1));
        int t = items[i]; items[i] = items[j]; items[j] = t;
    }
    state[0] = s.x; state[1] = s.c;
}

/* ------------------------------------------------------------------------------------------------
 * BuildQuasiIndex, Patterns.cpp:1667-1736
 * ---------------------------------------------------------------------------------------------- */
struct NumDigit { int N, digit; };
static int digit_of(int n, int ith, int base) { for (int i = 0; i < ith; i++) n = n / base; return n % base; }
static void quasi_shuffle(NumDigit *a, int ith, int base, int start, int limit)
{
    if (limit - start < 2) return;
    for (int i = start; i < limit; i++) a[i].digit = digit_of(a[i].N, ith, base);
    for (int j = start + 1; j < limit; j++) {       /* stable insertion sort on the digit */
        NumDigit key = a[j];
        int i;
        for (i = j; i > start && key.digit < a[i - 1].digit; i--) a[i] = a[i - 1];
        a[i] = key;
    }
    int s0 = start, d = a[s0].digit;
    for (int lim = s0 + 1; lim <= limit; ) {
        /* the reference tests rgnd[iLimit].nDigit before iLimit == nLimit (one-past read, Patterns.cpp:1712);
           the outcome is the same, so the bound is tested first here */
        if (lim == limit || a[lim].digit != d) {
            quasi_shuffle(a, ith + 1, base, s0, lim);
            s0 = lim;
            lim = s0 + 1;
            if (s0 < limit) d = a[s0].digit;
            continue;
        }
        lim++;
    }
}
extern "C" void orc_build_quasi_index(int *out, int prime, int n)
{
    std::vector<NumDigit> a(n + 1);
    for (int i = 0; i < n; i++) a[i].N = i;
    quasi_shuffle(&a[0], 0, prime, 0, n);
    for (int i = 0; i < n; i++) out[i] = a[i].N;
}

/* ------------------------------------------------------------------------------------------------
 * Scene handle
 * ---------------------------------------------------------------------------------------------- */
static Spec spec_from(const float *f) { Spec s; for (int i = 0; i < NSPEC; i++) s.v[i] = f[i]; return s; }

extern "C" void *orc_scene_create(const RtSceneDesc *d, int run_optimize)
{
    if (!g_t.loaded) return 0;
    Scene *s = new Scene;
    s->nodes.resize(d->n_nodes);
    for (int i = 0; i < d->n_nodes; i++) s->nodes[i] = new Node;
    for (int i = 0; i < d->n_nodes; i++) {
        const RtNode &rn = d->nodes[i];
        Node *n = s->nodes[i];
        const double *p = d->data + rn.data;
        n->type = rn.type; n->id = rn.id; n->index = i;
        n->left = rn.left >= 0 ? s->nodes[rn.left] : 0;
        n->right = rn.right >= 0 ? s->nodes[rn.right] : 0;
        n->link = rn.link >= 0 ? s->nodes[rn.link] : 0;
        n->i0 = rn.i0; n->i1 = rn.i1; n->i2 = rn.i2;
        switch (rn.type) {
        case RT_NODE_CONEHYPERCYL: n->Rneck = p[3];     /* falls through */
        case RT_NODE_QUADRIC:      n->P = p[0]; n->Q = p[1]; n->R = p[2]; break;
        case RT_NODE_POLYHEDRON:
            for (int k = 0; k < rn.i0; k++) n->planes.push_back(V3(p[3 * k], p[3 * k + 1], p[3 * k + 2]));
            break;
        case RT_NODE_TORUS:        n->minor = p[0]; break;
        case RT_NODE_MESH:
            for (int k = 0; k < rn.i0; k++) {
                V3 v(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
                n->mverts.push_back(v);
                if (k == 0) n->ex.lo = n->ex.hi = v;            /* ObjectFormat.cpp:51-56 */
                else {
                    n->ex.lo = V3(v.x < n->ex.lo.x ? v.x : n->ex.lo.x, v.y < n->ex.lo.y ? v.y : n->ex.lo.y, v.z < n->ex.lo.z ? v.z : n->ex.lo.z);
                    n->ex.hi = V3(v.x > n->ex.hi.x ? v.x : n->ex.hi.x, v.y > n->ex.hi.y ? v.y : n->ex.hi.y, v.z > n->ex.hi.z ? v.z : n->ex.hi.z);
                }
            }
            for (int k = 0; k < 3 * rn.i1; k++) n->mtris.push_back((int)p[3 * (size_t)rn.i0 + k]);
            break;
        case RT_NODE_AFFINE:
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { n->m.m[r][c] = p[3 * r + c]; n->mi.m[r][c] = p[9 + 3 * r + c]; }
            n->tr = V3(p[18], p[19], p[20]);
            break;
        case RT_NODE_ANGULARCLIP:  n->cv1 = V3(p[0], p[1], p[2]); n->cv2 = V3(p[3], p[4], p[5]); break;
        case RT_NODE_BOUNDINGBOX:  n->ex.lo = V3(p[0], p[1], p[2]); n->ex.hi = V3(p[3], p[4], p[5]); break;
        case RT_NODE_MATERIAL:
            n->density = p[0]; n->refr = p[1];
            for (int k = 0; k < NSPEC; k++) n->color.v[k] = float(p[2 + k]);
            break;
        default: break;
        }
    }
    if (d->n_images == RT_IMG__COUNT && d->images) {
        s->images.assign(d->images, d->images + RT_IMG__COUNT);
        s->image_px.resize(RT_IMG__COUNT);
        for (int k = 0; k < RT_IMG__COUNT; k++) {
            RtImageMap &m = s->images[k];
            if (!m.pixels) continue;
            s->image_px[k].assign(m.pixels, m.pixels + (size_t)m.width * m.height * m.channels);
            m.pixels = &s->image_px[k][0];
        }
        for (int i = 0; i < d->n_nodes; i++)
            if (s->nodes[i]->type == RT_NODE_MATERIAL || s->nodes[i]->type == RT_NODE_SURFACE) s->nodes[i]->images = &s->images[0];
    }
    s->root = d->root >= 0 ? s->nodes[d->root] : 0;
    if (run_optimize && s->root)
        s->root = optimize(s->root, 0);
    for (int i = 0; i < d->n_lights; i++) {
        Light l;
        const RtLight &rl = d->lights[i];
        l.type = rl.type; l.no_shadow = rl.no_shadow;
        l.org = V3(rl.org[0], rl.org[1], rl.org[2]);
        l.vA = V3(rl.vA[0], rl.vA[1], rl.vA[2]);
        l.vB = V3(rl.vB[0], rl.vB[1], rl.vB[2]);
        l.flux = spec_from(rl.flux);
        s->lights.push_back(l);
    }
    s->has_ambient = d->has_ambient != 0;
    s->ambient = spec_from(d->ambient_flux);
    s->cam = d->camera;
    s->sampler_type = d->sampler.type;
    s->venera = d->sampler.venera;
    if (d->sampler.qindex2) s->qi2.assign(d->sampler.qindex2, d->sampler.qindex2 + RT_BLUE_N);
    if (d->sampler.qindex3) s->qi3.assign(d->sampler.qindex3, d->sampler.qindex3 + RT_BLUE_N);
    for (int i = 0; i < 6; i++) s->aux_init[i] = d->sampler.aux_init[i];
    s->prng_x = d->sampler.prng_x; s->prng_c = d->sampler.prng_c;
    return s;
}

extern "C" void orc_scene_destroy(void *h) { delete (Scene *)h; }

/* Extent of every BoundingBox node recomputed from its subtree (Solids.h:494), for checking the
   front end's GetExtent: returns the number of mismatching boxes. */
extern "C" int orc_check_bounding_boxes(void *h)
{
    Scene *s = (Scene *)h;
    int bad = 0;
    for (size_t i = 0; i < s->nodes.size(); i++) {
        Node *n = s->nodes[i];
        if (n->type != RT_NODE_BOUNDINGBOX || !n->left) continue;
        Extent e = node_extent(n->left);
        if (e.lo.x != n->ex.lo.x || e.lo.y != n->ex.lo.y || e.lo.z != n->ex.lo.z ||
            e.hi.x != n->ex.hi.x || e.hi.y != n->ex.hi.y || e.hi.z != n->ex.hi.z) bad++;
    }
    return bad;
}

/* node extent probe (GetExtent of node index i) */
extern "C" void orc_node_extent(void *h, int index, double *out6)
{
    Scene *s = (Scene *)h;
    Extent e = node_extent(s->nodes[index]);
    out6[0] = e.lo.x; out6[1] = e.lo.y; out6[2] = e.lo.z; out6[3] = e.hi.x; out6[4] = e.hi.y; out6[5] = e.hi.z;
}

static int mat_id(const Node *m) { return m ? m->id : -9; }

/* Full hit lists for a batch of rays: psModel->Intersect(ray) followed, when transform != 0, by
   TransformNormals on every hit.  Per ray up to max_hits records:
   t, normal[3], leaf id, material id (-1 vacuum, -2 neutronium), surface id (-3 static matte, -9 none). */
extern "C" void orc_intersect(void *h, int n, const double *org, const double *dir, const double *aux /*6 per ray or null*/,
                              int max_hits, int transform, int *out_count, double *out_t, double *out_n,
                              int *out_leaf, int *out_mat, int *out_surf)
{
    Scene *s = (Scene *)h;
    Ctx cx;
    for (int i = 0; i < n; i++) {
        Ray r;
        r.org = V3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        r.dir = V3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        for (int k = 0; k < 6; k++) cx.aux[k] = aux ? aux[6 * i + k] : s->aux_init[k];
        HitList hl;
        node_hits(s->root, r, cx, hl);
        out_count[i] = (int)hl.size();
        for (int k = 0; k < max_hits; k++) {
            size_t o = (size_t)i * max_hits + k;
            if (k < (int)hl.size()) {
                Hit &hh = hl[k];
                if (transform) {
                    /* same walk as Shade.cpp:264-284 */
                    Hit copy = hh;
                    transform_normals_export(copy, cx);
                    hh = copy;
                }
                out_t[o] = hh.t;
                const V3 &nn = (transform && hh.perturbed) ? hh.np : hh.n;
                out_n[3 * o] = nn.x; out_n[3 * o + 1] = nn.y; out_n[3 * o + 2] = nn.z;
                out_leaf[o] = hh.leaf ? hh.leaf->id : -9;
                out_mat[o] = mat_id(hh.mat);
                out_surf[o] = hh.surf ? hh.surf->id : -9;
            } else {
                out_t[o] = 0.0; out_n[3 * o] = out_n[3 * o + 1] = out_n[3 * o + 2] = 0.0;
                out_leaf[o] = out_mat[o] = out_surf[o] = -9;
            }
        }
    }
}

/* Radiance along given rays (RayTracer::VisualRay) */
extern "C" void orc_radiance(void *h, int n, const double *org, const double *dir, const double *aux, float *out_spec)
{
    Scene *s = (Scene *)h;
    Ctx cx;
    for (int i = 0; i < n; i++) {
        Ray r;
        r.org = V3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        r.dir = V3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        for (int k = 0; k < 6; k++) cx.aux[k] = aux ? aux[6 * i + k] : s->aux_init[k];
        cx.depth = 0;
        Spec R = visual_ray(*s, r, cx);
        memcpy(out_spec + (size_t)NSPEC * i, R.v, sizeof(float) * NSPEC);
    }
}

extern "C" void orc_camera_emit(void *h, int n, const double *v2, const double *aux, double *org, double *dir)
{
    Scene *s = (Scene *)h;
    Ctx cx;
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 6; k++) cx.aux[k] = aux ? aux[6 * i + k] : s->aux_init[k];
        Ray r = camera_emit(s->cam, v2[2 * i], v2[2 * i + 1], cx);
        org[3 * i] = r.org.x; org[3 * i + 1] = r.org.y; org[3 * i + 2] = r.org.z;
        dir[3 * i] = r.dir.x; dir[3 * i + 1] = r.dir.y; dir[3 * i + 2] = r.dir.z;
    }
}

extern "C" void orc_noise_probe(int n, const float *p3, float *out_band, float *out_gval, float *out_grad3)
{
    for (int i = 0; i < n; i++) {
        out_band[i] = band_noise(p3 + 3 * i, 3);
        out_gval[i] = gradient_band_noise(out_grad3 + 3 * i, p3 + 3 * i, 3);
    }
}

/* TestBandNoise (Test.cpp:1635-1653) and TestVoronoi (Test.cpp:3798-3836) evaluate the scalar noises on their own: BandNoise(p, 3),
   FractalNoise(p, 3, 5, 1.0f, 1.9f, 0) and Voronoi(p) at n points */
extern "C" void orc_noise_probe2(int n, const float *p3, float *out_band, float *out_fractal5, float *out_voronoi)
{
    for (int i = 0; i < n; i++) {
        float g[3];
        out_band[i] = band_noise(p3 + 3 * i, 3);
        out_fractal5[i] = fractal_noise_export(p3 + 3 * i, 3, 5, 0);
        out_voronoi[i] = gradient_voronoi(g, p3 + 3 * i, 1);
    }
}

/* TestFresnel (Test.cpp:1467-1515): the Fresnel routines at n cosines */
extern "C" void orc_fresnel_probe(int n, const double *cos_ne, double n1, double n2, int metal_luster, double *out_F, double *out_fNT, float *out_spec10)
{
    for (int i = 0; i < n; i++) fresnel_probe(cos_ne[i], n1, n2, metal_luster, out_F + i, out_fNT + i, out_spec10 ? out_spec10 + (size_t)NSPEC * i : 0);
}

extern "C" void orc_roots_probe(const double *coef5, double *roots4, int *count, double *cbrt_in_out, int ncbrt)
{
    *count = solve_quartic(roots4, coef5[0], coef5[1], coef5[2], coef5[3], coef5[4]);
    for (int i = 0; i < ncbrt; i++) cbrt_in_out[i] = cube_root(cbrt_in_out[i]);
}

/* ------------------------------------------------------------------------------------------------
 * Image function, sampling tiles, reconstruction
 * ---------------------------------------------------------------------------------------------- */
struct Img {            /* planar float image, Image.h:134-155, index i + W*(j + H*k) (Image.h:235) */
    int w, h, c;
    std::vector<float> px;
    void alloc(int W, int H, int C) { w = W; h = H; c = C; px.assign((size_t)W * H * C, 0.0f); }
    float &at(int i, int j, int k) { return px[(size_t)i + (size_t)w * ((size_t)j + (size_t)h * k)]; }
    bool inside(int i, int j) const { return (unsigned)i < (unsigned)w && (unsigned)j < (unsigned)h; }
};

/* Image::FastSplatRGB with WRAP_CLAMP (Image.cpp:1309-1334, :59-89, Image.h:252-264, :304-323):
   out-of-range corners are not drawn. */
static void fast_splat(Img &im, Img &wt, const Rgb &rgb, double x, double y)
{
    x -= 0.5; y -= 0.5;
    double fx = floor(x), fy = floor(y);
    int i1 = (int)fx, j1 = (int)fy, i2 = i1 + 1, j2 = j1 + 1;
    double dx = x - fx, dy = y - fy;
    const double wx[2] = { 1.0 - dx, dx }, wy[2] = { 1.0 - dy, dy };
    const int ii[2] = { i1, i2 }, jj[2] = { j1, j2 };
    for (int b = 0; b < 2; b++)
        for (int a = 0; a < 2; a++) {
            if (!im.inside(ii[a], jj[b])) continue;
            /* rgb*(wx)*(wy): DisplayRGB*double rounds to float after each factor (Image.h:69) */
            float r = float(double(float(rgb.r * wx[a])) * wy[b]);
            float g = float(double(float(rgb.g * wx[a])) * wy[b]);
            float bl = float(double(float(rgb.b * wx[a])) * wy[b]);
            im.at(ii[a], jj[b], 0) += r;
            im.at(ii[a], jj[b], 1) += g;
            im.at(ii[a], jj[b], 2) += bl;
        }
    for (int b = 0; b < 2; b++)
        for (int a = 0; a < 2; a++) {
            if (!wt.inside(ii[a], jj[b])) continue;
            wt.at(ii[a], jj[b], 0) += float(wx[a] * wy[b]);
        }
}

/* Sampling.cpp:21-65 */
static double bessel_i0(double x2)
{
    double x = 0.25 * x2, u = 1.0, s = 1.0;
    for (int n = 1; u / s > 1.0e-10; n++) { u *= x / (double)(n * n); s += u; }
    return s;
}
static double kaiser_window(double x)
{
    double x2 = x * x;
    if (x2 > 1.0) return 0.0;
    return bessel_i0(4.0 * 4.0 * (1.0 - x2)) * (1.0 / 11.3019219521363);
}
static double sinc(double x)
{
    x *= kPi;
    if (x < 0.01 && x > -0.01) {
        x = x * x;
        return 1.0 + x * (-1.0 / 6.0 + x * (1.0 / 120.0 + x * (-1.0 / 5040.0 + x / 362880.0)));
    }
    return sin(x) / x;
}
static double kaiser4(double x) { return sinc(x) * kaiser_window(x / 4.0); }

static inline int lowest_index(double x) { return int(floor(x + 0.5)); }    /* Sampling.cpp:73-76 */
static inline int highest_index(double x) { return int(floor(x - 0.5)); }

/* ComputeTaps, Sampling.cpp:86-120.  Returns first-index and tap-count per destination sample too. */
extern "C" int orc_compute_taps(int n_dst, int n_src, double half_width, double *taps, int max_taps, int *first, int *count)
{
    double ratio = double(n_src) / double(n_dst), decim = (n_dst < n_src) ? ratio : 1.0;
    half_width *= decim;
    double scale = 1.0 / decim;
    int k = 0;
    for (int id = 0; id < n_dst; id++) {
        double xs = (double(id) + 0.5) * ratio;
        int i0 = lowest_index(xs - half_width), i1 = highest_index(xs + half_width);
        if (first) { first[id] = i0; count[id] = i1 - i0 + 1; }
        for (int is = i0; is <= i1; is++) {
            double x = double(is) + 0.5;
            if (k < max_taps) taps[k] = kaiser4(float((x - xs) * scale));
            k++;
        }
    }
    return k;
}

/* Resample with WRAP_CLAMP, Sampling.cpp:131-184.  (fRatio is float(nSrc)/float(nDst) there.) */
static void resample_line(float *dst, int n_dst, const float *src, int n_src, const std::vector<double> &taps)
{
    double ratio = float(n_src) / float(n_dst), decim = (n_dst < n_src) ? ratio : 1.0;
    double half = 4.0 * decim;
    const double *pf = &taps[0];
    for (int id = 0; id < n_dst; id++) {
        double xs = (double(id) + 0.5) * ratio;
        int i0 = lowest_index(xs - half), i1 = highest_index(xs + half);
        double sum = 0.0, wsum = 0.0;
        for (int is = i0; is <= i1; is++) {
            int i = is;
            if ((unsigned)is >= (unsigned)n_src) i = is < 0 ? 0 : n_src - 1;
            double y = src[i], f = *pf++;
            sum += f * y;
            wsum += f;
        }
        if (wsum) sum /= wsum;
        dst[id] = float(sum);
    }
}

/* ImageResample, Sampling.cpp:186-234 */
static void image_resample(Img &dst, int nw, int nh, const Img &src)
{
    dst.alloc(nw, nh, src.c);
    std::vector<double> tx((size_t)orc_compute_taps(nw, src.w, 4.0, 0, 0, 0, 0)), ty((size_t)orc_compute_taps(nh, src.h, 4.0, 0, 0, 0, 0));
    orc_compute_taps(nw, src.w, 4.0, &tx[0], (int)tx.size(), 0, 0);
    orc_compute_taps(nh, src.h, 4.0, &ty[0], (int)ty.size(), 0, 0);
    std::vector<float> row(nw), col(nh), tmp((size_t)src.h * nw);
    for (int k = 0; k < src.c; k++) {
        const float *ps = &src.px[(size_t)k * src.w * src.h];
        for (int j = 0; j < src.h; j++) {
            resample_line(&row[0], nw, ps + (size_t)j * src.w, src.w, tx);
            for (int i = 0; i < nw; i++) tmp[(size_t)j + (size_t)i * src.h] = row[i];
        }
        float *pd = &dst.px[(size_t)k * nw * nh];
        for (int j = 0; j < nw; j++) {
            resample_line(&col[0], nh, &tmp[(size_t)j * src.h], src.h, ty);
            for (int i = 0; i < nh; i++) pd[(size_t)j + (size_t)i * nw] = col[i];
        }
    }
}

/* Image::Normalize, Image.cpp:1238-1260 */
static void normalize_image(Img &im, const Img &wt)
{
    size_t n = (size_t)im.w * im.h;
    for (int k = 0; k < im.c; k++)
        for (size_t i = 0; i < n; i++) {
            float w = wt.px[i];
            if (w) im.px[k * n + i] /= w;
        }
}

/* float Kaiser4Filter of the splats, Image.cpp:886-918 (the resampler has its own double one, Sampling.cpp:40-65) */
static float kaiser4_filter_f(float x)
{
    static double gs_fI0Alpha = 1.0 / 11.3019219521363, gs_fAlphaSquared = 4.0 * 4.0;
    float x1, x2, xTerm, fN, fSinc, fKaiser;
    x1 = x * 1.0f / 4.0f;
    x2 = x1 * x1;
    if (x2 > 1.0) return 0.0;
    x2 = 0.25f * float(gs_fAlphaSquared) * (1.0f - x2);
    fKaiser = 1.0f + x2;
    fN = 2.0f;
    xTerm = x2;
    while (xTerm > 1.0e-4) {
        xTerm *= x2 / (fN * fN);
        fN += 1.0;
        fKaiser += xTerm;
        xTerm *= x2 / (fN * fN);
        fN += 1.0;
        fKaiser += xTerm;
    }
    x1 = 3.1415926535897932384626433832795028842f * x;
    x2 = x1 * x1;
    if (x2 < 0.0001f) fSinc = 1.0f + x2 * (-1.0f / 6.0f + x2 * 1.0f / 120.0f);
    else fSinc = sinf(x1) / x1;
    return fSinc * fKaiser * float(gs_fI0Alpha);
}
static inline int wrap_mirror(int i, int n)             /* ML_Wrap with WRAP_MIRROR, Image.cpp:59-83 */
{
    i = i % (2 * n);
    if (i < 0) i += 2 * n;
    if (i >= n) i = 2 * n - i - 1;
    return i;
}
/* Image::SplatRGB(rgb, xSample, ySample, pimWeight, fScale = 1) (Image.cpp:1203-1236) into images created with
   NewImage's default WRAP_MIRROR (Image.h:157): out-of-range indices fold back into the image. */
static void kaiser_splat(Img &im, Img &wt, const Rgb &rgb, double xSample, double ySample)
{
    double x, y, xLow, xHigh, yLow, yHigh, fWeightX, fWeight, fRecipScale, fHalfWidth, fScale = 1.0;
    int i, j, iLow, jLow;
    float rgf[512];
    fHalfWidth = 4.0f;
    xLow = xSample - fHalfWidth * fScale;
    xHigh = xSample + fHalfWidth * fScale;
    iLow = lowest_index(xLow);
    yLow = ySample - fHalfWidth * fScale;
    yHigh = ySample + fHalfWidth * fScale;
    jLow = lowest_index(yLow);
    fRecipScale = 1.0 / fScale;
    for (j = jLow, y = double(jLow) + 0.5; y <= yHigh; j++, y += 1.0) rgf[j - jLow] = kaiser4_filter_f(float((y - ySample) * fRecipScale));
    rgf[j - jLow] = 0.0;
    for (i = iLow, x = double(iLow) + 0.5; x <= xHigh; i++, x += 1.0) {
        fWeightX = kaiser4_filter_f(float((x - xSample) * fRecipScale));
        for (j = jLow, y = double(jLow) + 0.5; y <= yHigh; j++, y += 1.0) {
            fWeight = fWeightX * rgf[j - jLow];
            int ii = wrap_mirror(i, im.w), jj = wrap_mirror(j, im.h);
            im.at(ii, jj, 0) += float(rgb.r * fWeight);         /* rgb*fWeight: DisplayRGB(red*f, ...), Image.h:69 */
            im.at(ii, jj, 1) += float(rgb.g * fWeight);
            im.at(ii, jj, 2) += float(rgb.b * fWeight);
            wt.at(ii, jj, 0) += float(fWeight);
        }
    }
}

extern "C" void orc_build_quasi_index(int *out, int prime, int n);

/* BlueNoiseSamplingTile (Sampling.cpp:427-466): tile positions by accumulation, sample positions and the six
   auxiliary dimensions (DiskSamples128 through rgnQuasiIndex2/3/5) in loop order */
struct BlueNoisePlan {
    double fSqrt, fSize, ax, ay;
    std::vector<double> txs, tys;
    int q[3][128];
    BlueNoisePlan(int W, int H, double spp)
    {
        fSqrt = double(RT_BLUE_NSQRT) / sqrt(spp);
        if (W > H) { fSize = W; ax = 0.0; ay = 0.5 * (W - H); } else { fSize = H; ax = 0.5 * (H - W); ay = 0.0; }
        for (double x = 0.0; x < W; x += fSqrt) txs.push_back(x);
        for (double y = 0.0; y < H; y += fSqrt) tys.push_back(y);
        orc_build_quasi_index(q[0], 2, 128); orc_build_quasi_index(q[1], 3, 128); orc_build_quasi_index(q[2], 5, 128);
    }
    void sample(long tile, int i, double &vx, double &vy, double *aux) const
    {
        double tx = txs[tile / (long)tys.size()], ty = tys[tile % (long)tys.size()];
        vx = g_t.blue[2 * i] * fSqrt + tx; vy = g_t.blue[2 * i + 1] * fSqrt + ty;
        for (int k = 0; k < 3; k++) { aux[2 * k] = g_t.disk128[2 * q[k][i & 127]]; aux[2 * k + 1] = g_t.disk128[2 * q[k][i & 127] + 1]; }
    }
};

extern "C" long orc_bluenoise_samples(void *h, int width, int height, double spp, long first, long n, double *out8, double *out_xy)
{
    (void)h;
    BlueNoisePlan P(width, height, spp);
    long total = (long)P.txs.size() * (long)P.tys.size() * RT_BLUE_N;
    for (long idx = first; idx < first + n && idx < total && out8; idx++) {
        double vx, vy, *o = out8 + 8 * (idx - first);
        P.sample(idx / RT_BLUE_N, (int)(idx % RT_BLUE_N), vx, vy, o + 2);
        o[0] = (vx + P.ax) / P.fSize; o[1] = (vy + P.ay) / P.fSize;
        if (out_xy) { out_xy[2 * (idx - first)] = vx; out_xy[2 * (idx - first) + 1] = vy; }
    }
    return total;
}

static Rgb sample_rgb(Scene &s, double vx, double vy, Ctx &cx)      /* CastFunction, RayTracer2020.cpp:20-31 */
{
    Ray r = camera_emit(s.cam, vx, vy, cx);
    cx.depth = 0;
    Spec R = visual_ray(s, r, cx);
    return s.venera ? spectrum_vrgb(R) : spectrum_srgb(R);
}

/* Sample positions and auxiliary dimensions of FastBlueSamplingTile (Sampling.cpp:484-517) for
   samples [first, first+n) in the reference's loop order (tile x outer, tile y inner, i innermost).
   out8 = normalised vx, vy, then the six g_u.. / g_v.. values; out_xy = supersample-image position. */
extern "C" long orc_fastblue_samples(void *h, int width, int height, double spp, long first, long n, double *out8, double *out_xy)
{
    Scene *s = (Scene *)h;
    int nW = int(width * sqrt(spp) + 0.5), nH = int(height * sqrt(spp) + 0.5);
    double fSize, ax, ay;
    if (nW > nH) { fSize = nW; ax = 0.0; ay = 0.5 * (nW - nH); } else { fSize = nH; ax = 0.5 * (nH - nW); ay = 0.0; }
    long idx = 0, total = 0;
    for (double tx = 0.0; tx < nW; tx += RT_BLUE_NSQRT)
        for (double ty = 0.0; ty < nH; ty += RT_BLUE_NSQRT)
            for (int i = 0; i < RT_BLUE_N; i++, idx++) {
                total++;
                if (idx < first || idx >= first + n || !out8) continue;
                double vx = g_t.blue[2 * i] * RT_BLUE_NSQRT + tx, vy = g_t.blue[2 * i + 1] * RT_BLUE_NSQRT + ty;
                double *o = out8 + 8 * (idx - first);
                o[0] = (vx + ax) / fSize; o[1] = (vy + ay) / fSize;
                o[2] = 2.0 * g_t.blue[2 * s->qi3[i % RT_BLUE_N]] - 1.0;
                o[3] = 2.0 * g_t.blue[2 * s->qi3[i % RT_BLUE_N] + 1] - 1.0;
                o[4] = 2.0 * g_t.blue[2 * s->qi2[i % RT_BLUE_N]] - 1.0;
                o[5] = 2.0 * g_t.blue[2 * s->qi2[i % RT_BLUE_N] + 1] - 1.0;
                o[6] = o[4]; o[7] = o[5];
                if (out_xy) { out_xy[2 * (idx - first)] = vx; out_xy[2 * (idx - first) + 1] = vy; }
            }
    return total;
}

/* JitterSamplingTile sample positions (Sampling.cpp:404-408) in loop order: out2 = (vx, vy) normalised */
extern "C" long orc_jitter_samples(void *h, int width, int height, double spp, long first, long n, double *out2)
{
    Scene *s = (Scene *)h;
    int nn = int(width * sqrt(spp) + 0.5), nHi = int(height * sqrt(spp) + 0.5);
    uint32_t px = s->prng_x, pc = s->prng_c;
    auto draw = [&]() -> double {
        uint64_t a = (uint64_t)1965537969u * px + pc; px = (uint32_t)a; pc = (uint32_t)(a >> 32); uint32_t hi = px;
        a = (uint64_t)1965537969u * px + pc; px = (uint32_t)a; pc = (uint32_t)(a >> 32); uint32_t lo = px;
        return (double(hi) + double(lo) * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0);
    };
    long idx = 0;
    for (int j = 0; j < nHi; j++)
        for (int i = 0; i < nn; i++, idx++) {
            double vy = (double(j) + (nn - nHi) / 2 + draw()) / double(nn);
            double vx = (double(i) + draw()) / double(nn);
            if (idx >= first && idx < first + n && out2) { out2[2 * (idx - first)] = vx; out2[2 * (idx - first) + 1] = vy; }
        }
    return idx;
}

/* RayTracer::Render (RayTracer2020.cpp:33-38) through TestSamplingTile (Sampling.cpp:324-348) or
   FastBlueSamplingTile (Sampling.cpp:474-533).  planes: float[3][H][W].  sample_rgb (optional):
   per-sample RGB in loop order.  threads > 1 evaluates samples in parallel (the image function is
   pure once the globals are per-sample) and then splats sequentially in the reference's order. */
extern "C" int orc_render(void *h, const RtRenderDesc *rd, float *planes, float *sample_rgb_out, RtStats *st, int threads)
{
    Scene *s = (Scene *)h;
    auto t0 = std::chrono::steady_clock::now();
    int W = rd->width, H = rd->height;
    long ns = 0, np = 0, nsh = 0, nsec = 0, refined = 0;
    if (threads < 1) threads = 1;
    if (s->sampler_type == RT_SAMPLER_TEST) {
        int n = W, nHi = H;
        std::vector<Rgb> rows((size_t)W * H);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4) reduction(+:np,nsh,nsec)
        for (int j = 0; j < nHi; j++) {
            Ctx cx;
            for (int k = 0; k < 6; k++) cx.aux[k] = s->aux_init[k];
            double vy = (double(j) + (n - nHi) / 2 + 0.5) / double(n);
            for (int i = 0; i < n; i++) {
                double vx = (double(i) + 0.5) / double(n);
                rows[(size_t)j * W + i] = sample_rgb(*s, vx, vy, cx);
            }
            np += cx.n_primary; nsh += cx.n_shadow; nsec += cx.n_secondary;
        }
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++) {
                const Rgb &c = rows[(size_t)j * W + i];
                planes[(size_t)i + (size_t)W * (j + (size_t)H * 0)] = c.r;
                planes[(size_t)i + (size_t)W * (j + (size_t)H * 1)] = c.g;
                planes[(size_t)i + (size_t)W * (j + (size_t)H * 2)] = c.b;
                if (sample_rgb_out) { float *o = sample_rgb_out + 3 * ((size_t)j * W + i); o[0] = c.r; o[1] = c.g; o[2] = c.b; }
            }
        ns = (long)W * H;
    } else if (s->sampler_type == RT_SAMPLER_DEBUG) {
        /* DebugSamplingTile::Render, Sampling.cpp:356-372: one (verbose, in the reference) ray through the centre fills the image */
        Ctx cx;
        for (int k = 0; k < 6; k++) cx.aux[k] = s->aux_init[k];
        Rgb c = sample_rgb(*s, 0.5, 0.5, cx);
        np += cx.n_primary; nsh += cx.n_shadow; nsec += cx.n_secondary;
        for (size_t i = 0; i < (size_t)W * H; i++) { planes[i] = c.r; planes[(size_t)W * H + i] = c.g; planes[2 * (size_t)W * H + i] = c.b; }
        if (sample_rgb_out) { sample_rgb_out[0] = c.r; sample_rgb_out[1] = c.g; sample_rgb_out[2] = c.b; }
        ns = 1;
    } else if (s->sampler_type == RT_SAMPLER_JITTER) {
        /* JitterSamplingTile::Render, Sampling.cpp:394-419: one sample per supersample pixel at a random position drawn
           from the global MWC stream (y before x, RandomDouble = two RandomUnsigned, high word first), SetRGB, then
           ImageResample without weights.  The draws are made serially in the reference's order; only the radiance
           evaluation is spread over threads. */
        double spp = rd->spp;
        int n = int(W * sqrt(spp) + 0.5), nHi = int(H * sqrt(spp) + 0.5);
        std::vector<double> vxs((size_t)n * nHi), vys((size_t)n * nHi);
        uint32_t px = s->prng_x, pc = s->prng_c;
        auto draw = [&]() -> double {
            uint64_t a = (uint64_t)1965537969u * px + pc; px = (uint32_t)a; pc = (uint32_t)(a >> 32); uint32_t hi = px;
            a = (uint64_t)1965537969u * px + pc; px = (uint32_t)a; pc = (uint32_t)(a >> 32); uint32_t lo = px;
            return (double(hi) + double(lo) * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0);
        };
        for (int j = 0; j < nHi; j++)
            for (int i = 0; i < n; i++) {
                vys[(size_t)j * n + i] = (double(j) + (n - nHi) / 2 + draw()) / double(n);
                vxs[(size_t)j * n + i] = (double(i) + draw()) / double(n);
            }
        Img im2;
        im2.alloc(n, nHi, 3);
        std::vector<Rgb> rgb((size_t)n * nHi);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4) reduction(+:np,nsh,nsec)
        for (int j = 0; j < nHi; j++) {
            Ctx cx;
            for (int k = 0; k < 6; k++) cx.aux[k] = s->aux_init[k];
            for (int i = 0; i < n; i++) rgb[(size_t)j * n + i] = sample_rgb(*s, vxs[(size_t)j * n + i], vys[(size_t)j * n + i], cx);
            np += cx.n_primary; nsh += cx.n_shadow; nsec += cx.n_secondary;
        }
        for (int j = 0; j < nHi; j++)
            for (int i = 0; i < n; i++) {
                const Rgb &c = rgb[(size_t)j * n + i];
                im2.px[(size_t)i + (size_t)n * (j + (size_t)nHi * 0)] = c.r;
                im2.px[(size_t)i + (size_t)n * (j + (size_t)nHi * 1)] = c.g;
                im2.px[(size_t)i + (size_t)n * (j + (size_t)nHi * 2)] = c.b;
            }
        ns = (long)n * nHi;
        if (sample_rgb_out) memcpy(sample_rgb_out, &rgb[0], sizeof(Rgb) * rgb.size());
        Img im;
        image_resample(im, W, H, im2);
        memcpy(planes, &im.px[0], sizeof(float) * im.px.size());
    } else if (s->sampler_type == RT_SAMPLER_BLUENOISE) {
        /* BlueNoiseSamplingTile::Render, Sampling.cpp:427-466: radiance evaluation spread over threads, splats serial in
           the reference's order */
        BlueNoisePlan P(W, H, rd->spp);
        long ntile = (long)P.txs.size() * (long)P.tys.size();
        std::vector<Rgb> rgb((size_t)ntile * RT_BLUE_N);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+:np,nsh,nsec)
        for (long tile = 0; tile < ntile; tile++) {
            Ctx cx;
            for (int i = 0; i < RT_BLUE_N; i++) {
                double vx, vy;
                P.sample(tile, i, vx, vy, cx.aux);
                rgb[(size_t)tile * RT_BLUE_N + i] = sample_rgb(*s, (vx + P.ax) / P.fSize, (vy + P.ay) / P.fSize, cx);
            }
            np += cx.n_primary; nsh += cx.n_shadow; nsec += cx.n_secondary;
        }
        Img im, wt;
        im.alloc(W, H, 3);
        wt.alloc(W, H, 1);
        for (long tile = 0; tile < ntile; tile++)
            for (int i = 0; i < RT_BLUE_N; i++) {
                double vx, vy, aux[6];
                P.sample(tile, i, vx, vy, aux);
                kaiser_splat(im, wt, rgb[(size_t)tile * RT_BLUE_N + i], vx, vy);
            }
        ns = ntile * RT_BLUE_N;
        if (sample_rgb_out) memcpy(sample_rgb_out, &rgb[0], sizeof(Rgb) * rgb.size());
        normalize_image(im, wt);
        memcpy(planes, &im.px[0], sizeof(float) * im.px.size());
    } else {
        const bool adaptive = s->sampler_type == RT_SAMPLER_ADAPTIVEBLUE;
        double spp = rd->spp;
        int nW = int(W * sqrt(spp) + 0.5), nH = int(H * sqrt(spp) + 0.5);
        Img im2, wt2;
        im2.alloc(nW, nH, 3);
        wt2.alloc(nW, nH, 1);
        double fSize, ax, ay;
        if (nW > nH) { fSize = nW; ax = 0.0; ay = 0.5 * (nW - nH); } else { fSize = nH; ax = 0.5 * (nH - nW); ay = 0.0; }
        std::vector<double> txs, tys;
        for (double tx = 0.0; tx < nW; tx += RT_BLUE_NSQRT) txs.push_back(tx);
        for (double ty = 0.0; ty < nH; ty += RT_BLUE_NSQRT) tys.push_back(ty);
        long ntile = (long)txs.size() * (long)tys.size();
        std::vector<Rgb> rgb((size_t)ntile * RT_BLUE_N);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+:np,nsh,nsec)
        for (long tile = 0; tile < ntile; tile++) {
            Ctx cx;
            double tx = txs[tile / (long)tys.size()], ty = tys[tile % (long)tys.size()];
            for (int i = 0; i < RT_BLUE_N; i++) {
                double vx = g_t.blue[2 * i] * RT_BLUE_NSQRT + tx, vy = g_t.blue[2 * i + 1] * RT_BLUE_NSQRT + ty;
                cx.aux[0] = 2.0 * g_t.blue[2 * s->qi3[i % RT_BLUE_N]] - 1.0;
                cx.aux[1] = 2.0 * g_t.blue[2 * s->qi3[i % RT_BLUE_N] + 1] - 1.0;
                cx.aux[2] = 2.0 * g_t.blue[2 * s->qi2[i % RT_BLUE_N]] - 1.0;
                cx.aux[3] = 2.0 * g_t.blue[2 * s->qi2[i % RT_BLUE_N] + 1] - 1.0;
                cx.aux[4] = cx.aux[2];
                cx.aux[5] = cx.aux[3];
                rgb[(size_t)tile * RT_BLUE_N + i] = sample_rgb(*s, (vx + ax) / fSize, (vy + ay) / fSize, cx);
            }
            np += cx.n_primary; nsh += cx.n_shadow; nsec += cx.n_secondary;
        }
        for (long tile = 0; tile < ntile; tile++) {
            double tx = txs[tile / (long)tys.size()], ty = tys[tile % (long)tys.size()];
            for (int i = 0; i < RT_BLUE_N; i++) {
                double vx = g_t.blue[2 * i] * RT_BLUE_NSQRT + tx, vy = g_t.blue[2 * i + 1] * RT_BLUE_NSQRT + ty;
                fast_splat(im2, wt2, rgb[(size_t)tile * RT_BLUE_N + i], vx, vy);
            }
        }
        ns = ntile * RT_BLUE_N;
        if (sample_rgb_out) memcpy(sample_rgb_out, &rgb[0], sizeof(Rgb) * rgb.size());
        if (adaptive) {
            /* AdaptiveBlueSamplingTile::Render, Sampling.cpp:619-679 (the base pass above is its lines 592-617; the two
               diagnostic BMP files it writes on the way are not produced) */
            Img det;
            det.alloc(nW, nH, 1);                       /* WRAP_MIRROR, Sampling.cpp:580 */
            for (int j = 0; j < nH; j++)
                for (int i = 0; i < nW; i++) {
                    double f00 = 0.212649390101432f * im2.at(i, j, 0) + 0.715169131755828f * im2.at(i, j, 1) + 0.0721815153956413f * im2.at(i, j, 2);
                    det.at(i, j, 0) = float(f00);
                }
            normalize_image(det, wt2);
            auto get = [&](int i, int j) -> double { return det.at(wrap_mirror(i, nW), wrap_mirror(j, nH), 0); };
            for (int j = 0; j < nH; j++)                /* in place, like the reference */
                for (int i = 0; i < nW; i++) {
                    double f00 = get(i + 0, j + 0), f01 = get(i + 0, j + 1), f10 = get(i + 1, j + 0), f11 = get(i + 1, j + 1);
                    double g1 = f00 - f11, g2 = f01 - f10, sum = f00 + f01 + f10 + f11, gv;
                    if (sum > 0.0) gv = sqrt(g1 * g1 + g2 * g2) / sum * 1.0; else gv = 0.0;
                    det.at(i, j, 0) = float(gv);
                }
            const int B = RT_BLUE_NSQRT / 2, HYST = 2;
            std::vector<std::pair<int, int> > blocks;
            for (int j = 0; j < nH; j += B)
                for (int i = 0; i < nW; i += B) {
                    double maxg = 0;
                    for (int is = i - HYST; is < i + B + HYST; is++)
                        for (int js = j - HYST; js < j + B + HYST; js++) {
                            double gv = get(is, js);
                            if (gv > maxg) maxg = gv;
                        }
                    if (maxg > 0.30) blocks.push_back(std::make_pair(i, j));
                }
            refined = (long)blocks.size();
            std::vector<Rgb> rgb2(blocks.size() * RT_BLUE_N);
            auto pos2 = [&](size_t b, int k, double &vx, double &vy) {
                vx = g_t.blue[2 * k] * RT_BLUE_NSQRT / 2 + double(blocks[b].first);
                vy = g_t.blue[2 * k + 1] * RT_BLUE_NSQRT / 2 + double(blocks[b].second);
            };
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+:np,nsh,nsec)
            for (long b = 0; b < (long)blocks.size(); b++) {
                Ctx cx;
                for (int k = 0; k < RT_BLUE_N; k++) {
                    double vx, vy;
                    pos2((size_t)b, k, vx, vy);
                    cx.aux[0] = 2.0 * g_t.blue[2 * s->qi3[k % RT_BLUE_N]] - 1.0;
                    cx.aux[1] = 2.0 * g_t.blue[2 * s->qi3[k % RT_BLUE_N] + 1] - 1.0;
                    cx.aux[2] = 2.0 * g_t.blue[2 * s->qi2[k % RT_BLUE_N]] - 1.0;
                    cx.aux[3] = 2.0 * g_t.blue[2 * s->qi2[k % RT_BLUE_N] + 1] - 1.0;
                    cx.aux[4] = cx.aux[2];
                    cx.aux[5] = cx.aux[3];
                    rgb2[(size_t)b * RT_BLUE_N + k] = sample_rgb(*s, (vx + ax) / fSize, (vy + ay) / fSize, cx);
                }
                np += cx.n_primary; nsh += cx.n_shadow; nsec += cx.n_secondary;
            }
            for (size_t b = 0; b < blocks.size(); b++)
                for (int k = 0; k < RT_BLUE_N; k++) {
                    double vx, vy;
                    pos2(b, k, vx, vy);
                    fast_splat(im2, wt2, rgb2[b * RT_BLUE_N + k], vx, vy);
                }
        }
        if (adaptive && spp == 1.0) {                   /* Sampling.cpp:687-692 */
            normalize_image(im2, wt2);
            memcpy(planes, &im2.px[0], sizeof(float) * im2.px.size());
        } else {
            Img im, wt;
            image_resample(im, W, H, im2);
            image_resample(wt, W, H, wt2);
            normalize_image(im, wt);
            memcpy(planes, &im.px[0], sizeof(float) * im.px.size());
        }
    }
    if (st) {
        memset(st, 0, sizeof *st);
        st->samples = ns; st->rays_primary = np; st->rays_shadow = nsh; st->rays_secondary = nsec; st->refined_blocks = refined;
        st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return (int)ns;
}

/* ------------------------------------------------------------------------------------------------
 * Output stage: Image::WriteBMP (Image.cpp:416-474) -- SetGamma (Image.cpp:498-517), ClampToGamut
 * (Image.cpp:293-327), GammaEncode (Image.h:206-220), WriteHeaderBMP (Image.cpp:330-414).  TEST INFRASTRUCTURE.
 * Writes the whole 24-bit file image (54-byte header + bottom-up BGR lines padded to 4 bytes; the reference leaves
 * the padding bytes uninitialised, here they are 0).  Returns the byte count, or the required size when out == 0. */
static void orc_set_gamma(double gamma, float *T, float *A)
{
    for (int n = 0; n < 256; n++) {
        if (gamma == -1.0) {                         /* GAMMA_SRGB, Image.h:21 */
            double f = (double)n * (1.0 / 255.0);
            if (f <= 0.03928) f = f / 12.92;
            else f = (float)pow((f + 0.055) / 1.055, 2.4);
            T[n] = (float)f;
        } else
            T[n] = (float)pow((double)n * (1.0 / 255.0), gamma);
    }
    A[0] = 0.0f;
    for (int n = 1; n < 256; n++) A[n] = (T[n - 1] + T[n]) / 2;
}
static int orc_gamma_encode(const float *A, double f)
{
    int n = 0;
    if (f > A[128]) n = 128;
    if (f > A[n + 64]) n = n + 64;
    if (f > A[n + 32]) n = n + 32;
    if (f > A[n + 16]) n = n + 16;
    if (f > A[n + 8]) n = n + 8;
    if (f > A[n + 4]) n = n + 4;
    if (f > A[n + 2]) n = n + 2;
    if (f > A[n + 1]) n = n + 1;
    return n;
}
static void orc_clamp_to_gamut(float &red, float &grn, float &blu)
{
    float fLum = 0.2126f * red + 0.7152f * grn + 0.0722f * blu, t;      /* Luminance(), Image.h:110-115 */
    if (fLum < 0.0) { red = grn = blu = 0.0f; return; }
    if (fLum > 1.0) { red = grn = blu = 1.0f; return; }
    t = 1.0;
    if (red < 0.0 && -fLum > (red - fLum) * t) t = -fLum / (red - fLum);
    if (red > 1.0 && 1.0 - fLum < (red - fLum) * t) t = (1.0f - fLum) / (red - fLum);
    if (grn < 0.0 && -fLum > (grn - fLum) * t) t = -fLum / (grn - fLum);
    if (grn > 1.0 && 1.0 - fLum < (grn - fLum) * t) t = (1.0f - fLum) / (grn - fLum);
    if (blu < 0.0 && -fLum > (blu - fLum) * t) t = -fLum / (blu - fLum);
    if (blu > 1.0 && 1.0 - fLum < (blu - fLum) * t) t = (1.0f - fLum) / (blu - fLum);
    float r = fLum + t * (red - fLum), g = fLum + t * (grn - fLum), b = fLum + t * (blu - fLum);
    red = r; grn = g; blu = b;
}
extern "C" long orc_bmp_encode(const float *planes, int width, int height, double gamma, unsigned char *out, long cap)
{
    long line = 4 * ((24L * width + 31) >> 5), total = 54 + line * height;
    if (!out) return total;
    if (cap < total) return -1;
    memset(out, 0, (size_t)total);
    out[0] = 'B'; out[1] = 'M';
    uint32_t h32[13];
    h32[0] = (uint32_t)total;           /* bfSize */
    h32[1] = 0;                         /* bfReserved1/2 */
    h32[2] = 54;                        /* bfOffBits */
    h32[3] = 40; h32[4] = (uint32_t)width; h32[5] = (uint32_t)height;
    h32[6] = 1u | (24u << 16);          /* biPlanes, biBitCount */
    h32[7] = 0; h32[8] = 0;
    h32[9] = h32[10] = (uint32_t)(int)(300.0 / 0.0254 + 0.5);
    h32[11] = 0; h32[12] = 0;
    memcpy(out + 2, h32, sizeof h32);
    float T[256], A[256];
    orc_set_gamma(gamma, T, A);
    const size_t plane = (size_t)width * height;
    for (int j = height - 1, row = 0; j >= 0; --j, ++row) {
        unsigned char *p = out + 54 + (size_t)row * line;
        for (int i = 0; i < width; i++) {
            float r = planes[(size_t)j * width + i], g = planes[plane + (size_t)j * width + i], b = planes[2 * plane + (size_t)j * width + i];
            if (r < 0.0 || r > 1.0 || g < 0.0 || g > 1.0 || b < 0.0 || b > 1.0) orc_clamp_to_gamut(r, g, b);
            p[3 * i + 0] = (unsigned char)orc_gamma_encode(A, b);
            p[3 * i + 1] = (unsigned char)orc_gamma_encode(A, g);
            p[3 * i + 2] = (unsigned char)orc_gamma_encode(A, r);
        }
    }
    return total;
}

/* Video::WriteFrame (Video.cpp:49-80): "FRAME\n" + Y, Cb, Cr planes (C444).  TEST INFRASTRUCTURE.
 * pow(float, float) resolves to the float overload under MSVC's <math.h> in C++; powf here. */
extern "C" long orc_y4m_encode_frame(const float *planes, int width, int height, float gamma, unsigned char *out, long cap)
{
    long total = 6 + 3L * width * height;
    if (!out) return total;
    if (cap < total) return -1;
    memcpy(out, "FRAME\n", 6);
    unsigned char *buf = out + 6;
    const size_t plane = (size_t)width * height;
    for (int j = 0; j < height; j++)
        for (int i = 0; i < width; i++) {
            size_t o = (size_t)j * width + i;
            float r = planes[o], g = planes[plane + o], b = planes[2 * plane + o];
            if (r < 0.0 || r > 1.0 || g < 0.0 || g > 1.0 || b < 0.0 || b > 1.0) orc_clamp_to_gamut(r, g, b);
            r = powf(r, 1.0f / gamma); g = powf(g, 1.0f / gamma); b = powf(b, 1.0f / gamma);
            double Y  =  16 +  65.481 * r + 128.553 * g +  24.966 * b;
            double Cb = 128 -  37.797 * r -  74.203 * g + 112.000 * b;
            double Cr = 128 + 112.000 * r -  93.786 * g -  18.214 * b;
            buf[o] = (unsigned char)(int)Y;
            buf[plane + o] = (unsigned char)(int)Cb;
            buf[2 * plane + o] = (unsigned char)(int)Cr;
        }
    return total;
}
