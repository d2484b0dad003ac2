/*
 * rt_kernels.hip -- the wavefront pipeline behind RayTracer::Render (RayTracer2020.cpp:33-38) on gfx950.
 *
 * The reference evaluates one sample at a time through mutually recursive VisualRay / LightRay calls
 * (Shade.cpp:301-520).  Here a chunk of samples advances level by level through queues in HBM:
 *
 *   k_generate   sample index -> sample position + auxiliary dimensions (Sampling.cpp:335-346, 498-517)
 *                -> camera ray (Cameras.h) -> root frame
 *   k_visual     one visual ray per lane: CSG intersection (stack machine, rt_core.hip.h), walk of the hit
 *                list through transparent media up to the first boundary (Shade.cpp:433-511), ambient
 *                term; appends one shadow ray per light that can see the surface to the shadow queue
 *   k_shadow     one shadow ray per lane: intersection + transmission walk (Shade.cpp:301-359)
 *   k_shade      per frame: sums the lights in the reference's order (Shade.cpp:477-489), then appends
 *                the reflection / refraction child frames (Shade.cpp:490-501) to the next level's queue
 *   k_resolve    deepest level first: R = ((R + refl*w1) + refr*w2) * tr (+ sterisent), result handed
 *                to the parent frame or, at level 0, converted to RGB (RadiometryColor.h:62-66)
 *
 * Queue appends are compacted per wavefront: a 64-lane prefix sum over the lanes' counts and ONE atomic
 * per wave.  All frame / shadow-ray state is SoA ([field][index]) so that consecutive lanes touch
 * consecutive addresses.
 */
#include <hipcub/hipcub.hpp>
#include "rt_core.hip.h"
#include "rt_pipeline.h"

/* ------------------------------------------------------------------ wave-level queue allocation */
__device__ int wave_alloc(int count, int *counter)
{
    const int lane = threadIdx.x & 63;
    int incl = count;
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(counter, total);
    base = __shfl(base, 63, 64);
    return base + incl - count;
}

/* two queues at once: counts a (low half) and b (high half) of one 64-bit counter, one atomic per wave */
__device__ void wave_alloc2(int count_a, int count_b, int *counter_pair, int &base_a, int &base_b)
{
    const int lane = threadIdx.x & 63;
    unsigned long long incl = (unsigned long long)(unsigned)count_a | ((unsigned long long)(unsigned)count_b << 32);
    const unsigned long long own = incl;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const unsigned long long total = __shfl(incl, 63, 64);
    unsigned long long base = 0;
    if (lane == 63 && total != 0) base = atomicAdd((unsigned long long *)counter_pair, total);
    base = __shfl(base, 63, 64);
    const unsigned long long mine = base + incl - own;      /* no carry between the halves: both queues hold < 2^31 entries */
    base_a = (int)(unsigned)(mine & 0xFFFFFFFFull);
    base_b = (int)(unsigned)(mine >> 32);
}

/* Frame and queue arrays are streamed: written once by one kernel, read once by the next.  Non-temporal accesses
   keep them from displacing the per-lane private memory (hit arenas) that the tracing kernels re-use from L2. */
template <typename T> RT_DEV void nt_store(T *p, T v) { __builtin_nontemporal_store(v, p); }
template <typename T> RT_DEV T nt_load(const T *p) { return __builtin_nontemporal_load(p); }
RT_DEV void spec_store(float *base, size_t cap, size_t idx, const Spec &s) { for (int k = 0; k < RT_NSPEC; k++) nt_store(&base[(size_t)k * cap + idx], s.v[k]); }
RT_DEV Spec spec_fetch(const float *base, size_t cap, size_t idx) { Spec s; for (int k = 0; k < RT_NSPEC; k++) s.v[k] = nt_load(&base[(size_t)k * cap + idx]); return s; }
RT_DEV void d3_store(double *base, size_t cap, int field, size_t idx, const D3 &v) { nt_store(&base[(size_t)field * cap + idx], v.x); nt_store(&base[(size_t)(field + 1) * cap + idx], v.y); nt_store(&base[(size_t)(field + 2) * cap + idx], v.z); }
RT_DEV D3 d3_fetch(const double *base, size_t cap, int field, size_t idx) { return d3(nt_load(&base[(size_t)field * cap + idx]), nt_load(&base[(size_t)(field + 1) * cap + idx]), nt_load(&base[(size_t)(field + 2) * cap + idx])); }

#define RT_ADAPT_BLOCK 20       /* BLUE_NSQRT/NSQRT1, Sampling.cpp:649 */
#define RT_ADAPT_HYST  2        /* HYSTERESIS */

/* ------------------------------------------------------------------ sample generation + camera */
__device__ void sample_at(const RtDevScene &S, const RtSampleGrid &g, int64_t s, double *pos2, double *v2, double *aux)
{
    if (g.sampler == RT_SAMPLER_TEST) {                 /* TestSamplingTile::Render, Sampling.cpp:324-348 */
        int j = g.test_row0 + (int)(s / g.width), i = (int)(s % g.width);
        int n = g.width, nHi = g.height;
        v2[1] = ((double)j + (double)((n - nHi) / 2) + 0.5) / (double)n;
        v2[0] = ((double)i + 0.5) / (double)n;
        pos2[0] = (double)i + 0.5; pos2[1] = (double)j + 0.5;
        for (int k = 0; k < 6; k++) aux[k] = S.aux_init[k];
        return;
    }
    if (g.sampler == RT_SAMPLER_DEBUG) {                /* DebugSamplingTile::Render, Sampling.cpp:356-372: one ray through the image centre */
        v2[0] = 0.5; v2[1] = 0.5; pos2[0] = 0.5 * g.width; pos2[1] = 0.5 * g.height;
        for (int k = 0; k < 6; k++) aux[k] = S.aux_init[k];
        return;
    }
    if (g.sampler == RT_SAMPLER_JITTER) {               /* JitterSamplingTile::Render, Sampling.cpp:404-413 */
        const int n = g.nW, nHi = g.nH;
        const int jl = (int)(s / n), i = (int)(s - (int64_t)jl * n), j = g.jit_row0 + jl;
        /* the reference draws four RandomUnsigned per sample from one global stream (MonteCarlo.h:17-40), row by
           row; the host provides the stream's state at every 64th pixel (rt_random_skip arithmetic) and the lane
           steps from there */
        const uint32_t *stp = g.jit_states + 2 * ((size_t)jl * g.jit_segs + (i >> 6));
        uint32_t x = stp[0], c = stp[1], u[4];
        for (int k = 4 * (i & 63); k > 0; --k) { uint64_t a = (uint64_t)1965537969u * x + c; x = (uint32_t)a; c = (uint32_t)(a >> 32); }
        for (int k = 0; k < 4; k++) { uint64_t a = (uint64_t)1965537969u * x + c; x = (uint32_t)a; c = (uint32_t)(a >> 32); u[k] = x; }
        const double ry = ((double)u[0] + (double)u[1] * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0);
        const double rx = ((double)u[2] + (double)u[3] * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0);
        v2[1] = ((double)j + (double)((n - nHi) / 2) + ry) / (double)n;
        v2[0] = ((double)i + rx) / (double)n;
        pos2[0] = (double)i; pos2[1] = (double)j;
        for (int k = 0; k < 6; k++) aux[k] = S.aux_init[k];
        return;
    }
    if (g.sampler == RT_SAMPLER_BLUENOISE) {            /* BlueNoiseSamplingTile::Render, Sampling.cpp:448-458 */
        const uint32_t tile = (uint32_t)(s / RT_BLUE_N);
        const int i = (int)(s - (int64_t)tile * RT_BLUE_N);
        const uint32_t txu = tile / (uint32_t)g.tiles_y_local;
        const int tx = (int)txu, ty = (int)(tile - txu * (uint32_t)g.tiles_y_local) + g.tile_y0;
        const double vx = S.blue[2 * i] * g.bn_sqrt + g.bn_tx[tx], vy = S.blue[2 * i + 1] * g.bn_sqrt + g.bn_ty[ty];
        pos2[0] = vx; pos2[1] = vy;
        v2[0] = (vx + g.adjust_x) / g.size; v2[1] = (vy + g.adjust_y) / g.size;
        const int q2 = S.q128[i & 127], q3 = S.q128[128 + (i & 127)], q5 = S.q128[256 + (i & 127)];
        aux[0] = S.disk128[2 * q2]; aux[1] = S.disk128[2 * q2 + 1];
        aux[2] = S.disk128[2 * q3]; aux[3] = S.disk128[2 * q3 + 1];
        aux[4] = S.disk128[2 * q5]; aux[5] = S.disk128[2 * q5 + 1];
        return;
    }
    if (g.sampler == RT_SAMPLER_ADAPTIVEBLUE && g.ad_list) {    /* refinement pass, Sampling.cpp:660-669 */
        const int blk = g.ad_list[s / RT_BLUE_N], k = (int)(s % RT_BLUE_N);
        const double vtx = (double)((blk % g.ad_bx) * RT_ADAPT_BLOCK), vty = (double)((blk / g.ad_bx) * RT_ADAPT_BLOCK);
        /* v = g_rgvBlueNoise[k]*BLUE_NSQRT/NSQRT1 + vTile */
        const double vx = S.blue[2 * k] * (double)RT_BLUE_NSQRT / 2.0 + vtx, vy = S.blue[2 * k + 1] * (double)RT_BLUE_NSQRT / 2.0 + vty;
        pos2[0] = vx; pos2[1] = vy;
        v2[0] = (vx + g.adjust_x) / g.size; v2[1] = (vy + g.adjust_y) / g.size;
        const int q3 = S.qi3[k % RT_BLUE_N], q2 = S.qi2[k % RT_BLUE_N];
        aux[0] = 2.0 * S.blue[2 * q3] - 1.0;
        aux[1] = 2.0 * S.blue[2 * q3 + 1] - 1.0;
        aux[2] = 2.0 * S.blue[2 * q2] - 1.0;
        aux[3] = 2.0 * S.blue[2 * q2 + 1] - 1.0;
        aux[4] = aux[2];
        aux[5] = aux[3];
        return;
    }
    /* FastBlueSamplingTile::Render, Sampling.cpp:498-517 (and the base pass of the adaptive tile, :592-617): tile x outer,
       tile y inner, i innermost */
    const uint32_t tile = (uint32_t)(s / RT_BLUE_N);      /* < 2^31 tiles (checked by the host) */
    int i = (int)(s - (int64_t)tile * RT_BLUE_N);
    const uint32_t txu = tile / (uint32_t)g.tiles_y_local;
    int tx = (int)txu, ty = (int)(tile - txu * (uint32_t)g.tiles_y_local) + g.tile_y0;
    double vtx = (double)tx * (double)RT_BLUE_NSQRT, vty = (double)ty * (double)RT_BLUE_NSQRT;
    double vx = S.blue[2 * i] * (double)RT_BLUE_NSQRT + vtx, vy = S.blue[2 * i + 1] * (double)RT_BLUE_NSQRT + vty;
    pos2[0] = vx; pos2[1] = vy;
    v2[0] = (vx + g.adjust_x) / g.size; v2[1] = (vy + g.adjust_y) / g.size;
    int q3 = S.qi3[i % RT_BLUE_N], q2 = S.qi2[i % RT_BLUE_N];
    aux[0] = 2.0 * S.blue[2 * q3] - 1.0;
    aux[1] = 2.0 * S.blue[2 * q3 + 1] - 1.0;
    aux[2] = 2.0 * S.blue[2 * q2] - 1.0;
    aux[3] = 2.0 * S.blue[2 * q2 + 1] - 1.0;
    aux[4] = aux[2];
    aux[5] = aux[3];
}

__global__ void k_probe_samples(const RtDevScene Sv, RtSampleGrid g, int64_t first, int64_t n, double *out8, double *out_xy)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double pos[2], v[2], aux[6];
    sample_at(Sv, g, first + k, pos, v, aux);
    out8[8 * k] = v[0]; out8[8 * k + 1] = v[1];
    for (int i = 0; i < 6; i++) out8[8 * k + 2 + i] = aux[i];
    out_xy[2 * k] = pos[0]; out_xy[2 * k + 1] = pos[1];
}

__global__ void k_probe_camera(const RtDevScene Sv, int n, const double *v2, const double *aux6, double *org, double *dir)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const RtDevScene &S = Sv;
    double uC = aux6 ? aux6[6 * k] : S.aux_init[0], vC = aux6 ? aux6[6 * k + 1] : S.aux_init[1];
    DRay r = camera_emit(S.cam, v2[2 * k], v2[2 * k + 1], uC, vC);
    org[3 * k] = r.org.x; org[3 * k + 1] = r.org.y; org[3 * k + 2] = r.org.z;
    dir[3 * k] = r.dir.x; dir[3 * k + 1] = r.dir.y; dir[3 * k + 2] = r.dir.z;
}

/* root ray k of a chunk: sample first + k of the grid, or user ray k (probes) */
/* sample first + k of the grid -> its tile and its index in the tile.  `first` is the same for a whole launch: its quotient and
   remainder are scalar work, and what is left per sample is a 32-bit division by a constant (the 64-bit one is a sequence of some
   thirty instructions, and the generator and the fast kernels ran it two or three times per sample) */
RT_DEV uint32_t tile_of(const int64_t first, const int k, int &i)
{
    const int64_t ft = first / RT_BLUE_N;
    const uint32_t u = (uint32_t)(first - ft * RT_BLUE_N) + (uint32_t)k, q = u / (uint32_t)RT_BLUE_N;
    i = (int)(u - q * (uint32_t)RT_BLUE_N);
    return (uint32_t)ft + q;
}
RT_DEV void root_ray(const RtDevScene &S, const RtSampleGrid &g, int64_t first, int k, const double *rays_org, const double *rays_dir, const double *rays_aux,
                     DRay &r, double *aux)
{
    if (rays_org) {
        r.org = d3(rays_org[3 * k], rays_org[3 * k + 1], rays_org[3 * k + 2]);
        r.dir = d3(rays_dir[3 * k], rays_dir[3 * k + 1], rays_dir[3 * k + 2]);
        for (int i = 0; i < 6; i++) aux[i] = rays_aux ? rays_aux[6 * k + i] : S.aux_init[i];
    } else {
        double pos[2], v[2];
        sample_at(S, g, first + k, pos, v, aux);
        r = camera_emit(S.cam, v[0], v[1], aux[0], aux[1]);     /* CastFunction, RayTracer2020.cpp:20-31 */
    }
}
/* The same ray to within ~1e-13 relative, with the divisions of the sample position and of Camera::Emit replaced
   by multiplications with refined reciprocals: ONLY for the conservative cull in k_generate_cull (the culling boxes
   are padded by 1e-6 relative + 1e-9 absolute, orders above this error); survivors get the exact ray. */
RT_DEV DRay root_ray_for_cull(const RtDevScene &S, const RtSampleGrid &g, int64_t first, int k, const double *rays_org, const double *rays_dir)
{
    DRay r;
    if (rays_org) {
        r.org = d3(rays_org[3 * k], rays_org[3 * k + 1], rays_org[3 * k + 2]);
        r.dir = d3(rays_dir[3 * k], rays_dir[3 * k + 1], rays_dir[3 * k + 2]);
        return r;
    }
    if (g.sampler != RT_SAMPLER_FASTBLUE || S.cam.type == RT_CAM_APERTURE) {
        double pos[2], v[2], aux[6];
        sample_at(S, g, first + k, pos, v, aux);
        return camera_emit(S.cam, v[0], v[1], aux[0], aux[1]);
    }
    int i;
    const uint32_t tile = tile_of(first, k, i);
    const uint32_t txu = tile / (uint32_t)g.tiles_y_local;
    const int tx = (int)txu, ty = (int)(tile - txu * (uint32_t)g.tiles_y_local) + g.tile_y0;
    const double vx = S.blue[2 * i] * (double)RT_BLUE_NSQRT + (double)tx * (double)RT_BLUE_NSQRT;
    const double vy = S.blue[2 * i + 1] * (double)RT_BLUE_NSQRT + (double)ty * (double)RT_BLUE_NSQRT;
    const double v0 = (vx + g.adjust_x) * g.inv_size, v1 = (vy + g.adjust_y) * g.inv_size;
    const RtCamera &c = S.cam;
    const double kf = 35.0 * cull_rcp(c.focal);
    const double x = (v0 - 0.5) * kf, y = (v1 - 0.5) * kf;
    r.org = d3(c.org[0], c.org[1], c.org[2]);
    if (c.type == RT_CAM_PERSPECTIVE) { r.dir = d3(x, y, -1.0); return r; }
    if (c.type == RT_CAM_ORTHOGRAPHIC) { r.org = d3(x, y, c.org[2]); r.dir = d3(0.0, 0.0, -1.0); return r; }
    const double id = cull_rcp(x * x + y * y + 4.0);
    r.dir = d3(4.0 * x * id, 4.0 * y * id, 1.0 - 8.0 * id);
    return r;
}

/* ------------------------------------------------------------------ tile mask (no counterpart in the reference)
 * A blue-noise tile is a small rectangle of the image; the rays through it lie in a cone around the ray through its
 * centre whose half-angle is reached at one of its corners (the set of image points within an angle of a direction is
 * a disc for the stereographic camera and a conic for the perspective one -- convex, so four corners decide; for the
 * orthographic camera the origins move instead, and the boxes are inflated by that much).  A box is replaced by its
 * bounding sphere: cone and sphere do not meet  =>  no ray of the tile meets the box.  Conservative by construction,
 * with slack far above the rounding of the sample positions. */
/* lens, inv_fd (ApertureCamera, Cameras.h:52-67): a ray starts at most `lens` away from the apex and passes through the point of the
   central ray -- the one the cone is about -- at the focus distance, so at distance l along it is at most lens * |1 - l / fd| away from
   the central ray's point: the sphere grows by lens * max(1, (dist + r) / fd) */
RT_DEV bool cone_may_touch(const D3 &o, const D3 &d /*unit*/, double cos_t, double sin_t, double grow, const double lo[3], const double hi[3],
                           const double lens = 0.0, const double inv_fd = 0.0)
{
    const D3 c = d3(0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]));
    const D3 h = d3(0.5 * (hi[0] - lo[0]), 0.5 * (hi[1] - lo[1]), 0.5 * (hi[2] - lo[2]));
    double r = sqrt(dot(h, h)) * (1.0 + 1e-9) + grow + 1e-9;
    if (!(r < 1e100)) return true;                      /* unbounded or not a number: cannot exclude */
    const D3 v = c - o;
    if (lens > 0.0) r += lens * fmax(1.0, (sqrt(dot(v, v)) + r) * inv_fd) * (1.0 + 1e-9);
    const double dist2 = dot(v, v);
    if (!(dist2 > r * r)) return true;                  /* the apex is inside the sphere */
    const double dist = sqrt(dist2), sin_b = r / dist, cos_b = sqrt(fmax(0.0, 1.0 - sin_b * sin_b));
    const double cos_a = dot(v, d) / dist;              /* angle between the axis and the centre of the sphere */
    return cos_a >= cos_b * cos_t - sin_b * sin_t - 1e-9;       /* a <= b + t, with b + t < pi */
}
__global__ void __launch_bounds__(256) k_tile_mask(const RtDevScene Sv, RtSampleGrid g, unsigned char *mask, unsigned short *cand)
{
    const RtDevScene &S = Sv;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.tiles_x * g.tiles_y_local) return;
    const int tx = t / g.tiles_y_local, ty = t - tx * g.tiles_y_local + g.tile_y0;
    const double N = (double)RT_BLUE_NSQRT, x0 = (double)tx * N, y0 = (double)ty * N;
    DRay rc = camera_emit(S.cam, (x0 + 0.5 * N + g.adjust_x) / g.size, (y0 + 0.5 * N + g.adjust_y) / g.size, 0.0, 0.0);
    const D3 dc = normalize(rc.dir);
    double cos_t = 1.0, grow = 0.0;
    for (int k = 0; k < 4; k++) {
        /* samples lie in [x0, x0 + 40] x [y0, y0 + 40] (the tile holds points at exactly 1.0); a hair wider */
        const double cx = (k & 1) ? x0 + N + 1e-6 : x0 - 1e-6, cy = (k & 2) ? y0 + N + 1e-6 : y0 - 1e-6;
        const DRay rk = camera_emit(S.cam, (cx + g.adjust_x) / g.size, (cy + g.adjust_y) / g.size, 0.0, 0.0);
        const double c = dot(normalize(rk.dir), dc);
        if (c < cos_t) cos_t = c;
        const D3 dorg = rk.org - rc.org;
        const double m = sqrt(dot(dorg, dorg));
        if (m > grow) grow = m;
    }
    cos_t = cos_t - 1e-12 - 1e-6 * (1.0 - cos_t);       /* widen the cone a little */
    if (!(cos_t > 0.0)) { mask[t] = 255; return; }      /* absurdly wide tile: no statement */
    double lens = 0.0, inv_fd = 0.0;
    if (S.cam.type == RT_CAM_APERTURE) {                /* lens points: org + vX u + vY v with |u|, |v| <= 1 */
        lens = (sqrt(S.cam.vX[0] * S.cam.vX[0] + S.cam.vX[1] * S.cam.vX[1] + S.cam.vX[2] * S.cam.vX[2]) +
                sqrt(S.cam.vY[0] * S.cam.vY[0] + S.cam.vY[1] * S.cam.vY[1] + S.cam.vY[2] * S.cam.vY[2])) * (1.0 + 1e-9) + 1e-12;
        if (!(S.cam.focus_distance > 0.0)) { mask[t] = 255; return; }
        inv_fd = (1.0 + 1e-9) / S.cam.focus_distance;
    }
    const double sin_t = sqrt(fmax(0.0, 1.0 - cos_t * cos_t));
    grow *= 1.0 + 1e-6;
    /* 0 = nothing can be touched; 1..RT_TILE_CAND = that many root-region children listed in cand (all of them, ascending);
       255 = something can be touched, no list */
    int n = 0;
    bool walk = false;
    unsigned short list[RT_TILE_CAND];
    if (S.root_cull == 2) walk = cone_may_touch(rc.org, dc, cos_t, sin_t, grow, S.boxes[S.root_cull_arg].lo, S.boxes[S.root_cull_arg].hi, lens, inv_fd);
    else if (S.root_cull == 1) {
        const RtRegion &rg = S.regions[S.root_cull_arg];
        /* children without a finite box (a sky dome) are on every tile's list; without lists they leave nothing to say */
        for (int q = 0; q < rg.n_always; q++) {
            if (n < RT_TILE_CAND && cand) list[n++] = (unsigned short)S.region_always[rg.always0 + q];
            else walk = true;
        }
        if (!walk && rg.bvh_root >= 0) {
            int stack[RT_BVH_STACK], sp = 0, node = rg.bvh_root;
            for (;;) {
                const RtBvhNode &nd = S.bvh[node];
                for (int k = 0; k < 4 && !walk; k++) {
                    if (nd.child[k] == RT_BVH_EMPTY) continue;
                    const double lo[3] = { nd.lo[0][k], nd.lo[1][k], nd.lo[2][k] }, hi[3] = { nd.hi[0][k], nd.hi[1][k], nd.hi[2][k] };
                    if (!cone_may_touch(rc.org, dc, cos_t, sin_t, grow, lo, hi, lens, inv_fd)) continue;
                    if (nd.child[k] < 0) {
                        if (n < RT_TILE_CAND && cand) list[n++] = (unsigned short)(~nd.child[k]);
                        else walk = true;
                    } else if (sp < RT_BVH_STACK) stack[sp++] = nd.child[k];
                    else walk = true;
                }
                if (walk || sp == 0) break;
                node = stack[--sp];
            }
        }
    } else
        walk = true;
    if (walk) { mask[t] = 255; return; }
    for (int i = 1; i < n; i++) {               /* ascending = the reference's evaluation order */
        unsigned short v = list[i];
        int j = i - 1;
        for (; j >= 0 && list[j] > v; --j) list[j + 1] = list[j];
        list[j + 1] = v;
    }
    for (int i = 0; i < n; i++) cand[(size_t)t * RT_TILE_CAND + i] = list[i];
    mask[t] = (unsigned char)n;
}

/* need[t] = some tile within D tiles of t (Chebyshev) is not masked */
__global__ void __launch_bounds__(256) k_tile_need(const unsigned char *mask, unsigned char *need, int tiles_x, int tiles_y, int D)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    const int tx = t / tiles_y, ty = t - tx * tiles_y;
    unsigned char any = 0;
    for (int x = tx - D; x <= tx + D && !any; x++) {
        if (x < 0 || x >= tiles_x) continue;
        for (int y = ty - D; y <= ty + D; y++) {
            if (y < 0 || y >= tiles_y) continue;
            if (mask[x * tiles_y + y]) { any = 1; break; }
        }
    }
    need[t] = any;
}

/* the auxiliary dimensions of root sample k that later stages can ask for */
RT_DEV void aux_store(const RtDevScene &S, const RtPipe &P, int k, const double *aux)
{
    if (S.aux_light) { P.aux[(size_t)2 * P.ncap + k] = aux[2]; P.aux[(size_t)3 * P.ncap + k] = aux[3]; }
    if (S.aux_tex) { P.aux[(size_t)4 * P.ncap + k] = aux[4]; P.aux[(size_t)5 * P.ncap + k] = aux[5]; }
}
template <bool LITE>
RT_DEV void aux_load(const RtDevScene &S, const RtPipe &P, int root, double *aux)
{
    for (int i = 0; i < 6; i++) aux[i] = 0.0;
    if (S.aux_light) { aux[2] = P.aux[(size_t)2 * P.ncap + root]; aux[3] = P.aux[(size_t)3 * P.ncap + root]; }
    if constexpr (!LITE) if (S.aux_tex) { aux[4] = P.aux[(size_t)4 * P.ncap + root]; aux[5] = P.aux[(size_t)5 * P.ncap + root]; }   /* LITE hands textured hits to FULL */
}

/* hint: bits 21..31 of a root frame's flags word = 1 + the ONE child of the root region the sample's tile can touch (0 = no
   such statement): k_visual<.., TILES> then needs no tile lookups for the ray (FR_HINT_SHIFT, dropped by FR_KEEP_MASK) */
#define RT_BOUND_ON_DEVICE (1 << 29)    /* in k_visual's n_chains_lds argument: see there */
#define FR_HINT_SHIFT 21
#define FR_TRACED     (1 << 20) /* a level-0 frame k_fast has traced already (it needs shadow rays or children): k_visual passes over it */
RT_DEV void root_frame(const RtDevScene &S, const RtPipe &P, int f, int k, const DRay &r, const double *aux, int hint = 0)
{
    d3_store(P.f_ray, P.fcap, 0, f, r.org);
    d3_store(P.f_ray, P.fcap, 3, f, r.dir);
    aux_store(S, P, k, aux);
    P.f_parent[f] = -1;
    P.f_root[f] = k;
    P.f_flags[f] = (int)((unsigned)hint << FR_HINT_SHIFT);
}

/* sample k of a chunk (chunks start on tile boundaries) -> its place in the sample buffer (RtSampleGrid::rgb_slot) */
RT_DEV int rgb_slot_of(const unsigned short *slot, const int k)
{
    if (!slot) return k;
    const int i = k % RT_BLUE_N;
    return k - i + (int)slot[i];
}
/* work item k of a chunk -> the sample it traces (RtSampleGrid::fb_order) */
RT_DEV int trace_order(const RtSampleGrid &g, int64_t first, int k)
{
    if (!g.fb_order) return k;
    (void)first;                            /* chunks start on tile boundaries when fb_order is set (grid_device_tables, run_range) */
    const int i = k % RT_BLUE_N;            /* 32-bit, by a constant: a 64-bit remainder here cost the generator 9 % */
    return k - i + (int)g.fb_order[i];
}

/* root frames for rays [0, n) of a chunk, frame index = ray index */
__global__ void __launch_bounds__(256) k_generate(const RtDevScene Sv, RtSampleGrid g, RtPipe P, int64_t first, int n, const double *rays_org, const double *rays_dir, const double *rays_aux)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const int k = trace_order(g, first, f);
    double aux[6];
    DRay r;
    root_ray(Sv, g, first, k, rays_org, rays_dir, rays_aux, r, aux);
    root_frame(Sv, P, f, k, r, aux);
}

/* The same for scenes with a root cull (RtDevScene.root_cull): rays that miss every conservative box of the
   scene (scene_may_hit) are finished here -- black -- and only the others become frames.  One workgroup
   handles RT_GEN_SPAN consecutive rays and appends its survivors with ONE atomic, in ray order, so that the
   frames of a workgroup of k_visual still come from one or two 40x40 sample tiles (coherent waves, even
   workgroup run times). */
#define RT_GEN_ITERS 6
#define RT_GEN_SPAN  (RT_GEN_ITERS * 256)
static_assert(RT_GEN_SPAN <= RT_BLUE_N, "a generator workgroup's samples must lie in at most two sampling tiles (its early exit looks at the first and the last)");
/* Survivors go one of two ways.  FAST: the sample's tile lists the children of the root region its rays can touch, and
   every child whose box the (approximate, conservative) ray touches is a bare chain of a two-root primitive without rare
   features (RtDevScene.root_fast): k_fast handles such a ray from (sample, chain) or (sample, touched entries of the
   tile's list) without a frame.  Everything else becomes a root frame for k_visual as before.  Both queues are appended
   in sample order with one atomic per workgroup each. */
__global__ void __launch_bounds__(256) k_generate_cull(const RtDevScene Sv, RtSampleGrid g, RtPipe P, int64_t first, int n, const double *rays_org, const double *rays_dir,
                                                       const double *rays_aux, float *out_rgb, float *out_spec)
{
    const RtDevScene &S = Sv;
    __shared__ int wave_count[2][RT_GEN_ITERS][4], wave_base[2][RT_GEN_ITERS][4], block_base[2];
    const int k0 = blockIdx.x * RT_GEN_SPAN;
    if (g.tile_mask && g.mask_skips_rgb && !rays_org && k0 < n) {
        /* the workgroup's samples belong to one or two sampling tiles (1536 consecutive samples, tiles of 1600): when no ray through
           either can touch the scene there is nothing to do here at all -- not even the staging below.  Half of S0's samples. */
        int i_;
        const int k1 = (k0 + RT_GEN_SPAN < n ? k0 + RT_GEN_SPAN : n) - 1;
        const uint32_t ta = tile_of(first, k0, i_), tb = tile_of(first, k1, i_);
        if (g.tile_mask[ta] == 0 && g.tile_mask[tb] == 0) return;
    }
    const SceneLds lds = scene_lds_load(S, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool fast_on = S.root_fast && P.tile_cand && !rays_org;       /* the host runs k_fast and the tile-list build of k_visual together */
    unsigned keep_bits = 0, fast_bits = 0;
    unsigned long long prefix = 0, prefix_f = 0;    /* 7 bits per iteration: survivors of that kind before this lane in its wave */
    int fast_chain[RT_GEN_ITERS];       /* one chain (>= 0), or bit 31 + the touched entries of the tile's list */
#pragma unroll
    for (int it = 0; it < RT_GEN_ITERS; it++) {
        int k = k0 + it * 256 + threadIdx.x;
        bool keep = false, fast = false;
        fast_chain[it] = 0;
        if (k < n) {
            k = trace_order(g, first, k);
            /* a wave's 64 samples belong to one tile (1600 = 25 x 64, chunks start at multiples of 64): uniform branch */
            int ti_;
            /* ... and says so to the compiler: the tile's mask, its list and the listed boxes are then scalar loads and scalar loop
               control instead of 64 identical vector loads each */
            const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_of(first, k, ti_));
            const int tm = (g.tile_mask && !rays_org) ? (int)g.tile_mask[tile] : 255;
            const bool masked = tm == 0;
            if (!masked) {
                if (tm != 255 && fast_on && S.cull32 && S.cam.type <= RT_CAM_STEREOGRAPHIC) {
                    /* The tile's list in single precision (option cull_f32).  This test only decides which of the listed boxes a sample is
                       tried against: a box kept in error costs a chain test that misses, a box dropped in error would lose a hit -- so
                       the boxes of S.cull32 are the culling boxes grown by 1e-4 of the scene's scale (camera distance + box
                       coordinates), some thirty times what this arithmetic can be off by: the direction is good to ~1e-6 relative
                       (a dozen float operations on numbers of order one), so a point at distance L along the ray is displaced by
                       L * 1e-6, and the slab arithmetic adds a few float roundings of numbers up to the scene's scale.  A direction
                       component too small to invert asks whether the origin lies between the planes, as cull_ray does.  The survivors'
                       rays are computed exactly by the kernels that take them. */
                    const uint32_t txu = tile / (uint32_t)g.tiles_y_local;
                    const int tx = (int)txu, ty = (int)(tile - txu * (uint32_t)g.tiles_y_local) + g.tile_y0;
                    const float vx = (float)S.blue[2 * ti_] * (float)RT_BLUE_NSQRT + (float)(tx * RT_BLUE_NSQRT);
                    const float vy = (float)S.blue[2 * ti_ + 1] * (float)RT_BLUE_NSQRT + (float)(ty * RT_BLUE_NSQRT);
                    const float v0 = (vx + (float)g.adjust_x) * (float)g.inv_size, v1 = (vy + (float)g.adjust_y) * (float)g.inv_size;
                    const float kf = 35.0f / (float)S.cam.focal;
                    const float x = (v0 - 0.5f) * kf, y = (v1 - 0.5f) * kf;
                    float d[3];
                    if (S.cam.type == RT_CAM_PERSPECTIVE) { d[0] = x; d[1] = y; d[2] = -1.0f; }
                    else { const float id = __builtin_amdgcn_rcpf(x * x + y * y + 4.0f); d[0] = 4.0f * x * id; d[1] = 4.0f * y * id; d[2] = 1.0f - 8.0f * id; }
                    const float o[3] = { (float)S.cam.org[0], (float)S.cam.org[1], (float)S.cam.org[2] };
                    float inv[3], noi[3];
                    bool par[3];
                    for (int a = 0; a < 3; a++) {
                        par[a] = !(fabsf(d[a]) > 1.0e-30f);
                        inv[a] = par[a] ? 0.0f : __builtin_amdgcn_rcpf(d[a]);
                        noi[a] = -(o[a] * inv[a]);
                    }
                    const float *cb = S.cull32 + (size_t)6 * S.regions[S.root_cull_arg].child_pc0;
                    const unsigned short *tl = g.tile_cand + (size_t)tile * RT_TILE_CAND;
                    unsigned m = 0;
                    bool elig = true;
                    int c1 = 0;
                    for (int q = 0; q < tm; q++) {
                        const float *bx = cb + 6 * (size_t)tl[q];
                        float t0 = 0.0f, t1 = 3.0e38f;
                        bool in = true;
                        for (int a = 0; a < 3; a++) {
                            if (!par[a]) {
                                const float ta = __builtin_fmaf(bx[a], inv[a], noi[a]), tb = __builtin_fmaf(bx[3 + a], inv[a], noi[a]);
                                t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
                            } else in = in && !(o[a] < bx[a] || o[a] > bx[3 + a]);
                        }
                        if (in && t0 - t1 <= (fabsf(t0) + fabsf(t1)) * 4.0e-6f) { m |= 1u << q; c1 = S.root_fast[tl[q]]; elig = elig && c1 >= 0; }
                    }
                    keep = m != 0;
                    if (keep && elig) { fast = true; keep = false; fast_chain[it] = (m & (m - 1)) ? (int)(0x80000000u | m) : c1; }
                } else {
                const DRay r = root_ray_for_cull(S, g, first, k, rays_org, rays_dir);
                if (tm != 255 && g.tile_cand) {
                    /* the tile's list holds every child a ray through the tile can touch: test their boxes, no BVH walk */
                    const CullRay cr = cull_ray(r, RT_INFINITY);
                    const RtBox *cb = S.region_child_boxes + S.regions[S.root_cull_arg].child_pc0;
                    const unsigned short *tl = g.tile_cand + (size_t)tile * RT_TILE_CAND;
                    if (fast_on) {
                        /* which listed children the ray can touch; k_fast takes the ray when every one of them is eligible */
                        unsigned m = 0;
                        bool elig = true;
                        int c1 = 0;
                        for (int q = 0; q < tm; q++)
                            if (cull_box_hit(cb[tl[q]].lo, cb[tl[q]].hi, cr)) { m |= 1u << q; c1 = S.root_fast[tl[q]]; elig = elig && c1 >= 0; }
                        keep = m != 0;
                        if (keep && elig) { fast = true; keep = false; fast_chain[it] = (m & (m - 1)) ? (int)(0x80000000u | m) : c1; }
                    } else
                        for (int q = 0; q < tm && !keep; q++) keep = cull_box_hit(cb[tl[q]].lo, cb[tl[q]].hi, cr);
                } else {
                    keep = scene_may_hit(S, lds, r, RT_INFINITY);
#ifdef RT_NH_HIST
                    if (keep) atomicAdd(&P.counters[CTR_PROF + 8], 1);
#endif
                }
                }
            }
            if (!keep && !fast && !(masked && g.mask_skips_rgb)) {
                if (out_rgb) { const size_t ks = (size_t)rgb_slot_of(rays_org ? 0 : g.rgb_slot, k); out_rgb[ks * 3] = 0.0f; out_rgb[ks * 3 + 1] = 0.0f; out_rgb[ks * 3 + 2] = 0.0f; }
                if (out_spec) for (int i = 0; i < RT_NSPEC; i++) out_spec[(size_t)k * RT_NSPEC + i] = 0.0f;
            }
        }
        const unsigned long long b = __ballot(keep), bf = __ballot(fast), below = (1ull << lane) - 1ull;
        if (keep) keep_bits |= 1u << it;
        if (fast) fast_bits |= 1u << it;
        prefix |= (unsigned long long)__popcll(b & below) << (7 * it);
        prefix_f |= (unsigned long long)__popcll(bf & below) << (7 * it);
        if (lane == 0) { wave_count[0][it][wave] = __popcll(b); wave_count[1][it][wave] = __popcll(bf); }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int q = threadIdx.x;
        int total = 0;
        for (int it = 0; it < RT_GEN_ITERS; it++)
            for (int w = 0; w < 4; w++) { wave_base[q][it][w] = total; total += wave_count[q][it][w]; }
        block_base[q] = total > 0 ? atomicAdd(&P.counters[q ? CTR_FAST : CTR_FRAMES], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RT_GEN_ITERS; it++) {
        if (fast_bits & (1u << it)) {
            const int k = trace_order(g, first, k0 + it * 256 + threadIdx.x);
            const int e = block_base[1] + wave_base[1][it][wave] + (int)((prefix_f >> (7 * it)) & 127u);
            nt_store(&P.fast_q[e], k);
            nt_store(&P.fast_q[(size_t)P.ncap + e], fast_chain[it]);
        }
        if (!(keep_bits & (1u << it))) continue;
        const int k = trace_order(g, first, k0 + it * 256 + threadIdx.x);
        double aux[6];
        DRay r;
        root_ray(S, g, first, k, rays_org, rays_dir, rays_aux, r, aux);
        int hint = 0;
        if (g.tile_mask && g.tile_cand && !rays_org) {
            int ti_;
            const uint32_t tile = tile_of(first, k, ti_);
            if (g.tile_mask[tile] == 1) { const int c = g.tile_cand[(size_t)tile * RT_TILE_CAND]; if (c < 2047) hint = c + 1; }
        }
        root_frame(S, P, block_base[0] + wave_base[0][it][wave] + (int)((prefix >> (7 * it)) & 127u), k, r, aux, hint);
    }
}

/* first kernel of a pass: every counter to zero except the overflow line (kept for the whole render call), the frame
   count to the number of root frames k_generate makes (0 when k_generate_cull counts them itself) */
__global__ void k_pass_begin(int *counters, int frames0)
{
    for (int i = threadIdx.x; i < CTR_COUNT; i += blockDim.x) {
#ifdef RT_NH_HIST
        if (i >= CTR_PROF && i < CTR_PROF + CTR_LINE) continue;
#endif
        if (i < CTR_OVERFLOW || i >= CTR_OVERFLOW + CTR_LINE) counters[i] = (i == CTR_FRAMES) ? frames0 : 0;
    }
}
/* between the tracing kernels and the shading of a level: freeze the level's frame and shadow-ray counts where the next
   kernels of the stream read them (k_shade appends the next level's frames to CTR_FRAMES while it runs) */
__global__ void k_level_snap(RtPipe P, int level0, int gen_frames)
{
    if (threadIdx.x == 0) {
        const int nf = P.counters[CTR_FRAMES], ns = P.counters[CTR_SHADOW];
        P.counters[CTR_SNAP] = nf < P.fcap ? nf : P.fcap;
        P.counters[CTR_SNAP + 1] = ns < P.scap ? ns : P.scap;
        if (level0) {           /* measurement: how the call's primary rays were traced (the line outlives the pass) */
            P.counters[CTR_OVERFLOW + 2] += P.counters[CTR_FAST];
            P.counters[CTR_OVERFLOW + 3] += (gen_frames < 0 ? nf : gen_frames) - P.counters[CTR_FAST + 1];     /* level-0 frames k_visual traced: all but k_fast's own */
            if (nf > 0) P.counters[CTR_OVERFLOW + 4] = 1;      /* level 0 had frames at all (the next call's expectation: k_shade_loop) */
        }
    }
}

/* workgroup sizes of the common (LITE) kernels: one workgroup per CU, three waves per SIMD, which also bounds
   their VGPRs (512 / waves per SIMD) */
#ifdef RT_SMALL_BLOCKS      /* A/B variant: 256-thread workgroups, several per CU, the same waves per SIMD */
#define RT_LITE_VISUAL_THREADS 256
#define RT_LITE_SHADOW_THREADS 256
#define RT_LITE_VISUAL_PER_CU  3
#define RT_LITE_SHADOW_PER_CU  4
#define RT_LB_VISUAL(LITE, MODE) __launch_bounds__(256, (LITE) ? 3 : ((MODE) < 0 ? 1 : 2))
#define RT_LB_SHADOW(LITE, MODE) __launch_bounds__(256, (LITE) ? 4 : ((MODE) < 0 ? 1 : 2))
#else
#ifndef RT_LITE_VISUAL_THREADS
#define RT_LITE_VISUAL_THREADS 768
#endif
#ifndef RT_LITE_SHADOW_THREADS
#define RT_LITE_SHADOW_THREADS 768     /* (1024 = four waves per SIMD and 128 registers until round 3: S2 435 -> 411 ms, MESH10M 741 -> 730, S1 282 -> 279 with three and 168; 512: 436 / 739 / 295) */
#endif
#define RT_LITE_VISUAL_PER_CU  1
#define RT_LITE_SHADOW_PER_CU  1
/* FULL kernels come in two register budgets.  MODE 0 (default): 256 registers = two waves per SIMD (two workgroups of
   256 per CU): mesh walks wait on memory and run 1.4x faster with a second wave to switch to; S2 458 ms against 472,
   S1 the same.  MODE -1 (option full_waves = 1): the whole register file, one wave per SIMD, no spills. */
#ifndef RT_FULL_PER_CU
#define RT_FULL_PER_CU 2
#endif
#define RT_LB_VISUAL(LITE, MODE) __launch_bounds__((LITE) ? RT_LITE_VISUAL_THREADS : 256, ((LITE) || (MODE) < 0) ? 1 : RT_FULL_PER_CU)
#define RT_LB_SHADOW(LITE, MODE) __launch_bounds__((LITE) ? RT_LITE_SHADOW_THREADS : 256, ((LITE) || (MODE) < 0) ? 1 : RT_FULL_PER_CU)
#endif
/* ------------------------------------------------------------------ k_visual
 * RayTracer::VisualRay (Shade.cpp:401-520) up to the light loop, for frames [f0, f1).
 * LITE = true is the common kernel: a ray that meets a torus, a textured hit, a propagating medium or a spectral
 * metal is put on the slow list instead and redone by the FULL instantiation (launched over that list). */
/* MODE: what the workgroup stages in LDS besides the top of the BVH -- 0 nothing, 1 the chain records, 2 the chain records
   and the small scene tables (scene_lds_load); chosen per scene at upload, compiled in so that the loads are LDS loads */
/* What the walk of one visual ray through its hit list leaves behind (Shade.cpp:424-516): whether it stopped at a
   boundary, the radiance and transmittance gathered on the way, and which lights need a shadow ray.  The boundary's
   geometry goes to a SINK the moment it is known: k_visual's sink stores it as the frame record for k_shade (the values
   die at once: the interpreter kernels have no registers to keep them), k_fast's keeps it in registers and finishes the
   sample from it where nothing else is needed. */
struct Walk {
    int      flags;             /* FR_BOUNDARY | FR_PERTURBED | FR_FINAL | FR_FIN */
    unsigned mask;              /* lights that get a shadow ray (Shade.cpp:477-481) */
    int      n_shadow;
    bool     bail;              /* LITE: a rare feature on the way -- redo the ray in the FULL kernel */
    bool     tr_one;
    int      r_state;           /* 0: R is zero, 1: R is the boundary's ambient term alone, 2: anything else */
    D3       P;                 /* the boundary point (origin of the shadow rays) */
    Spec     R, tr;
};
struct FrameSink {              /* the frame record of frame f; `flags` collects what k_shade can tell from a flag instead of a load */
    const RtPipe &P; int f; int flags;
    RT_DEV void boundary(const D3 &Ppt, const D3 &cvN, const D3 &vE, const D3 &hn, double fNE, double F, double fNT, double eta,
                         int rough, int luster, int mat, const Spec &color, bool color_is_mat)
    {
        d3_store(P.f_geo, P.fcap, GEO_P, f, Ppt);
        d3_store(P.f_geo, P.fcap, GEO_N, f, cvN);
        d3_store(P.f_geo, P.fcap, GEO_E, f, vE);
        d3_store(P.f_geo, P.fcap, GEO_HN, f, hn);
        P.f_geo[(size_t)GEO_FNE * P.fcap + f] = fNE;
        P.f_geo[(size_t)GEO_F * P.fcap + f] = F;
        P.f_geo[(size_t)GEO_FNT * P.fcap + f] = fNT;
        P.f_geo[(size_t)GEO_ETA * P.fcap + f] = eta;
        P.f_ids[(size_t)IDS_ROUGH * P.fcap + f] = rough;
        P.f_ids[(size_t)IDS_LUSTER * P.fcap + f] = luster | (mat << 8);
        if (color_is_mat) flags |= FR_COLOR_MAT; else spec_store(P.f_color, P.fcap, f, color);
    }
    RT_DEV void fin(const Spec &v) { spec_store(P.f_fin, P.fcap, f, v); }
};
struct RegSink {                /* the same values, kept */
    D3 N, E, HN; double fNE, F, fNT, eta; int rough, luster, mat; Spec color; bool color_is_mat;
    RT_DEV void boundary(const D3 &, const D3 &cvN, const D3 &vE, const D3 &hn, double fNE_, double F_, double fNT_, double eta_,
                         int rough_, int luster_, int mat_, const Spec &color_, bool color_is_mat_)
    {
        N = cvN; E = vE; HN = hn; fNE = fNE_; F = F_; fNT = fNT_; eta = eta_; rough = rough_; luster = luster_; mat = mat_; color = color_; color_is_mat = color_is_mat_;
    }
    RT_DEV void fin(const Spec &) {}
};
struct ArenaHits {              /* a hit list in the interpreter's arena */
    const Arena &A; int start;
    RT_DEV const HitRec &get(int i) const { return A.h[start + i]; }
    RT_DEV const double *own_minv() const { return 0; }
};

template <bool LITE, class Hits, class Sink>
RT_DEV void visual_walk(const RtDevScene &S, const DRay &ray, const double *aux, const Hits &H, const int count, Walk &B, Sink &K)
{
    B.flags = 0; B.mask = 0; B.n_shadow = 0; B.bail = false; B.tr_one = true; B.r_state = 0;
    B.P = d3(0, 0, 0);
    /* Shade.cpp:424-432: regularise into (0, t, ...) */
    int i = 0, pm = 0;
    if (count > 0 && !(H.get(0).t > 0.0)) { pm = H.get(0).mat; i = 1; }
    double t = 0.0;
    Spec tr = spec_const(1.0f), R = spec_const(0.0f);
    /* Q13: colour / luster a solid texture gave the material the walk is inside of (FULL kernel only: LITE hands
       textured hits over) */
    int pm_luster_ov = -1, pm_has_color = 0;
    Spec pm_color = spec_const(0.0f);
    bool bail = false;
    for (; i < count && !bail; i++) {
        ShadeHit h;
#ifdef RT_Z2
        transform_normals<LITE>(S, ray, H.get(i), h, aux, bail);
#else
        transform_normals<LITE>(S, ray, H.get(i), h, aux, bail, H.own_minv());
#endif
        if (bail) break;
        const RtMaterialDev &mp = S.mats[pm];
        int mp_luster = mp.luster;
        const float *mp_colorp = mp.color;
        if constexpr (!LITE) { if (pm_luster_ov >= 0) mp_luster = pm_luster_ov; if (pm_has_color) mp_colorp = pm_color.v; }
        if (mp_luster == RT_MAT_NOSHADE) {                  /* Shade.cpp:436-437 (Q2: no depth leak) */
            R = spec_load(mp_colorp);
            B.r_state = 2;
            B.flags |= FR_FINAL;
            break;
        }
        if (mp.prop) {
            if constexpr (LITE) { bail = true; break; }
            else {
                Spec ptr, pst;
                propagate_in(S, mp.prop, ray_at(ray, t), ray_at(ray, h.t), mp_colorp, mp.refr, ptr, pst);
                tr = tr * ptr;
                R = R + pst;
                B.r_state = 2; B.tr_one = false;
            }
        }
        const RtMaterialDev &mh = S.mats[h.mat];
        int mh_luster = mh.luster;
        if constexpr (!LITE) if (h.luster_ov >= 0) mh_luster = h.luster_ov;
        if ((mp.refr != mh.refr) || (mh_luster != RT_MAT_TRANSPARENT)) {
            D3 vE, cvN;
            double fNE, F = 0.0, fNT = 0.0;
            consistent_normal(ray, h, vE, cvN, fNE);
            Spec color = spec_load(mh.color);
            bool color_is_mat = true;
            if constexpr (!LITE) if (h.has_color) { color = h.color; color_is_mat = false; }
            if (spectral_metal(mh_luster)) {                /* Q3 */
                if constexpr (LITE) { bail = true; break; }
                else {
                    color = fresnel_spectral(fNE, mp.refr, S.metal_nk + (size_t)mh_luster * 2 * RT_NSPEC, S.metal_nk + (size_t)mh_luster * 2 * RT_NSPEC + RT_NSPEC);
                    F = 1.0;
                    color_is_mat = false;
                }
            } else
                fresnel_real(fNE, mp.refr, mh.refr, F, fNT);
            int rough = hit_roughness(S, h);            /* surf 0 (none) reads as matte: Q11 */
            if (S.has_ambient && rough != RT_SURF_POLISHED) {
                Spec E = spec_load(S.ambient) * RT_PI;      /* Lights.cpp:33-37 */
                R = R + color * E;
                B.r_state = (B.r_state == 0 && color_is_mat) ? 1 : 2;
            }
            const D3 Ppt = ray_at(ray, h.t);
            /* which lights get a shadow ray (Shade.cpp:477-481) */
            for (int li = 0; li < S.n_lights; li++) {
                const RtLightDev &l = S.lights[li];
                D3 lo = d3(l.org[0], l.org[1], l.org[2]);
                if (l.type == RT_LIGHT_AREA) lo = lo + d3(l.vA[0], l.vA[1], l.vA[2]) * aux[2] + d3(l.vB[0], l.vB[1], l.vB[2]) * aux[3];
                D3 dir = lo - Ppt;
                if (dot(cvN, dir) <= 0.0 && mh_luster != RT_MAT_TRANSPARENT) continue;
                if (l.no_shadow) continue;
                B.mask |= 1u << li;
                B.n_shadow++;
            }
            B.flags |= FR_BOUNDARY;
            if (h.perturbed) B.flags |= FR_PERTURBED;
            B.P = Ppt;
            K.boundary(Ppt, cvN, vE, h.n, fNE, F, fNT, mp.refr / mh.refr, rough, mh_luster, h.mat, color, color_is_mat);
            t = h.t;
            pm = h.mat;
            if constexpr (!LITE) { pm_luster_ov = h.luster_ov; pm_has_color = h.has_color; pm_color = h.color; }
            break;
        } else {
            t = h.t;
            pm = h.mat;
            if constexpr (!LITE) { pm_luster_ov = h.luster_ov; pm_has_color = h.has_color; pm_color = h.color; }
        }
    }
    if (!bail && !(B.flags & FR_FINAL) && S.mats[pm].prop) {  /* Shade.cpp:512-516 */
        if constexpr (LITE) bail = true;
        else {
            Spec ptr, pst;
            propagate_in(S, S.mats[pm].prop, ray_at(ray, t), ray_at(ray, RT_INFINITY), pm_has_color ? pm_color.v : S.mats[pm].color, S.mats[pm].refr, ptr, pst);
            K.fin(tr * pst);
            B.flags |= FR_FIN;
        }
    }
    B.bail = bail;
    B.R = R;
    if constexpr (!LITE) B.tr = tr;
}

/* what is left of a frame record after the walk: radiance and transmittance, as flags where a flag says it all (bytes
   not written are the cheapest); returns the frame's flags word */
template <bool LITE>
RT_DEV int frame_finish(const RtPipe &P, const int f, const Walk &B, int flags)
{
    flags |= B.flags;
    if (B.r_state == 0) flags |= FR_R_ZERO; else if (B.r_state == 1) flags |= FR_R_AMBIENT; else spec_store(P.f_R, P.fcap, f, B.R);
    if (LITE || B.tr_one) flags |= FR_TR_ONE; else spec_store(P.f_tr, P.fcap, f, B.tr);
    return flags;
}
/* the shadow rays of a stored frame: slots [base, base + n_shadow) of the queue, in light order */
RT_DEV void shadow_append(const RtDevScene &S, const RtPipe &P, const D3 &Ppt, const unsigned mask, const int base, const int root, const double *aux)
{
    int k = 0;
    for (int li = 0; li < S.n_lights; li++) {
        if (!(mask & (1u << li))) continue;
        int s = base + k++;
        if (s >= P.scap) continue;          /* overflow is reported by the host from the counter */
        const RtLightDev &l = S.lights[li];
        D3 lo = d3(l.org[0], l.org[1], l.org[2]);
        if (l.type == RT_LIGHT_AREA) lo = lo + d3(l.vA[0], l.vA[1], l.vA[2]) * aux[2] + d3(l.vB[0], l.vB[1], l.vB[2]) * aux[3];
        d3_store(P.s_ray, P.scap, 0, s, Ppt);
        d3_store(P.s_ray, P.scap, 3, s, lo - Ppt);
        P.s_light[s] = li;
        P.s_root[s] = root;
    }
}

/* The walk below is written out in this kernel (and again, as visual_walk, for k_fast): routed through the shared function
   the interpreter kernels spill 300 B more per lane and S1 / S2 render 6-7 % slower (same box, alternating: S2 483 against
   454 ms, S1 361 against 338).  Every parity test runs both copies against the oracle. */
/* ALT: the LITE kernels' MESHPRE (the scene's mesh is served from pre-walked crossings), the FULL kernels' SCHED (intersect_scene
   runs one class of opcodes per round) */
template <bool LITE, int MODE, bool TILES = false, bool ALT = false>
__global__ void RT_LB_VISUAL(LITE, MODE) k_visual(const RtDevScene Sv, RtPipe P, int f0, int f1, int n_chains_lds)
{
    constexpr bool CH_LDS = MODE >= 1;
    RtDevScene SL = Sv;
    SceneLds lds = scene_lds_load(Sv, n_chains_lds, MODE == 2 ? &SL : 0);
    const RtDevScene &S = SL;
    if constexpr (!LITE) if (f1 <= f0) { lds.mx_n = P.mx_n; lds.mx_t = P.mx_t; lds.mx_q = P.mx_q; lds.mxcap = P.mxcap; }     /* slow list: k_mesh_cross went first */
    constexpr bool MESHPRE = LITE && ALT, SCHED = !LITE && ALT;
    if constexpr (LITE && MESHPRE) { lds.mx_n = P.mx_n; lds.mx_t = P.mx_t; lds.mx_q = P.mx_q; lds.mxcap = P.mxcap; }
    /* persistent waves: every wave claims 64 consecutive frames at a time until none are left, so the tables staged
       in LDS serve the whole launch and a workgroup never idles behind its slowest wave */
    /* LITE && TILES (the primary level of a pass with tile lists) with RT_BOUND_ON_DEVICE in n_chains_lds: [f0, f1) is an
       upper bound, the level's real end is on the device (CTR_FRAMES: no kernel of the stream changes it while this one
       runs) and is looked at with every claim.  Only in the TILES builds: the code costs the interpreter 300 B of spills
       per lane wherever it is compiled in (S1 / S2, which never use it, rendered 6 % slower with it).  FULL: over the
       slow list the LITE launch left, or (f1 > f0: a small level, no LITE launch before it) over the frames themselves */
    const bool direct = LITE || f1 > f0;
    const int n = direct ? (f1 - f0) : P.counters[CTR_SLOW_V];
    const bool dev_bound = LITE && TILES && (n_chains_lds & RT_BOUND_ON_DEVICE);
    int *const work = &P.counters[LITE ? CTR_WORK_V : CTR_WORK_VS];
    const int wlane = threadIdx.x & 63;
  int wnext = 0, wlimit = 0;      /* the wave's current claim: items [wnext, wlimit) */
  for (;;) {
    if (wnext >= wlimit) {
        int b = 0;
        if (wlane == 0) b = atomicAdd(work, RT_CLAIM);
        b = __shfl(b, 0, 64);
        if (b >= n) break;
        int lim = n;
        if constexpr (LITE && TILES) if (dev_bound) {
            lim = __builtin_amdgcn_readfirstlane(__builtin_nontemporal_load(&P.counters[CTR_FRAMES])) - f0;
            if (lim > n) lim = n;
            if (b >= lim) break;
        }
        wnext = b; wlimit = b + RT_CLAIM < lim ? b + RT_CLAIM : lim;
    }
    const int wbase = wnext;
    wnext += 64;
    bool live = wbase + wlane < wlimit;
    int f;
    if (direct) f = f0 + wbase + wlane;
    else f = live ? P.slow_v[wbase + wlane] : 0;
    if constexpr (LITE && TILES) {      /* level 0 of a pass with tile lists: k_fast's own (traced) frames lie among the generator's */
        if (live && (P.f_flags[f] & FR_TRACED)) live = false;
    }
#ifdef RT_PHASE_PROF
    long long prof_t0 = clock64();
    int prof_v[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    int n_shadow = 0;
    unsigned mask = 0;
    D3 Ppt = d3(0, 0, 0);
    double aux[6];
    int flags = 0;
    bool bail = false;
    if (live) {
        const int flags_raw = P.f_flags[f];
        flags = flags_raw & FR_KEEP_MASK;
        DRay ray;
        ray.org = d3_fetch(P.f_ray, P.fcap, 0, f);
        ray.dir = d3_fetch(P.f_ray, P.fcap, 3, f);
        int root = P.f_root[f];
        aux_load<LITE>(S, P, root, aux);
#ifdef RT_SCHED_POOL     /* round 2's layout for the scheduled kernels: per-lane arenas in a pool in global memory (see intersect_scene) */
        ArenaT<SCHED> A;
        if constexpr (SCHED) A.bind(P.arena_pool, (size_t)(blockIdx.x * blockDim.x + threadIdx.x));
#else
        ArenaT<false> A;
#endif
        int start;
        const unsigned short *tile_list = 0;
        int tile_n = 0, one = -1;
        if constexpr (TILES) {                                  /* launched for the chunk's primary rays only */
            const int hint = (int)((unsigned)flags_raw >> FR_HINT_SHIFT);
            if (hint) one = hint - 1;                           /* the generator has looked the tile up already */
            else {
                const int64_t tile = (P.tile_first + root) / RT_BLUE_N;
                const int m = P.tile_n[tile];
                if (m != 255) { tile_list = P.tile_cand + (size_t)tile * RT_TILE_CAND; tile_n = m; }
            }
        }
        int count = trace_scene<LITE, CH_LDS, TILES, MESHPRE, SCHED>(S, lds, ray, RT_INFINITY, A, start, tile_list, tile_n, one,
                                                                  (LITE && MESHPRE) ? f - P.mx_f0 : ((!LITE && !direct) ? (P.mx_all ? f - P.mx_f0 : wbase + wlane) : -1));
        if (A.overflow) { atomicAdd(&P.counters[CTR_OVERFLOW], 1); atomicOr(&P.counters[CTR_OVERFLOW + 1], A.overflow); }
        bail = A.bail != 0;
#ifdef RT_PHASE_PROF
        for (int q = 1; q < 8; q++) prof_v[q] = A.prof[q];
        prof_v[9] = A.prof[9];
        prof_v[8] = (int)(clock64() - prof_t0);
#endif
        /* Shade.cpp:424-432: regularise into (0, t, ...) */
        int i = 0, pm = 0;
        if (count > 0 && !(A.h[start].t > 0.0)) { pm = A.h[start].mat; i = 1; }
        double t = 0.0;
        Spec tr = spec_const(1.0f), R = spec_const(0.0f);
        int r_state = 0;                /* 0: R is zero, 1: R is the boundary's ambient term alone, 2: anything else */
        bool tr_one = true;
        /* Q13: colour / luster a solid texture gave the material the walk is inside of (FULL kernel only: LITE hands
           textured hits over) */
        int pm_luster_ov = -1, pm_has_color = 0;
        Spec pm_color = spec_const(0.0f);
        for (; i < count && !bail; i++) {
            ShadeHit h;
            transform_normals<LITE>(S, ray, A.h[start + i], h, aux, bail);
            if (bail) break;
            const RtMaterialDev &mp = S.mats[pm];
            int mp_luster = mp.luster;
            const float *mp_colorp = mp.color;
            if constexpr (!LITE) { if (pm_luster_ov >= 0) mp_luster = pm_luster_ov; if (pm_has_color) mp_colorp = pm_color.v; }
            if (mp_luster == RT_MAT_NOSHADE) {                  /* Shade.cpp:436-437 (Q2: no depth leak) */
                R = spec_load(mp_colorp);
                r_state = 2;
                flags |= FR_FINAL;
                break;
            }
            if (mp.prop) {
                if constexpr (LITE) { bail = true; break; }
                else {
                    Spec ptr, pst;
                    propagate_in(S, mp.prop, ray_at(ray, t), ray_at(ray, h.t), mp_colorp, mp.refr, ptr, pst);
                    tr = tr * ptr;
                    R = R + pst;
                    r_state = 2; tr_one = false;
                }
            }
            const RtMaterialDev &mh = S.mats[h.mat];
            int mh_luster = mh.luster;
            if constexpr (!LITE) if (h.luster_ov >= 0) mh_luster = h.luster_ov;
            if ((mp.refr != mh.refr) || (mh_luster != RT_MAT_TRANSPARENT)) {
                D3 vE, cvN;
                double fNE, F = 0.0, fNT = 0.0;
                consistent_normal(ray, h, vE, cvN, fNE);
                Spec color = spec_load(mh.color);
                bool color_is_mat = true;
                if constexpr (!LITE) if (h.has_color) { color = h.color; color_is_mat = false; }
                if (spectral_metal(mh_luster)) {                /* Q3 */
                    if constexpr (LITE) { bail = true; break; }
                    else {
                        color = fresnel_spectral(fNE, mp.refr, S.metal_nk + (size_t)mh_luster * 2 * RT_NSPEC, S.metal_nk + (size_t)mh_luster * 2 * RT_NSPEC + RT_NSPEC);
                        F = 1.0;
                        color_is_mat = false;
                    }
                } else
                    fresnel_real(fNE, mp.refr, mh.refr, F, fNT);
                int rough = hit_roughness(S, h);            /* surf 0 (none) reads as matte: Q11 */
                if (S.has_ambient && rough != RT_SURF_POLISHED) {
                    Spec E = spec_load(S.ambient) * RT_PI;      /* Lights.cpp:33-37 */
                    R = R + color * E;
                    r_state = (r_state == 0 && color_is_mat) ? 1 : 2;
                }
                Ppt = ray_at(ray, h.t);
                /* which lights get a shadow ray (Shade.cpp:477-481) */
                for (int li = 0; li < S.n_lights; li++) {
                    const RtLightDev &l = S.lights[li];
                    D3 lo = d3(l.org[0], l.org[1], l.org[2]);
                    if (l.type == RT_LIGHT_AREA) lo = lo + d3(l.vA[0], l.vA[1], l.vA[2]) * aux[2] + d3(l.vB[0], l.vB[1], l.vB[2]) * aux[3];
                    D3 dir = lo - Ppt;
                    if (dot(cvN, dir) <= 0.0 && mh_luster != RT_MAT_TRANSPARENT) continue;
                    if (l.no_shadow) continue;
                    mask |= 1u << li;
                    n_shadow++;
                }
                flags |= FR_BOUNDARY;
                if (h.perturbed) flags |= FR_PERTURBED;
                d3_store(P.f_geo, P.fcap, GEO_P, f, Ppt);
                d3_store(P.f_geo, P.fcap, GEO_N, f, cvN);
                d3_store(P.f_geo, P.fcap, GEO_E, f, vE);
                d3_store(P.f_geo, P.fcap, GEO_HN, f, h.n);
                P.f_geo[(size_t)GEO_FNE * P.fcap + f] = fNE;
                P.f_geo[(size_t)GEO_F * P.fcap + f] = F;
                P.f_geo[(size_t)GEO_FNT * P.fcap + f] = fNT;
                P.f_geo[(size_t)GEO_ETA * P.fcap + f] = mp.refr / mh.refr;
                P.f_ids[(size_t)IDS_ROUGH * P.fcap + f] = rough;
                P.f_ids[(size_t)IDS_LUSTER * P.fcap + f] = mh_luster | ((int)h.mat << 8);
                if (color_is_mat) flags |= FR_COLOR_MAT; else spec_store(P.f_color, P.fcap, f, color);
                t = h.t;
                pm = h.mat;
                if constexpr (!LITE) { pm_luster_ov = h.luster_ov; pm_has_color = h.has_color; pm_color = h.color; }
                break;
            } else {
                t = h.t;
                pm = h.mat;
                if constexpr (!LITE) { pm_luster_ov = h.luster_ov; pm_has_color = h.has_color; pm_color = h.color; }
            }
        }
        if (!bail && !(flags & FR_FINAL) && S.mats[pm].prop) {  /* Shade.cpp:512-516 */
            if constexpr (LITE) bail = true;
            else {
                Spec ptr, pst;
                propagate_in(S, S.mats[pm].prop, ray_at(ray, t), ray_at(ray, RT_INFINITY), pm_has_color ? pm_color.v : S.mats[pm].color, S.mats[pm].refr, ptr, pst);
                spec_store(P.f_fin, P.fcap, f, tr * pst);
                flags |= FR_FIN;
            }
        }
        if (!bail) {
            if (r_state == 0) flags |= FR_R_ZERO; else if (r_state == 1) flags |= FR_R_AMBIENT; else spec_store(P.f_R, P.fcap, f, R);
            if (tr_one) flags |= FR_TR_ONE; else spec_store(P.f_tr, P.fcap, f, tr);
        }
    }
    /* append this lane's shadow rays to the queue and (LITE) the frame to the slow list: one atomic per wave for both */
    int base;
    if constexpr (LITE) {
        if (bail) { n_shadow = 0; mask = 0; }
        int slot;
        wave_alloc2(n_shadow, bail ? 1 : 0, &P.counters[CTR_SHADOW], base, slot);
        if (bail) P.slow_v[slot] = f;
    } else
        base = wave_alloc(n_shadow, &P.counters[CTR_SHADOW]);
#ifdef RT_PHASE_PROF
    prof_v[0] = (int)(clock64() - prof_t0);
    for (int q = 0; q < 10; q++) {          /* per-wave maxima, in units of 1024 cycles (int counters: 16 overflowed on long frames) */
        int v = prof_v[q];
        for (int o = 32; o > 0; o >>= 1) { int w = __shfl_xor(v, o); v = w > v ? w : v; }
        if ((threadIdx.x & 63) == 0) atomicAdd(&P.counters[CTR_PROF + q], v >> 10);
    }
#endif
    if (!live || bail) continue;
    P.f_ids[(size_t)IDS_SHBASE * P.fcap + f] = base;
    P.f_ids[(size_t)IDS_SHMASK * P.fcap + f] = (int)mask;
    P.f_flags[f] = flags;
    if (n_shadow) {
        int k = 0;
        for (int li = 0; li < S.n_lights; li++) {
            if (!(mask & (1u << li))) continue;
            int s = base + k++;
            if (s >= P.scap) continue;          /* overflow is reported by the host from the counter */
            const RtLightDev &l = S.lights[li];
            D3 lo = d3(l.org[0], l.org[1], l.org[2]);
            if (l.type == RT_LIGHT_AREA) lo = lo + d3(l.vA[0], l.vA[1], l.vA[2]) * aux[2] + d3(l.vB[0], l.vB[1], l.vB[2]) * aux[3];
            d3_store(P.s_ray, P.scap, 0, s, Ppt);
            d3_store(P.s_ray, P.scap, 3, s, lo - Ppt);
            P.s_light[s] = li;
            P.s_root[s] = P.f_root[f];
        }
    }
  }
}

/* ------------------------------------------------------------------ ray order for the scheduled FULL kernels (no counterpart in the reference)
 * Secondary rays of a deep CSG scene arrive in the order of their parents and have little to do with their neighbours any
 * more; a wave of them visits 64 different parts of the program.  Levels that go to the scheduled FULL kernels directly are
 * traced in the order of a 32-bit key instead -- direction (octant and dominant axis, 5 bits), then the cell of the origin (9 bits
 * per axis of the level's own bounding box, Morton order) -- through the index lists those kernels read anyway (slow_v / slow_s).
 * Measured on S4 (ms per frame): unsorted 1643; cell / direction / fine cell 875; direction first 843; origin first 978; a cube-map
 * direction of 7 / 9 / 11 bits first 886 / 887 / 937 -- a coarse direction on top is what a wave's rays should share.
 * Every ray is still traced by the same code, and nothing downstream depends on the order in which a level's rays are traced.
 * k_sort_bounds: box of the origins (floats as ordered integers); k_sort_keys: keys and identity values, and the list's length. */
RT_DEV int float_ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
RT_DEV float ordered_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }
#define RT_SORT_BOUNDS_PER_THREAD 16
template <bool SHADOW>
__global__ void __launch_bounds__(256) k_sort_bounds(RtPipe P, int f0, int n_upper, int *box /*[6]: ordered min x y z, max x y z*/)
{
    __shared__ float s_lo[4][3], s_hi[4][3];
    int n = SHADOW ? P.counters[CTR_SNAP + 1] : n_upper;
    if (n > n_upper) n = n_upper;
    float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
    const int k0 = blockIdx.x * (256 * RT_SORT_BOUNDS_PER_THREAD) + threadIdx.x;
    for (int q = 0; q < RT_SORT_BOUNDS_PER_THREAD; q++) {
        const int k = k0 + q * 256;
        if (k >= n) break;
        const D3 o = SHADOW ? d3_fetch(P.s_ray, P.scap, 0, k) : d3_fetch(P.f_ray, P.fcap, 0, f0 + k);
        const float of[3] = { (float)o.x, (float)o.y, (float)o.z };
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], of[a]); hi[a] = fmaxf(hi[a], of[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        for (int m = 32; m > 0; m >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], m)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m)); }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a])), h = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
        if (l <= h) { atomicMin(&box[a], float_ordered(l)); atomicMax(&box[3 + a], float_ordered(h)); }
    }
}
RT_DEV unsigned spread3(unsigned v)         /* 10 bits -> every third bit */
{
    v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
template <bool SHADOW>
__global__ void __launch_bounds__(256) k_sort_keys(RtPipe P, int f0, int n_upper, const int *box, unsigned *keys, int *vals)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = SHADOW ? P.counters[CTR_SNAP + 1] : n_upper;      /* the level's shadow rays, frozen by k_level_snap */
    if (k == 0) P.counters[SHADOW ? CTR_SLOW_S : CTR_SLOW_V] = n < n_upper ? n : n_upper;
    if (k >= n_upper) return;
    unsigned key = 0xFFFFFFFFu;
    if (k < n) {
        const int e = SHADOW ? k : f0 + k;
        const D3 o = SHADOW ? d3_fetch(P.s_ray, P.scap, 0, e) : d3_fetch(P.f_ray, P.fcap, 0, e);
        const D3 d = SHADOW ? d3_fetch(P.s_ray, P.scap, 3, e) : d3_fetch(P.f_ray, P.fcap, 3, e);
        unsigned c[3];
        const float of[3] = { (float)o.x, (float)o.y, (float)o.z };
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float lo = ordered_float(box[a]), hi = ordered_float(box[3 + a]);
            float t = hi > lo ? (of[a] - lo) / (hi - lo) : 0.0f;
            t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            c[a] = (unsigned)(t * 511.0f);
        }
        const unsigned m = spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);      /* 27 bits */
        const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
        const unsigned dom = ax >= ay ? (ax >= az ? 0u : 2u) : (ay >= az ? 1u : 2u);
        const unsigned dir = ((d.x < 0.0) ? 1u : 0u) | ((d.y < 0.0) ? 2u : 0u) | ((d.z < 0.0) ? 4u : 0u) | (dom << 3);      /* 5 bits */
        key = (dir << 27) | m;
        if (key == 0xFFFFFFFFu) key = 0xFFFFFFFEu;
    }
    keys[k] = key;
    vals[k] = SHADOW ? k : f0 + k;
}

/* ------------------------------------------------------------------ k_fast (no counterpart in the reference)
 * Primary rays of the fast queue: VisualRay (Shade.cpp:401-520) for a ray whose hit list is ONE chain's -- affine,
 * two-root primitive, Material / Surface wrappers -- without the interpreter and without a hit arena: the (at most
 * three) hits stay in registers.  The ray is recomputed from the sample index (same arithmetic as the generator's
 * exact ray).  A sample that needs neither a shadow ray nor a reflection / refraction child is finished here: the light
 * loop of Shade.cpp:477-489 for its (nNoShadow) lights and the conversion to RGB, so it never becomes a frame.  Otherwise the lane
 * allocates a frame, stores the record k_visual would have stored and queues the shadow rays; k_shade takes it from
 * there.  The host only routes chains here that cannot need the FULL kernels (no torus / mesh, texture call-out,
 * propagator or spectral metal anywhere in the chain), so `bail` is an internal error, reported as an overflow. */
#define OVF_FAST_BAIL 64
template <class FetchE>
RT_DEV Spec light_loop(const RtDevScene &S, const D3 &Ppt, const D3 &cvN, const D3 &vE, const D3 &hn, const double F, const int rough, const int luster,
                       const Spec &color, Spec R, const double uL, const double vL_, const unsigned mask, const FetchE &fetch)
{
    for (int li = 0; li < S.n_lights; li++) {
        const RtLightDev &l = S.lights[li];
        D3 lo = d3(l.org[0], l.org[1], l.org[2]);
        if (l.type == RT_LIGHT_AREA) lo = lo + d3(l.vA[0], l.vA[1], l.vA[2]) * uL + d3(l.vB[0], l.vB[1], l.vB[2]) * vL_;
        D3 dir = lo - Ppt;
        if (dot(cvN, dir) <= 0.0 && luster != RT_MAT_TRANSPARENT) continue;
        D3 vL = normalize(dir);
        double fNL = dot(cvN, vL);
        Spec Eraw;
        if (l.no_shadow) Eraw = spec_div(spec_load(l.flux), 4.0 * RT_PI * dot(dir, dir));      /* Shade.cpp:315-318 */
        else {
            /* the shadow ray queued for this light: its slot follows from the mask stored with the frame, not from this
               loop's own count -- a light that was not queued (it cannot happen while the two predicates are the same
               code) contributes nothing instead of shifting the others' slots */
            if (!(mask & (1u << li))) continue;
            if (!fetch(__popc(mask & ((1u << li) - 1u)), Eraw)) continue;
        }
        Spec E = Eraw * fNL;
        if (spec_is_black(E)) continue;
        Spec refl = brdf(rough, luster, color, hn, vL, vE, cvN, F);
        R = R + refl * E;
    }
    return R;
}
RT_DEV void sample_deliver(const RtDevScene &S, const int root, const Spec &R, float *out_rgb, float *out_spec, const unsigned short *slot);

/* 512 threads = two waves per SIMD and 256 registers: measured on S0 4K / 64 spp, 512 / 768 / 1024 threads = 0.54 / 0.59 / 0.69 ms
   per launch of 5.4 M rays -- what the narrower builds spill (480 B and 700 B per lane) costs more than the waves they add */
#ifndef RT_FAST_THREADS
#define RT_FAST_THREADS 512
#endif
/* MODE 1: the workgroup stages everything a fast ray looks up per lane -- chain records, the blue-noise tile and its two
   permutations, materials, surfaces -- in LDS (rt_scene_upload checks that it fits: S0 needs 117 KB); MODE 0: nothing is
   staged.  The loads below then are LDS instructions: the pointers are derived from the LDS base inside this function. */
/* psModel->Intersect for a bare chain: affine, two-root primitive, wrappers; the (at most three) hits in registers */
RT_DEV void fast_chain_hits(const RtDevScene &S, const RtChain &ch, const DRay &ray, RegEmit &H)
{
    DRay r = ray;
    if (ch.n_affine) {                                      /* AffineSolid::Intersect, Solids.cpp:899-913 */
        DRay nr;
        nr.dir = mat_vec(ch.minv, r.dir);
        nr.org = mat_vec(ch.minv, r.org - d3(ch.tr[0], ch.tr[1], ch.tr[2]));
        r = nr;
    }
    Work w;
    if (ch.prim_op == OP_QUADRIC) quadric_roots(ch.p[0], ch.p[1], ch.p[2], ch.i0, r, w);
    else if (ch.prim_op == OP_POLY) poly_roots(S.planes + 3 * (size_t)ch.plane0, ch.i0, r, w);
    else slab_roots(r, w);
    H.minv_own = (ch.test_box & RT_CHAIN_OWN_NORMAL) ? ch.minv : (const double *)0;
    hit_list2_core(w, H, (uint16_t)ch.prim, ch.leaf_affine, &ch);
}
RT_HD size_t fast_lds_bytes(int n_chains, int n_mats, int n_surfs)
{
    return (size_t)n_chains * sizeof(RtChain) + (size_t)2 * RT_BLUE_N * sizeof(double) + (size_t)2 * RT_BLUE_N * sizeof(int32_t) +
           (size_t)n_mats * sizeof(RtMaterialDev) + (size_t)n_surfs * sizeof(RtSurfaceDev) + 64;
}
template <int MODE>
__global__ void __attribute__((flatten)) __launch_bounds__(RT_FAST_THREADS, 1) k_fast(const RtDevScene Sv, RtSampleGrid g, RtPipe P, int64_t first, float *out_rgb, float *out_spec)
{
    RtDevScene SL = Sv;
    const RtChain *chains = Sv.chains;
    if constexpr (MODE == 1) {
        extern __shared__ double rt_dyn_lds[];
        double *cur = rt_dyn_lds;
        chains = lds_stage(cur, Sv.chains, Sv.n_chains);
        SL.blue = lds_stage(cur, Sv.blue, 2 * RT_BLUE_N);
        SL.qi2 = lds_stage(cur, Sv.qi2, RT_BLUE_N);
        SL.qi3 = lds_stage(cur, Sv.qi3, RT_BLUE_N);
        SL.mats = lds_stage(cur, Sv.mats, Sv.n_mats);
        SL.surfs = lds_stage(cur, Sv.surfs, Sv.n_surfs);
        __syncthreads();
    }
    const RtDevScene &S = SL;
    int n = P.counters[CTR_FAST];
    if (n > P.ncap) n = P.ncap;
    int *const work = &P.counters[CTR_WORK_F];
    const int wlane = threadIdx.x & 63;
  int wnext = 0, wlimit = 0;
  for (;;) {
    if (wnext >= wlimit) {
        int b = 0;
        if (wlane == 0) b = atomicAdd(work, RT_CLAIM);
        b = __shfl(b, 0, 64);
        if (b >= n) break;
        wnext = b; wlimit = b + RT_CLAIM < n ? b + RT_CLAIM : n;
    }
    const int e = wnext + wlane;
    wnext += 64;
    const bool live = e < n;
    Walk B;
    RegSink K;
    B.n_shadow = 0; B.mask = 0; B.bail = false; B.flags = 0;
    double aux[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };    /* only the light pair is ever read here (no aperture camera, no texture call-out on this path) */
    int k = 0;
    bool need_frame = false, hand_over = false;
    DRay ray;
    if (live) {
        k = nt_load(&P.fast_q[e]);
        const int code = nt_load(&P.fast_q[(size_t)P.ncap + e]);
        const int64_t sidx = first + k;
        const uint32_t tile = (uint32_t)(sidx / RT_BLUE_N);
        {   /* FastBlueSamplingTile::Render, Sampling.cpp:498-517 + CastFunction, RayTracer2020.cpp:20-31: as sample_at / root_ray */
            const int i = (int)(sidx - (int64_t)tile * RT_BLUE_N);
            const uint32_t txu = tile / (uint32_t)g.tiles_y_local;
            const int tx = (int)txu, ty = (int)(tile - txu * (uint32_t)g.tiles_y_local) + g.tile_y0;
            const double vtx = (double)tx * (double)RT_BLUE_NSQRT, vty = (double)ty * (double)RT_BLUE_NSQRT;
            const double vx = S.blue[2 * i] * (double)RT_BLUE_NSQRT + vtx, vy = S.blue[2 * i + 1] * (double)RT_BLUE_NSQRT + vty;
            const double v0 = (vx + g.adjust_x) / g.size, v1 = (vy + g.adjust_y) / g.size;
            if (S.aux_light) {
                const int q2 = S.qi2[i % RT_BLUE_N];
                aux[2] = 2.0 * S.blue[2 * q2] - 1.0;
                aux[3] = 2.0 * S.blue[2 * q2 + 1] - 1.0;
            }
            double uC = 0.0, vC = 0.0;
            if (S.cam.type == RT_CAM_APERTURE) { const int q3 = S.qi3[i]; uC = 2.0 * S.blue[2 * q3] - 1.0; vC = 2.0 * S.blue[2 * q3 + 1] - 1.0; }
            ray = camera_emit(S.cam, v0, v1, uC, vC);
        }
        RegEmit H;
        H.L = 0; H.minv_own = 0;
        if (code >= 0) {
            /* ONE chain can be touched: its list is the ray's hit list */
            fast_chain_hits(S, chains[code], ray, H);
        } else {
            /* Several chains can be touched.  Their lists would be merged by the union's density lattice (Solids.cpp:1014-
               1058) and the walk would stop at the first boundary (Shade.cpp:443).  When every list starts ahead of the
               origin (no t = 0 hit: the origin is outside them all), the merged list starts with the nearest of their first
               hits, entering that hit's own material (the other lists still are in vacuum there) -- provided that material
               is denser than vacuum (else the merge drops the hit) and no second list has a first hit at exactly the same
               distance (a tie takes the left record and the denser material).  If the walk then stops at that hit, nothing
               behind it matters: the ray is finished from that one hit.  Anything else is handed to the interpreter. */
            const unsigned short *tl = g.tile_cand + (size_t)tile * RT_TILE_CAND;
            double t_best = RT_INFINITY, t_next = RT_INFINITY;
            for (unsigned m = (unsigned)code & 0x7FFFFFFFu; m; m &= m - 1) {     /* bit q: entry q of the tile's list (RT_TILE_CAND <= 31) */
                const RtChain &ch = chains[S.root_fast[tl[__builtin_ctz(m)]]];
                RegEmit T;
                fast_chain_hits(S, ch, ray, T);
                if (T.L == 0) continue;
                const double t = T.h0.t;
                if (!(t > 0.0)) { hand_over = true; break; }
                if (t < t_best) { t_next = t_best; t_best = t; H.h0 = T.h0; H.minv_own = T.minv_own; H.L = 1; }
                else if (t < t_next) t_next = t;
            }
            if (H.L == 1 && (t_next == t_best || !(S.mats[H.h0.mat].density > 0.0))) hand_over = true;
        }
        if (!hand_over) {
            visual_walk<true>(S, ray, aux, H, H.L, B, K);
            if (code < 0 && H.L == 1 && !B.bail && !(B.flags & FR_BOUNDARY)) hand_over = true;      /* the walk went on past the hit */
        }
        if (hand_over) { B.n_shadow = 0; B.mask = 0; }
        else if (B.bail) { atomicAdd(&P.counters[CTR_OVERFLOW], 1); atomicOr(&P.counters[CTR_OVERFLOW + 1], OVF_FAST_BAIL); B.n_shadow = 0; B.mask = 0; }
        else if ((B.flags & FR_BOUNDARY) && (B.n_shadow > 0 || K.rough == RT_SURF_POLISHED)) need_frame = true;
        else {
            Spec R = B.R;
            if (B.flags & FR_BOUNDARY)
                R = light_loop(S, B.P, K.N, K.E, K.HN, K.F, K.rough, K.luster, K.color, R, aux[2], aux[3], 0u, [](int, Spec &) { return false; });
            sample_deliver(S, k, R, out_rgb, out_spec, g.rgb_slot);
        }
    }
    /* the lanes that could not finish become level-0 frames behind the generator's: traced ones (FR_TRACED: the record
       k_visual would have stored, for k_shade) and root frames for the interpreter (k_visual runs behind this kernel) */
    const unsigned long long bt = __ballot(need_frame), bh = __ballot(hand_over);
    if ((bt | bh) == 0ull) continue;
    const int f = wave_alloc((need_frame || hand_over) ? 1 : 0, &P.counters[CTR_FRAMES]);
    const int base = wave_alloc(need_frame ? B.n_shadow : 0, &P.counters[CTR_SHADOW]);
    if (wlane == 0 && bt) atomicAdd(&P.counters[CTR_FAST + 1], __popcll(bt));
    if (!(need_frame || hand_over) || f >= P.fcap) continue;               /* a full pool is reported by the host from the counter */
    aux[4] = aux[2]; aux[5] = aux[3];                       /* the micro-facet pair follows the light's permutation (Sampling.cpp:512-515) */
    if (hand_over) { root_frame(S, P, f, k, ray, aux, 0); continue; }
    aux_store(S, P, k, aux);
    P.f_parent[f] = -1;
    P.f_root[f] = k;
    FrameSink FS = { P, f, FR_TRACED };
    if (B.flags & FR_BOUNDARY) FS.boundary(B.P, K.N, K.E, K.HN, K.fNE, K.F, K.fNT, K.eta, K.rough, K.luster, K.mat, K.color, K.color_is_mat);
    const int flags = frame_finish<true>(P, f, B, FS.flags);
    P.f_ids[(size_t)IDS_SHBASE * P.fcap + f] = base;
    P.f_ids[(size_t)IDS_SHMASK * P.fcap + f] = (int)B.mask;
    P.f_flags[f] = flags;
    if (B.n_shadow) shadow_append(S, P, B.P, B.mask, base, k, aux);
  }
}

/* ------------------------------------------------------------------ the fast path in two kernels (option fast_split)
 * k_fast above keeps a ray's whole life in one kernel: f64 geometry (camera ray, chain, hits, walk) and f32 shading (light loop,
 * BRDF, sRGB) together need the whole register budget of two waves per SIMD and still spill, and its waves spend two thirds of
 * their cycles waiting.  Split at the boundary the walk stops at, each half fits half the registers and runs at twice the
 * occupancy:
 *   k_fast_hit    sample -> camera ray -> chain(s) -> <= 3 hits in registers -> which hit the walk of Shade.cpp:433-511 stops
 *                 at (the same decisions in the same order as visual_walk, none of its arithmetic) -> a 44-byte record: sample,
 *                 distance, the hit's normal after the late transform (Shade.cpp:264-284), the materials on either side, the
 *                 surface.  Rays without a boundary end here (black, or a NOSHADE medium's colour); rays the interpreter must
 *                 see become root frames, as in k_fast.
 *   k_fast_shade  record -> the ray again from the sample index (same arithmetic) -> visual_walk over the two-entry list
 *                 "inside material a at t = 0; boundary hit" (which is where the full walk stood) -> light loop -> RGB, or the
 *                 frame record and shadow rays k_fast would have left for k_shade.
 * The record goes through a compact queue (one atomic per wave), never through the frame pool.  Same results as k_fast and as
 * the interpreter (test_fast_path_is_invisible). */
#ifndef RT_FASTA_THREADS
#define RT_FASTA_THREADS 768
#endif
#ifndef RT_FAST_CLAIM
#define RT_FAST_CLAIM 512       /* queue entries a wave of the two kernels claims per atomic */
#endif
#ifndef RT_FASTB_THREADS
#define RT_FASTB_THREADS 512
#endif
RT_HD size_t fast_hit_lds_bytes(int n_chains, int n_mats)
{
    return (size_t)n_chains * sizeof(RtChain) + (size_t)2 * RT_BLUE_N * sizeof(double) + (size_t)n_mats * sizeof(RtMaterialDev) + 64;
}
RT_HD size_t fast_shade_lds_bytes(int n_mats, int n_surfs, int n_lights)
{
    return (size_t)2 * RT_BLUE_N * sizeof(double) + (size_t)RT_BLUE_N * sizeof(int32_t) + (size_t)n_mats * sizeof(RtMaterialDev) + (size_t)n_surfs * sizeof(RtSurfaceDev) +
           (size_t)n_lights * sizeof(RtLightDev) + 64;
}
/* FastBlueSamplingTile::Render, Sampling.cpp:498-517 + CastFunction, RayTracer2020.cpp:20-31 for sample sidx of the grid: as sample_at / root_ray
   (no aperture camera on the fast path: the lens pair is not needed) */
RT_DEV DRay fast_ray(const RtDevScene &S, const RtSampleGrid &g, const int64_t first, const int k, uint32_t &tile, int &i)
{
    tile = tile_of(first, k, i);
    const uint32_t txu = tile / (uint32_t)g.tiles_y_local;
    const int tx = (int)txu, ty = (int)(tile - txu * (uint32_t)g.tiles_y_local) + g.tile_y0;
    const double vtx = (double)tx * (double)RT_BLUE_NSQRT, vty = (double)ty * (double)RT_BLUE_NSQRT;
    const double vx = S.blue[2 * i] * (double)RT_BLUE_NSQRT + vtx, vy = S.blue[2 * i + 1] * (double)RT_BLUE_NSQRT + vty;
    const double v0 = (vx + g.adjust_x) / g.size, v1 = (vy + g.adjust_y) / g.size;
    double uC = 0.0, vC = 0.0;
    if (S.cam.type == RT_CAM_APERTURE) {        /* the lens pair of sample_at: aux[0], aux[1] */
        const int q3 = S.qi3[i];
        uC = 2.0 * S.blue[2 * q3] - 1.0; vC = 2.0 * S.blue[2 * q3 + 1] - 1.0;
    }
    return camera_emit(S.cam, v0, v1, uC, vC);
}
RT_DEV void fast_aux(const RtDevScene &S, const int i, double *aux)
{
    for (int q = 0; q < 6; q++) aux[q] = 0.0;
    if (S.aux_light) {
        const int q2 = S.qi2[i % RT_BLUE_N];
        aux[2] = 2.0 * S.blue[2 * q2] - 1.0;
        aux[3] = 2.0 * S.blue[2 * q2 + 1] - 1.0;
    }
    aux[4] = aux[2]; aux[5] = aux[3];       /* the micro-facet pair follows the light's permutation (Sampling.cpp:512-515) */
}
template <int MODE>
__global__ void __attribute__((flatten)) __launch_bounds__(RT_FASTA_THREADS, 1) k_fast_hit(const RtDevScene Sv, RtSampleGrid g, RtPipe P, int64_t first, float *out_rgb, float *out_spec)
{
    RtDevScene SL = Sv;
    const RtChain *chains = Sv.chains;
    if constexpr (MODE == 1) {
        extern __shared__ double rt_dyn_lds[];
        double *cur = rt_dyn_lds;
        chains = lds_stage(cur, Sv.chains, Sv.n_chains);
        SL.blue = lds_stage(cur, Sv.blue, 2 * RT_BLUE_N);
        SL.mats = lds_stage(cur, Sv.mats, Sv.n_mats);
        __syncthreads();
    }
    const RtDevScene &S = SL;
    int n = P.counters[CTR_FAST];
    if (n > P.ncap) n = P.ncap;
    int *const work = &P.counters[CTR_WORK_F];
    const int wlane = threadIdx.x & 63;
    /* waves claim RT_FAST_CLAIM entries per atomic.  The claims are not free -- k_fast's 280 k same-address atomics per launch (claims
       of 128) cost it 0.55 of its 3.3 ms -- but dealing the chunks out statically (wave w takes chunks w, w + W, ...) was slower than
       claims of 512: 2.65 against 2.44 ms per launch; the entries' costs differ enough for the balance to matter */
  int wnext = 0, wlimit = 0;
  for (;;) {
    if (wnext >= wlimit) {
        int b = 0;
        if (wlane == 0) b = atomicAdd(work, RT_FAST_CLAIM);
        b = __shfl(b, 0, 64);
        if (b >= n) break;
        wnext = b; wlimit = b + RT_FAST_CLAIM < n ? b + RT_FAST_CLAIM : n;
    }
    const int e = wnext + wlane;
    wnext += 64;
    const bool live = e < n;
    int k = 0;
    bool hand_over = false, boundary = false;
    double bt = 0.0;
    D3 bn = d3(0, 0, 0);
    int bids = 0, bsurf = 0;
    if (live) {
        k = nt_load(&P.fast_q[e]);
        const int code = nt_load(&P.fast_q[(size_t)P.ncap + e]);
        uint32_t tile;
        int isamp;
        const DRay ray = fast_ray(S, g, first, k, tile, isamp);
        RegEmit H;
        H.L = 0; H.minv_own = 0;
        /* (one loop for both cases -- one copy of the chain code, 3400 instructions instead of 4800 -- was measured and is slower: 2.53
           against 2.44 ms per launch pair; the shared loop keeps both lists alive) */
        if (code >= 0) fast_chain_hits(S, chains[code], ray, H);       /* ONE chain can be touched: its list is the ray's hit list */
        else {
            /* several: the nearest first hit decides when every list starts ahead of the origin (see k_fast) */
            const unsigned short *tl = g.tile_cand + (size_t)tile * RT_TILE_CAND;
            double t_best = RT_INFINITY, t_next = RT_INFINITY;
            for (unsigned m = (unsigned)code & 0x7FFFFFFFu; m; m &= m - 1) {     /* bit q: entry q of the tile's list (RT_TILE_CAND <= 31) */
                const RtChain &ch = chains[S.root_fast[tl[__builtin_ctz(m)]]];
                RegEmit T;
                fast_chain_hits(S, ch, ray, T);
                if (T.L == 0) continue;
                const double t = T.h0.t;
                if (!(t > 0.0)) { hand_over = true; break; }
                if (t < t_best) { t_next = t_best; t_best = t; H.h0 = T.h0; H.minv_own = T.minv_own; H.L = 1; }
                else if (t < t_next) t_next = t;
            }
            if (H.L == 1 && (t_next == t_best || !(S.mats[H.h0.mat].density > 0.0))) hand_over = true;
        }
        bool bail = false, final = false;
        int pm = 0;
        if (!hand_over) {
            /* where the walk of Shade.cpp:424-516 stops: visual_walk<LITE>'s decisions, in its order */
            int i = 0;
            if (H.L > 0 && !(H.h0.t > 0.0)) { pm = H.h0.mat; i = 1; }
            for (; i < H.L; i++) {
                const HitRec hr = H.get(i);
                if (hr.ntex) { bail = true; break; }                    /* (transform_normals<LITE>) */
                const RtMaterialDev &mp = S.mats[pm];
                if (mp.luster == RT_MAT_NOSHADE) { final = true; break; }
                if (mp.prop) { bail = true; break; }
                const RtMaterialDev &mh = S.mats[hr.mat];
                if ((mp.refr != mh.refr) || (mh.luster != RT_MAT_TRANSPARENT)) {
                    if (spectral_metal(mh.luster)) { bail = true; break; }
                    boundary = true;
                    bt = hr.t;
                    bn = hr.n;                                          /* TransformNormals, Shade.cpp:264-284 */
                    if (H.minv_own) bn = covec_mat(bn, H.minv_own);
                    else for (int pa = (int)hr.leaf_affine - 1; pa >= 0; pa = S.affines[pa].link) bn = covec_mat(bn, S.affines[pa].minv);
                    bids = pm | ((int)hr.mat << 16);
                    bsurf = hr.surf;
                    break;
                }
                pm = hr.mat;
            }
            if (!bail && !final && !boundary && S.mats[pm].prop) bail = true;      /* Shade.cpp:512-516: a propagating medium to infinity */
            if (code < 0 && H.L == 1 && !bail && !boundary) hand_over = true;      /* the walk went on past the one hit that was kept */
        }
        if (hand_over) boundary = false;
        else if (bail) { atomicAdd(&P.counters[CTR_OVERFLOW], 1); atomicOr(&P.counters[CTR_OVERFLOW + 1], OVF_FAST_BAIL); boundary = false; }
        else if (!boundary) sample_deliver(S, k, final ? spec_load(S.mats[pm].color) : spec_const(0.0f), out_rgb, out_spec, g.rgb_slot);
    }
    /* the record sits at the queue entry's own index (-1 = none): compacting the records would cost one same-address atomic per
       64 rays -- measured: 560 k of them per launch made this kernel 1.6 x slower than all of k_fast -- and buys k_fast_shade the
       quarter of its lanes whose rays had no boundary */
    if (live) {
        nt_store(&P.fr_k[e], boundary ? k : -1);
        if (boundary) {
            nt_store(&P.fr_t[e], bt);
            d3_store(P.fr_n, P.ncap, 0, e, bn);
            nt_store(&P.fr_ids[e], bids);
            nt_store(&P.fr_ids[(size_t)P.ncap + e], bsurf);
        }
    }
    const unsigned long long bh = __ballot(hand_over);
    if (bh) {       /* root frames for the interpreter (k_visual runs behind the fast kernels) */
        const int f = wave_alloc(hand_over ? 1 : 0, &P.counters[CTR_FRAMES]);
        if (hand_over && f < P.fcap) {      /* rare: the ray is computed again rather than kept alive across the whole iteration */
            double aux[6];
            uint32_t tile;
            int isamp;
            const DRay ray = fast_ray(S, g, first, k, tile, isamp);
            fast_aux(S, isamp, aux);
            root_frame(S, P, f, k, ray, aux, 0);
        }
    }
  }
}

struct TwoHits {                /* "inside material a at t = 0" + the boundary hit, its normal already transformed */
    HitRec h0, h1;
    RT_DEV HitRec get(int i) const
    {
        HitRec r;
        r.t = i ? h1.t : h0.t; r.n.x = i ? h1.n.x : h0.n.x; r.n.y = i ? h1.n.y : h0.n.y; r.n.z = i ? h1.n.z : h0.n.z;
        r.leaf_affine = 0; r.surf = i ? h1.surf : h0.surf; r.mat = i ? h1.mat : h0.mat; r.prim = 0; r.ntex = 0; r.pad_ = 0; r.texpack = 0;
        return r;
    }
    RT_DEV const double *own_minv() const { return 0; }
};
RT_DEV TwoHits fast_record(const RtPipe &P, const int e)
{
    TwoHits H;
    const int ids = nt_load(&P.fr_ids[e]);
    H.h0.t = 0.0; H.h0.n = d3(0.0, 0.0, 0.0); H.h0.mat = (uint16_t)(ids & 0xFFFF); H.h0.surf = 0;
    H.h1.t = nt_load(&P.fr_t[e]);
    H.h1.n = d3_fetch(P.fr_n, P.ncap, 0, e);
    H.h1.mat = (uint16_t)((unsigned)ids >> 16);
    H.h1.surf = (uint16_t)nt_load(&P.fr_ids[(size_t)P.ncap + e]);
    return H;
}
template <int MODE>
__global__ void __attribute__((flatten)) __launch_bounds__(RT_FASTB_THREADS, 1) k_fast_shade(const RtDevScene Sv, RtSampleGrid g, RtPipe P, int64_t first, float *out_rgb, float *out_spec)
{
    RtDevScene SL = Sv;
    if constexpr (MODE == 1) {
        extern __shared__ double rt_dyn_lds[];
        double *cur = rt_dyn_lds;
        SL.blue = lds_stage(cur, Sv.blue, 2 * RT_BLUE_N);
        SL.qi2 = lds_stage(cur, Sv.qi2, RT_BLUE_N);
        SL.mats = lds_stage(cur, Sv.mats, Sv.n_mats);
        SL.surfs = lds_stage(cur, Sv.surfs, Sv.n_surfs);
        SL.lights = lds_stage(cur, Sv.lights, Sv.n_lights);
        __syncthreads();
    }
    const RtDevScene &S = SL;
    int n = P.counters[CTR_FAST];           /* one (possibly empty) record per entry of the fast queue */
    if (n > P.ncap) n = P.ncap;
    int *const work = &P.counters[CTR_WORK_F2];
    const int wlane = threadIdx.x & 63;
    /* waves claim RT_FAST_CLAIM entries per atomic.  The claims are not free -- k_fast's 280 k same-address atomics per launch (claims
       of 128) cost it 0.55 of its 3.3 ms -- but dealing the chunks out statically (wave w takes chunks w, w + W, ...) was slower than
       claims of 512: 2.65 against 2.44 ms per launch; the entries' costs differ enough for the balance to matter */
  int wnext = 0, wlimit = 0;
  for (;;) {
    if (wnext >= wlimit) {
        int b = 0;
        if (wlane == 0) b = atomicAdd(work, RT_FAST_CLAIM);
        b = __shfl(b, 0, 64);
        if (b >= n) break;
        wnext = b; wlimit = b + RT_FAST_CLAIM < n ? b + RT_FAST_CLAIM : n;
    }
    const int e = wnext + wlane;
    wnext += 64;
    const bool live = e < n;
    int k = 0, n_shadow = 0;
    bool need_frame = false;
    if (live) k = nt_load(&P.fr_k[e]);
    if (live && k >= 0) {
        const TwoHits H = fast_record(P, e);
        uint32_t tile;
        int isamp;
        double aux[6];
        const DRay ray = fast_ray(S, g, first, k, tile, isamp);
        fast_aux(S, isamp, aux);
        Walk B;
        RegSink K;
        visual_walk<true>(S, ray, aux, H, 2, B, K);
        if (B.bail || !(B.flags & FR_BOUNDARY)) {       /* cannot happen: k_fast_hit found the boundary with the same tests */
            atomicAdd(&P.counters[CTR_OVERFLOW], 1); atomicOr(&P.counters[CTR_OVERFLOW + 1], OVF_FAST_BAIL);
        } else if (B.n_shadow > 0 || K.rough == RT_SURF_POLISHED) { need_frame = true; n_shadow = B.n_shadow; }
        else {
            const Spec R = light_loop(S, B.P, K.N, K.E, K.HN, K.F, K.rough, K.luster, K.color, B.R, aux[2], aux[3], 0u, [](int, Spec &) { return false; });
            sample_deliver(S, k, R, out_rgb, out_spec, g.rgb_slot);
        }
    }
    /* The lanes that need shadow rays or children become traced level-0 frames (FR_TRACED) for k_shade, as in k_fast.  They walk
       again, this time into the frame record itself (what the first walk found is not kept alive across the light loop of the
       other lanes: that is what lets this kernel run at twice k_fast's occupancy). */
    const unsigned long long bt = __ballot(need_frame);
    if (bt == 0ull) continue;
    const int f = wave_alloc(need_frame ? 1 : 0, &P.counters[CTR_FRAMES]);
    const int base = wave_alloc(n_shadow, &P.counters[CTR_SHADOW]);
    if (wlane == 0) atomicAdd(&P.counters[CTR_FAST + 1], __popcll(bt));
    if (!need_frame || f >= P.fcap) continue;
    {
        const TwoHits H = fast_record(P, e);
        uint32_t tile;
        int isamp;
        double aux[6];
        const DRay ray = fast_ray(S, g, first, k, tile, isamp);
        fast_aux(S, isamp, aux);
        aux_store(S, P, k, aux);
        P.f_parent[f] = -1;
        P.f_root[f] = k;
        Walk B;
        FrameSink FS = { P, f, FR_TRACED };
        visual_walk<true>(S, ray, aux, H, 2, B, FS);
        const int flags = frame_finish<true>(P, f, B, FS.flags);
        P.f_ids[(size_t)IDS_SHBASE * P.fcap + f] = base;
        P.f_ids[(size_t)IDS_SHMASK * P.fcap + f] = (int)B.mask;
        P.f_flags[f] = flags;
        if (B.n_shadow) shadow_append(S, P, B.P, B.mask, base, k, aux);
    }
  }
}

/* ------------------------------------------------------------------ k_shadow: RayTracer::LightRay, Shade.cpp:301-359
 * shadow rays [0, n) (LITE) or those on the slow list (FULL) */
/* ------------------------------------------------------------------ k_mesh_cross: the mesh walk in a kernel of its own
 * The FULL kernels carry the whole interpreter, shading and the rare features: 256 registers, two waves per SIMD -- and a
 * mesh walk is a chain of dependent 128-byte node reads that wants many rays in flight.  For scenes whose program holds one
 * mesh instruction, the rays of a slow list (what the LITE launch handed over) are walked here first: the ray is taken into
 * the mesh's frame by the same affines in the same order as the interpreter would (bit-identical), the same walk runs with
 * a stack of RT_MESHPRE_STACK entries in LDS, and every crossing is stored as it is found (no private arrays, no sort: the
 * FULL kernel's mesh_hits reads them back, orders them and builds the hit list as it would have after walking itself).
 * A ray whose walk runs out of stack is marked -1 and walked again in line. */
#define RT_MESHPRE_STACK   32          /* LDS words per lane: node stack from the bottom, leaf queue from the top */
#define RT_MESHPRE_THREADS 256
#ifndef RT_MESHPRE_NODES
#define RT_MESHPRE_NODES   2
#endif
#ifndef RT_MESHPRE_WAVES
#define RT_MESHPRE_WAVES   4           /* waves per SIMD the kernel is built for */
#endif
#define RT_MESHPRE_ROUND   128         /* triangle tests a wave shares out per round */
#define RT_MESHPRE_LDS     ((RT_MESHPRE_STACK * RT_MESHPRE_THREADS + (RT_MESHPRE_THREADS / 64) * (64 + 2 * RT_MESHPRE_ROUND)) * (int)sizeof(int))
/* A crossing found by any lane for the ray of lane `owner` of the same wave: counted in LDS, stored in the owner's record
   while it has room (a 17th crossing makes the FULL kernel walk that ray again in line -- rare). */
RT_DEV void mesh_cross_emit(const RtPipe &P, int *wave_nh, const int slot0, const int owner, const double t, const int q)
{
    const int k = atomicAdd(&wave_nh[owner], 1);
    if (k < RT_MESH_MAXHITS) {
        P.mx_t[(size_t)k * P.mxcap + slot0 + owner] = t;
        P.mx_q[(size_t)k * P.mxcap + slot0 + owner] = q;
    }
}
/* n_all < 0: over the slow list (entry = position in the list); n_all >= 0 (P.mx_all): over every ray of the level -- visual
   frames f0 .. f0 + n_all (entry f - f0), or all the level's shadow rays (their count is on the device; entry = ray index) */
template <bool SHADOW>
__global__ void __launch_bounds__(RT_MESHPRE_THREADS, RT_MESHPRE_WAVES) k_mesh_cross(const RtDevScene S, RtPipe P, int f0, int n_all)
{
    extern __shared__ int mx_lds[];
    const bool all = n_all >= 0;
    const int n = all ? (SHADOW ? P.counters[CTR_SNAP + 1] : n_all) : P.counters[SHADOW ? CTR_SLOW_S : CTR_SLOW_V];
    int *const work = &P.counters[SHADOW ? CTR_WORK_MS : CTR_WORK_MV];
    const int wlane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *const stack = mx_lds + threadIdx.x;                        /* entry i at [i * RT_MESHPRE_THREADS] */
    int *const wave_nh = mx_lds + RT_MESHPRE_STACK * RT_MESHPRE_THREADS + wave * (64 + 2 * RT_MESHPRE_ROUND);
    int *const buf_q = wave_nh + 64, *const buf_o = buf_q + RT_MESHPRE_ROUND;
#define MSTK(i) stack[(i) * RT_MESHPRE_THREADS]
    const RtMeshInfo mi = S.meshes[S.mesh_pre_mesh];
    const RtMeshNode *nodes = S.mesh_nodes + mi.node0;
    const RtMeshTri *tris = S.mesh_tris + mi.tri0;
    for (;;) {
        int b = 0;
        if (wlane == 0) b = atomicAdd(work, 64);
        b = __shfl(b, 0, 64);
        if (b >= n) break;
        const int slot = b + wlane;
        const bool live = slot < n;
        DRay ray;
        ray.org = d3(0.0, 0.0, 0.0); ray.dir = d3(0.0, 0.0, 1.0);
        if (live) {
            if (SHADOW) { const int s = all ? slot : P.slow_s[slot]; ray.org = d3_fetch(P.s_ray, P.scap, 0, s); ray.dir = d3_fetch(P.s_ray, P.scap, 3, s); }
            else { const int f = all ? f0 + slot : P.slow_v[slot]; ray.org = d3_fetch(P.f_ray, P.fcap, 0, f); ray.dir = d3_fetch(P.f_ray, P.fcap, 3, f); }
            for (int k = 0; k < S.mesh_pre_naff; k++) {                 /* AffineSolid::Intersect (Solids.cpp:899-913), as OP_PUSH_AFFINE does it */
                const RtAffine &a = S.affines[S.mesh_pre_aff[k]];
                DRay nr;
                nr.dir = mat_vec(a.minv, ray.dir);
                nr.org = mat_vec(a.minv, ray.org - d3(a.tr[0], a.tr[1], a.tr[2]));
                ray = nr;
            }
            if (S.mesh_pre_chain >= 0) {                                /* ... and as OP_CHAIN does it */
                const RtChain &ch = S.chains[S.mesh_pre_chain];
                if (ch.n_affine) {
                    DRay nr;
                    nr.dir = mat_vec(ch.minv, ray.dir);
                    nr.org = mat_vec(ch.minv, ray.org - d3(ch.tr[0], ch.tr[1], ch.tr[2]));
                    ray = nr;
                }
            }
        }
        wave_nh[wlane] = 0;
        /* phase 1, per lane: the node walk; leaf slots that are hit are queued (first triangle | (count - 1) << 27) */
        int sp = 0, lq = RT_MESHPRE_STACK, ntri = 0;
        bool ok = true;
        if (live) { MSTK(0) = 0; sp = 1; MESH_COUNT(0, 1); }
        MeshRay mr;
        mr.o[0] = ray.org.x; mr.o[1] = ray.org.y; mr.o[2] = ray.org.z;
        {
            const double d[3] = { ray.dir.x, ray.dir.y, ray.dir.z };
#pragma unroll
            for (int a = 0; a < 3; a++) { mr.par[a] = d[a] == 0.0; mr.inv[a] = mr.par[a] ? 0.0 : 1.0 / d[a]; }
        }
        /* RT_MESHPRE_NODES nodes per iteration: the order in which pending nodes are visited does not matter (every crossing
           is collected), so a lane with several pending nodes has as many node reads in flight and needs that many fewer
           iterations -- the walk waits on memory, and reads in flight are what it is short of */
        while (sp > 0) {
            int idx[RT_MESHPRE_NODES];
            int nn = 0;
#pragma unroll
            for (int h = 0; h < RT_MESHPRE_NODES; h++) { const bool have = sp > 0; idx[h] = have ? MSTK(sp - 1) : idx[0]; if (have) { --sp; nn++; } }
            RtMeshNode nds[RT_MESHPRE_NODES];
#pragma unroll
            for (int h = 0; h < RT_MESHPRE_NODES; h++) nds[h] = nodes[idx[h]];
            MESH_COUNT(1, nn);
            MESH_WAVE_COUNT(4);
            if (sp + 4 * RT_MESHPRE_NODES > lq) {                  /* the column is full: this lane tests what it has queued, alone */
                while (lq < RT_MESHPRE_STACK) {
                    const int e = MSTK(lq); lq++;
                    const int q0 = e & 0x07FFFFFF, c = (e >> 27) + 1;
                    for (int k = 0; k < c; k++) {
                        double t;
                        MESH_COUNT(3, 1);
                        if (mesh_triangle(tris[q0 + k], ray, t)) { MESH_COUNT(6, 1); mesh_cross_emit(P, wave_nh, b, wlane, t, q0 + k); }
                    }
                }
                ntri = 0;
                if (sp + 4 * RT_MESHPRE_NODES > lq) { ok = false; break; }
            }
#pragma unroll
            for (int h = 0; h < RT_MESHPRE_NODES; h++) {
                const RtMeshNode &nd = nds[h];
                if (h >= nn) break;
                bool hit[4];
#pragma unroll
                for (int k = 0; k < 4; k++) hit[k] = mesh_slot_hit(nd, k, mr);
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const int c = nd.child[k];
                    if (!hit[k]) continue;
                    if (c >= 0) { MSTK(sp) = c; sp++; }
                    else if (c != RT_MESH_EMPTY) { --lq; MSTK(lq) = ~c; ntri += ((~c) >> 27) + 1; }
                }
            }
        }
        /* phase 2, the whole wave: the queued triangle tests of all 64 rays are dealt out evenly.  A skimming ray queues a
           hundred triangles where the average ray has ten (measured: 124 iterations per wave with every lane testing its
           own); shared out, a wave needs its total / 64.  Item i of the wave's list belongs to the lane whose prefix range
           holds i; rounds of RT_MESHPRE_ROUND items go through LDS as (triangle, owner), the owner's ray comes by shuffle. */
        int incl = ntri;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (wlane >= d) incl += v; }
        const int total = __shfl(incl, 63, 64);
        int next = incl - ntri;                 /* the list index of this lane's next triangle */
        int q_cur = 0, left = 0;
        for (int base = 0; base < total; base += RT_MESHPRE_ROUND) {
            const int end = base + RT_MESHPRE_ROUND;
            while (next < incl && next < end) {
                if (left == 0) { const int e = MSTK(lq); lq++; q_cur = e & 0x07FFFFFF; left = (e >> 27) + 1; }
                buf_q[next - base] = q_cur; buf_o[next - base] = wlane;
                q_cur++; left--; next++;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int cnt = total - base < RT_MESHPRE_ROUND ? total - base : RT_MESHPRE_ROUND;
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                const int i = i0 + wlane;
                const bool act = i < cnt;
                const int o = act ? buf_o[i] : wlane, q = act ? buf_q[i] : 0;
                DRay r;
                r.org = d3(__shfl(ray.org.x, o, 64), __shfl(ray.org.y, o, 64), __shfl(ray.org.z, o, 64));
                r.dir = d3(__shfl(ray.dir.x, o, 64), __shfl(ray.dir.y, o, 64), __shfl(ray.dir.z, o, 64));
                MESH_WAVE_COUNT(5);
                if (act) {
                    double t;
                    MESH_COUNT(3, 1);
                    if (mesh_triangle(tris[q], r, t)) { MESH_COUNT(6, 1); mesh_cross_emit(P, wave_nh, b, o, t, q); }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (live) {
            const int nh = wave_nh[wlane];
            P.mx_n[slot] = (ok && nh <= RT_MESH_MAXHITS) ? nh : -1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#undef MSTK
}
template __global__ void k_mesh_cross<false>(const RtDevScene, RtPipe, int, int);
template __global__ void k_mesh_cross<true>(const RtDevScene, RtPipe, int, int);

template <bool LITE, int MODE, bool ALT = false>
__global__ void RT_LB_SHADOW(LITE, MODE) k_shadow(const RtDevScene Sv, RtPipe P, int nrays, int n_chains_lds)
{
    constexpr bool CH_LDS = MODE >= 1;
    RtDevScene SL = Sv;
    SceneLds lds = scene_lds_load(Sv, n_chains_lds, MODE == 2 ? &SL : 0);
    const RtDevScene &S = SL;
    if constexpr (!LITE) if (nrays == 0) { lds.mx_n = P.mx_n; lds.mx_t = P.mx_t; lds.mx_q = P.mx_q; lds.mxcap = P.mxcap; }       /* slow list: k_mesh_cross went first */
    constexpr bool MESHPRE = LITE && ALT, SCHED = !LITE && ALT;
    if constexpr (LITE && MESHPRE) { lds.mx_n = P.mx_n; lds.mx_t = P.mx_t; lds.mx_q = P.mx_q; lds.mxcap = P.mxcap; }
    const bool slow_list = !LITE && nrays == 0;     /* FULL: nrays == 0 = the slow list of the LITE launch before it */
    if (nrays < 0) nrays = P.counters[CTR_SNAP + 1];     /* the level's shadow rays, frozen by k_level_snap: no host round trip */
    const bool direct = !slow_list;
    const int n = direct ? nrays : P.counters[CTR_SLOW_S];
    int *const work = &P.counters[LITE ? CTR_WORK_S : CTR_WORK_SS];
    const int wlane = threadIdx.x & 63;
  int wnext = 0, wlimit = 0;      /* the wave's current claim: items [wnext, wlimit) */
  for (;;) {
    if (wnext >= wlimit) {
        int b = 0;
        if (wlane == 0) b = atomicAdd(work, RT_CLAIM);
        b = __shfl(b, 0, 64);
        if (b >= n) break;
        wnext = b; wlimit = b + RT_CLAIM < n ? b + RT_CLAIM : n;
    }
    const int wbase = wnext;
    wnext += 64;
    const bool live = wbase + wlane < n;
    int s;
    if (direct) s = wbase + wlane;
    else s = live ? P.slow_s[wbase + wlane] : 0;
    bool bail = false;
    if (live) {
        DRay ray;
        ray.org = d3_fetch(P.s_ray, P.scap, 0, s);
        ray.dir = d3_fetch(P.s_ray, P.scap, 3, s);
        const RtLightDev &l = S.lights[P.s_light[s]];
        double aux[6];
        int root = P.s_root[s];
        aux_load<LITE>(S, P, root, aux);
        double r2 = dot(ray.dir, ray.dir);
        Spec E = spec_div(spec_load(l.flux), 4.0 * RT_PI * r2);
#ifdef RT_SCHED_POOL     /* round 2's layout for the scheduled kernels: per-lane arenas in a pool in global memory (see intersect_scene) */
        ArenaT<SCHED> A;
        if constexpr (SCHED) A.bind(P.arena_pool, (size_t)(blockIdx.x * blockDim.x + threadIdx.x));
#else
        ArenaT<false> A;
#endif
        int start;
        int count = trace_scene<LITE, CH_LDS, false, MESHPRE, SCHED>(S, lds, ray, 1.0, A, start, 0, 0, -1,
                                                                  (LITE && MESHPRE) ? s : ((!LITE && !direct) ? (P.mx_all ? s : wbase + wlane) : -1));
        if (A.overflow) { atomicAdd(&P.counters[CTR_OVERFLOW], 1); atomicOr(&P.counters[CTR_OVERFLOW + 1], A.overflow); }
        bail = A.bail != 0;
        int i = 0, pm = 0;
        if (count > 0 && !(A.h[start].t > 0.0)) { pm = A.h[start].mat; i = 1; }
        double t = 0.0;
        Spec tr = spec_const(1.0f);
        bool blocked = false;
        int pm_luster_ov = -1, pm_has_color = 0;        /* Q13, as in k_visual */
        Spec pm_color = spec_const(0.0f);
        for (; !bail && i < count && A.h[start + i].t < 1.0 - RT_EPSILON; i++) {
            ShadeHit h;
            transform_normals<LITE>(S, ray, A.h[start + i], h, aux, bail);
            if (bail) break;
            const RtMaterialDev &mp = S.mats[pm], &mh = S.mats[h.mat];
            int mp_luster = mp.luster, mh_luster = mh.luster;
            const float *mp_colorp = mp.color;
            if constexpr (!LITE) {
                if (pm_luster_ov >= 0) mp_luster = pm_luster_ov;
                if (pm_has_color) mp_colorp = pm_color.v;
                if (h.luster_ov >= 0) mh_luster = h.luster_ov;
            }
            if (mp_luster != RT_MAT_TRANSPARENT || mh_luster != RT_MAT_TRANSPARENT || hit_roughness(S, h) != RT_SURF_POLISHED) { blocked = true; break; }
            if (mp.prop) {
                if constexpr (LITE) { bail = true; break; }
                else {
                    Spec ptr, pst;
                    propagate_in(S, mp.prop, ray_at(ray, t), ray_at(ray, h.t), mp_colorp, mp.refr, ptr, pst);
                    tr = tr * ptr;
                }
            }
            if (mp.refr != mh.refr) {
                D3 vE, cvN;
                double fNE, F = 0.0, fNT = 0.0;
                consistent_normal(ray, h, vE, cvN, fNE);
                fresnel_real(fNE, mh.refr, mp.refr, F, fNT);
                if (F == 1.0) { blocked = true; break; }
                tr = tr * (1.0 - F);
            }
            t = h.t;
            pm = h.mat;
            if constexpr (!LITE) { pm_luster_ov = h.luster_ov; pm_has_color = h.has_color; pm_color = h.color; }
        }
        if (!bail) {
            Spec out;
            if (blocked) out = spec_const(0.0f);
            else {
                if (S.mats[pm].prop && (pm_luster_ov >= 0 ? pm_luster_ov : S.mats[pm].luster) == RT_MAT_TRANSPARENT) {
                    if constexpr (LITE) bail = true;
                    else {
                        Spec ptr, pst;
                        propagate_in(S, S.mats[pm].prop, ray_at(ray, t), ray_at(ray, 1.0), pm_has_color ? pm_color.v : S.mats[pm].color, S.mats[pm].refr, ptr, pst);
                        tr = tr * ptr;
                    }
                }
                out = E * tr;
            }
            if (!bail) spec_store(P.s_E, P.scap, s, out);
        }
    }
    if constexpr (LITE) {
        int slot = wave_alloc(bail ? 1 : 0, &P.counters[CTR_SLOW_S]);
        if (bail) P.slow_s[slot] = s;
    }
  }
}

RT_DEV void frame_deliver(const RtDevScene &S, const RtPipe &P, int f, int flags, const Spec &R, float *out_rgb, float *out_spec);

/* ------------------------------------------------------------------ k_shade: light loop + child rays (Shade.cpp:477-501)
 * Frames [f0, f1) of one level, one per thread; f1 < 0: the level's end is on the device (CTR_SNAP, written by k_level_snap
 * behind the tracing kernels) and the host has launched for an upper bound -- workgroups beyond the end leave at once.
 * (A grid-stride loop over the frames made the compiler hoist per-frame address arithmetic out of it and spill 308 B per
 * lane at three waves per SIMD: 2.3 x slower per frame.) */
RT_DEV void shade_frames(const RtDevScene &S, const RtPipe &P, const int fb, const int f1, float *out_rgb, float *out_spec)
{
  {
    const int f = fb + threadIdx.x;
    const bool live = f < f1;
    int nchild = 0, flags = 0;
    bool want_refl = false, want_refr = false, have_R = false;
    Spec Rfin;
    D3 Ppt = d3(0, 0, 0), dir_refl = d3(0, 0, 0), dir_refr = d3(0, 0, 0);
    int depth = 0;
    if (live) {
        flags = P.f_flags[f];
        depth = FR_DEPTH(flags);
        if (flags & FR_BOUNDARY) {
            int root = P.f_root[f];
            double uL = 0.0, vL_ = 0.0;
            if (S.aux_light) { uL = P.aux[(size_t)2 * P.ncap + root]; vL_ = P.aux[(size_t)3 * P.ncap + root]; }
            Ppt = d3_fetch(P.f_geo, P.fcap, GEO_P, f);
            D3 cvN = d3_fetch(P.f_geo, P.fcap, GEO_N, f), vE = d3_fetch(P.f_geo, P.fcap, GEO_E, f), hn = d3_fetch(P.f_geo, P.fcap, GEO_HN, f);
            double fNE = P.f_geo[(size_t)GEO_FNE * P.fcap + f], F = P.f_geo[(size_t)GEO_F * P.fcap + f];
            double fNT = P.f_geo[(size_t)GEO_FNT * P.fcap + f], eta = P.f_geo[(size_t)GEO_ETA * P.fcap + f];
            int rough = P.f_ids[(size_t)IDS_ROUGH * P.fcap + f], luster = P.f_ids[(size_t)IDS_LUSTER * P.fcap + f];
            const int mat = luster >> 8;
            luster &= 0xFF;
            const int base = P.f_ids[(size_t)IDS_SHBASE * P.fcap + f];
            const unsigned mask = (unsigned)P.f_ids[(size_t)IDS_SHMASK * P.fcap + f];
            Spec color = (flags & FR_COLOR_MAT) ? spec_load(S.mats[mat].color) : spec_fetch(P.f_color, P.fcap, f), R;
            if (flags & FR_R_ZERO) R = spec_const(0.0f);
            else if (flags & FR_R_AMBIENT) { Spec E = spec_load(S.ambient) * RT_PI; R = spec_const(0.0f); R = R + color * E; }     /* as k_visual would have */
            else R = spec_fetch(P.f_R, P.fcap, f);
            /* a slot beyond the queue: the pass overflowed, the host discards and splits it */
            R = light_loop(S, Ppt, cvN, vE, hn, F, rough, luster, color, R, uL, vL_, mask,
                           [&](int k, Spec &E) { if (base + k >= P.scap) return false; E = spec_fetch(P.s_E, P.scap, base + k); return true; });
            Rfin = R; have_R = true;
            if (rough == RT_SURF_POLISHED) {
                int perturbed = (flags & FR_PERTURBED) ? 1 : 0;
                if (F != 0.0 && depth + 1 < 6) {
                    spec_store(P.f_w1, P.fcap, f, ray_reflectance(luster, color, F));
                    flags |= FR_C1;
                    if (depth + 1 < RT_MAX_DEPTH) { want_refl = true; dir_refl = reflect_dir(cvN, vE, hn, perturbed, fNE); nchild++; }
                    else spec_store(P.f_c1, P.fcap, f, spec_const(0.0f));
                }
                if (luster == RT_MAT_TRANSPARENT && F < 1.0) {
                    P.f_geo[(size_t)GEO_W2 * P.fcap + f] = 1.0 - F;
                    flags |= FR_C2;
                    if (depth + 1 < RT_MAX_DEPTH) { want_refr = true; dir_refr = refract_dir(cvN, vE, hn, perturbed, eta, fNE, fNT); nchild++; }
                    else spec_store(P.f_c2, P.fcap, f, spec_const(0.0f));
                }
            }
        }
        /* A frame without reflection / refraction terms is finished here: R * tr + fin, then on to its parent or its
           sample (what k_resolve would do, without another pass over the frame); the others keep R for k_resolve. */
        if (!(flags & (FR_C1 | FR_C2))) {
            Spec R = have_R ? Rfin : ((flags & FR_R_ZERO) ? spec_const(0.0f) : spec_fetch(P.f_R, P.fcap, f));
            if ((flags & FR_BOUNDARY) && !(flags & FR_TR_ONE)) R = R * spec_fetch(P.f_tr, P.fcap, f);
            if (flags & FR_FIN) R = R + spec_fetch(P.f_fin, P.fcap, f);
            frame_deliver(S, P, f, flags, R, out_rgb, out_spec);
            flags |= FR_DONE;
            P.f_flags[f] = flags;
        } else {
            if (have_R) spec_store(P.f_R, P.fcap, f, Rfin);
            P.f_flags[f] = flags;
        }
    }
    int base = wave_alloc(nchild, &P.counters[CTR_FRAMES]);
    if (!live || nchild == 0) return;
    int c = base;
    for (int which = 0; which < 2; which++) {
        if (which == 0 ? !want_refl : !want_refr) continue;
        int nf = c++;
        if (nf >= P.fcap) continue;             /* overflow reported by the host */
        d3_store(P.f_ray, P.fcap, 0, nf, Ppt);
        d3_store(P.f_ray, P.fcap, 3, nf, which == 0 ? dir_refl : dir_refr);
        P.f_parent[nf] = f;
        P.f_root[nf] = P.f_root[f];
        P.f_flags[nf] = FR_MAKE_DEPTH(depth + 1) | (which ? FR_SLOT2 : 0);
    }
  }
}
__global__ void __launch_bounds__(256, 3) k_shade(const RtDevScene Sv, RtPipe P, int f0, int f1, float *out_rgb, float *out_spec)
{
    if (f1 < 0) f1 = P.counters[CTR_SNAP];
    const int fb = f0 + blockIdx.x * 256;
    if (fb >= f1) return;
    shade_frames(Sv, P, fb, f1, out_rgb, out_spec);
}
/* The same over a level whose end is on the device and is EXPECTED to be at f0 (the call before on this scene left level 0 empty:
   S0, whose primary rays all end in the generator or in k_fast): a few workgroups that loop, instead of one workgroup per 256
   frames of the upper bound -- 518 k workgroups that read the counter and leave cost 0.11 ms per pass, 4 % of a band of an
   eight-way split.  If the expectation is wrong the loop does the work, correctly and some 2 x slower per frame (see above). */
__global__ void __launch_bounds__(256, 3) k_shade_loop(const RtDevScene Sv, RtPipe P, int f0, float *out_rgb, float *out_spec)
{
    const int f1 = P.counters[CTR_SNAP];
    for (int fb = f0 + blockIdx.x * 256; fb < f1; fb += gridDim.x * 256) shade_frames(Sv, P, fb, f1, out_rgb, out_spec);
}

/* ------------------------------------------------------------------ k_resolve */
/* a root's finished radiance to its sample: Spectrum::sRGB / vRGB (RadiometryColor.h:62-66, RadiometryColor.cpp:971-983) */
RT_DEV void sample_deliver(const RtDevScene &S, const int root, const Spec &R, float *out_rgb, float *out_spec, const unsigned short *slot)
{
    if (out_spec) for (int k = 0; k < RT_NSPEC; k++) out_spec[(size_t)root * RT_NSPEC + k] = R.v[k];
    if (out_rgb) {
        float r = 0.0f, g = 0.0f, b = 0.0f;
        if (S.venera) {
            for (int i = 0; i < RT_NSPEC; i++) { r += R.v[i] * S.chan[0][i]; g += R.v[i] * S.chan[1][i]; b += R.v[i] * S.chan[2][i]; }
        } else {
            for (int i = 0; i < RT_NSPEC; i++) {
                double w = (double)R.v[i] * (5.0 / 10.0);
                r = r + (float)((double)S.spec2rgb[i][0] * w);
                g = g + (float)((double)S.spec2rgb[i][1] * w);
                b = b + (float)((double)S.spec2rgb[i][2] * w);
            }
        }
        const size_t rs = (size_t)rgb_slot_of(slot, root);
        out_rgb[rs * 3] = r; out_rgb[rs * 3 + 1] = g; out_rgb[rs * 3 + 2] = b;
    }
}
/* hand a frame's finished radiance to its parent (Shade.cpp:491-501) or, for a root, to the sample */
RT_DEV void frame_deliver(const RtDevScene &S, const RtPipe &P, int f, int flags, const Spec &R, float *out_rgb, float *out_spec)
{
    int parent = P.f_parent[f];
    if (parent >= 0) {
        spec_store((flags & FR_SLOT2) ? P.f_c2 : P.f_c1, P.fcap, parent, R);
        return;
    }
    sample_deliver(S, P.f_root[f], R, out_rgb, out_spec, P.rgb_slot);
}

__global__ void __launch_bounds__(256) k_resolve(const RtDevScene Sv, RtPipe P, int f0, int f1, float *out_rgb, float *out_spec)
{
    int f = f0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= f1) return;
    const RtDevScene &S = Sv;
    int flags = P.f_flags[f];
    if (flags & FR_DONE) return;
    Spec R = spec_fetch(P.f_R, P.fcap, f);
    if (flags & FR_BOUNDARY) {
        if (flags & FR_C1) R = R + spec_fetch(P.f_c1, P.fcap, f) * spec_fetch(P.f_w1, P.fcap, f);
        if (flags & FR_C2) R = R + spec_fetch(P.f_c2, P.fcap, f) * P.f_geo[(size_t)GEO_W2 * P.fcap + f];
        if (!(flags & FR_TR_ONE)) R = R * spec_fetch(P.f_tr, P.fcap, f);
    }
    if (flags & FR_FIN) R = R + spec_fetch(P.f_fin, P.fcap, f);
    frame_deliver(S, P, f, flags, R, out_rgb, out_spec);
}

/* ------------------------------------------------------------------ probe: raw hit lists */
__global__ void __launch_bounds__(256) k_probe_intersect(const RtDevScene Sv, int n, const double *org, const double *dir, const double *aux6,
                                                         int max_hits, int transform, int *out_count, double *out_t, double *out_n,
                                                         int *out_leaf, int *out_mat, int *out_surf, int *overflow)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    const RtDevScene &S = Sv;
    const SceneLds lds = scene_lds_load(S, 0);
    if (k >= n) return;
    DRay ray;
    ray.org = d3(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
    ray.dir = d3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
    double aux[6];
    for (int i = 0; i < 6; i++) aux[i] = aux6 ? aux6[6 * k + i] : S.aux_init[i];
    Arena A;
    int start;
    int count = intersect_scene<false>(S, lds, ray, RT_INFINITY, A, start);
    if (A.overflow) atomicOr(overflow, A.overflow);
    out_count[k] = count;
    for (int i = 0; i < max_hits; i++) {
        size_t o = (size_t)k * max_hits + i;
        if (i < count) {
            const HitRec &hr = A.h[start + i];
            D3 nn = hr.n;
            if (transform) {
                ShadeHit h;
                { bool nb = false; transform_normals<false>(S, ray, hr, h, aux, nb); }
                nn = h.perturbed ? h.np : h.n;
            }
            out_t[o] = hr.t;
            out_n[3 * o] = nn.x; out_n[3 * o + 1] = nn.y; out_n[3 * o + 2] = nn.z;
            out_leaf[o] = (hr.prim & 0x8000) ? S.clips[hr.prim & 0x7FFF].id : S.prims[hr.prim].id;
            out_mat[o] = S.mats[hr.mat].id;
            out_surf[o] = S.surfs[hr.surf].id;
        } else {
            out_t[o] = 0.0; out_n[3 * o] = out_n[3 * o + 1] = out_n[3 * o + 2] = 0.0;
            out_leaf[o] = out_mat[o] = out_surf[o] = -9;
        }
    }
}

/* ------------------------------------------------------------------ probe: ideal-culling event counts (bench.py roofline)
 * Counts, for each ray, the events of the IDEAL-culling algorithm of SURVEY.md section 8d: a primitive is
 * tested only if the ray touches that primitive's own box; the box tests are those of a balanced hierarchy
 * over the leaves (2 per level on the way down plus 2 per leaf reached). */
__global__ void __launch_bounds__(256) k_probe_work(const RtDevScene Sv, int n, const double *org, const double *dir, unsigned long long *out /*7*/)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    const RtDevScene &S = Sv;
    const SceneLds lds = scene_lds_load(S, 0);
    if (k >= n) return;
    DRay ray;
    ray.org = d3(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
    ray.dir = d3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
    unsigned leaf = 0, nleaves = 0;
    const CullRay cr = cull_ray(ray, RT_INFINITY);
    for (int pc = 0; pc < S.n_prog; pc++) {
        int op = S.prog[pc].op;
        if (op == OP_END) break;            /* behind it: the single-chain shortcuts of the root region (RtDevScene::root_alt_pc) */
        bool hit = true;
        if (op == OP_CHAIN) {
            int b = S.chains[S.prog[pc].a].box;
            if (b >= 0) hit = cull_box_hit(S.boxes[b].lo, S.boxes[b].hi, cr);
        } else if ((op >= OP_QUADRIC && op <= OP_TORUS) || op == OP_MESH) {
            if (pc > 0 && S.prog[pc - 1].op == OP_BOUND) hit = cull_box_hit(S.boxes[S.prog[pc - 1].a].lo, S.boxes[S.prog[pc - 1].a].hi, cr);
        } else continue;
        nleaves++;
        if (hit) leaf++;
    }
    unsigned levels = 0;
    while ((1u << levels) < nleaves) levels++;
    /* only rays that reach k_visual count: the others ended in k_generate_cull */
    if (!scene_may_hit(S, lds, ray, RT_INFINITY)) return;
    Arena A;
    int start;
    int count = intersect_scene<false>(S, lds, ray, RT_INFINITY, A, start);
    atomicAdd(&out[0], (unsigned long long)leaf);
    atomicAdd(&out[1], (unsigned long long)leaf);
    atomicAdd(&out[2], (unsigned long long)(2 * levels + 2 * leaf));
    atomicAdd(&out[3], (unsigned long long)(2 * leaf));
    atomicAdd(&out[4], (unsigned long long)(count > 0 ? 1 : 0));
    atomicAdd(&out[5], (unsigned long long)(count > 0 ? S.n_lights : 0));
    atomicAdd(&out[6], 1ull);
}

/* ------------------------------------------------------------------ probes: leaf functions the reference tests on their own
 * TestBandNoise (Test.cpp:1635-1653) / TestVoronoi (Test.cpp:3798-3836): BandNoise(p, 3), FractalNoise(p, 3, 5, 1.0f, 1.9f, 0), Voronoi(p) */
__global__ void __launch_bounds__(256) k_probe_noise(int n, const float *p3, float omega19, float *out_band, float *out_fractal5, float *out_voronoi)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float p[3] = { p3[3 * k], p3[3 * k + 1], p3[3 * k + 2] };
    float g[3];
    out_band[k] = band_noise3(p);
    out_fractal5[k] = fractal_noise3(p, 5, 0, omega19);
    out_voronoi[k] = gradient_voronoi(g, p, 1);
}
/* TestFresnel (Test.cpp:1467-1515): FresnelLaw_Real for (n1, n2) and Fresnel_Spectral of a metal's n / k spectra (nk: 10 + 10 floats) seen from n1 */
__global__ void __launch_bounds__(256) k_probe_fresnel(int n, const double *cos_ne, double n1, double n2, const float *nk, double *out_F, double *out_fNT, float *out_spec10)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double F = 0.0, fNT = 0.0;
    fresnel_real(cos_ne[k], n1, n2, F, fNT);
    out_F[k] = F; out_fNT[k] = fNT;
    if (out_spec10) {
        const Spec s = fresnel_spectral(cos_ne[k], n1, nk, nk + RT_NSPEC);
        for (int i = 0; i < RT_NSPEC; i++) out_spec10[(size_t)k * RT_NSPEC + i] = s.v[i];
    }
}

/* ------------------------------------------------------------------ reconstruction
 * FastBlueSamplingTile splats every sample bilinearly into a supersampled image (Image::FastSplatRGB,
 * Image.cpp:1309-1334), then resamples with a separable Kaiser-windowed sinc (ImageResample,
 * Sampling.cpp:186-234) and divides by the equally filtered weights (Image::Normalize, Image.cpp:1238-1260).
 * A scatter with float atomics would make the float sums order dependent.  Instead every supersample
 * pixel GATHERS its (on average four) contributions in the reference's loop order -- tile x outer, tile y
 * inner, sample index, corner -- from a per-tile-pixel contributor table built once from the blue-noise
 * tile, which reproduces the reference's float accumulation bit for bit and needs no atomics. */
RT_DEV void splat_pixel(const RtDevScene &S, const RtSampleGrid &g, const RtSplatTable &T, const float *sample_rgb, int px, int py,
                        float &r, float &gg, float &b, float &w)
{
    r = 0.0f; gg = 0.0f; b = 0.0f; w = 0.0f;
    int tcx = px / RT_BLUE_NSQRT, tcy = py / RT_BLUE_NSQRT;
    for (int tx = tcx - 1; tx <= tcx + 1; tx++) {
        if (tx < 0 || tx >= g.tiles_x) continue;
        int lx = px - tx * RT_BLUE_NSQRT;
        if (lx < -1 || lx > RT_BLUE_NSQRT) continue;
        for (int ty = tcy - 1; ty <= tcy + 1; ty++) {
            if (ty < g.tile_y0 || ty >= g.tile_y0 + g.tiles_y_local) continue;
            int ly = py - ty * RT_BLUE_NSQRT;
            if (ly < -1 || ly > RT_BLUE_NSQRT) continue;
            int cell = (ly + 1) * (RT_BLUE_NSQRT + 2) + (lx + 1);
            int e0 = T.offsets[cell], e1 = T.offsets[cell + 1];
            const int tile = tx * g.tiles_y_local + (ty - g.tile_y0);
            int64_t sbase = (int64_t)tile * RT_BLUE_N;
            /* a masked tile's samples are black: they add weight only (adding +0 to a sum that is never -0 changes nothing) */
            const bool black = g.tile_mask && g.mask_skips_rgb && g.tile_mask[tile] == 0;
            for (int e = e0; e < e1; e++) {
                int i = T.entries[e];
                /* v = g_rgvBlueNoise[i]*BLUE_NSQRT + vTile (Sampling.cpp:501), then FastSplatRGB's own arithmetic */
                double x = S.blue[2 * i] * (double)RT_BLUE_NSQRT + (double)tx * (double)RT_BLUE_NSQRT;
                double y = S.blue[2 * i + 1] * (double)RT_BLUE_NSQRT + (double)ty * (double)RT_BLUE_NSQRT;
                x -= 0.5; y -= 0.5;
                double fx = floor(x), fy = floor(y);
                int i1 = (int)fx, j1 = (int)fy;
                double dx = x - fx, dy = y - fy, wx, wy;
                if (px == i1) wx = 1.0 - dx; else if (px == i1 + 1) wx = dx; else continue;
                if (py == j1) wy = 1.0 - dy; else if (py == j1 + 1) wy = dy; else continue;
                w += (float)(wx * wy);
                if (black) continue;
                const float *c = sample_rgb + 3 * (sbase + i);
                r += (float)((double)(float)((double)c[0] * wx) * wy);
                gg += (float)((double)(float)((double)c[1] * wx) * wy);
                b += (float)((double)(float)((double)c[2] * wx) * wy);
            }
        }
    }
}
__global__ void __launch_bounds__(256) k_splat_gather(const RtDevScene Sv, RtSampleGrid g, RtSplatTable T, const float *sample_rgb,
                                                      int y0, int y1, float *im2 /*[3][rows][nW]*/, float *wt2 /*[rows][nW]*/)
{
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    int py = y0 + blockIdx.y;
    if (px >= g.nW || py >= y1) return;
    float r = 0.0f, gg = 0.0f, b = 0.0f, w = 0.0f;
    bool needed = true;
    if (g.tile_need) {
        const int tx = px / RT_BLUE_NSQRT, ty = py / RT_BLUE_NSQRT - g.tile_y0;
        if (tx < g.tiles_x && ty >= 0 && ty < g.tiles_y_local) needed = g.tile_need[tx * g.tiles_y_local + ty] != 0;
    }
    if (needed) splat_pixel(Sv, g, T, sample_rgb, px, py, r, gg, b, w);
    else if (g.need_lazy) return;           /* never read: k_resample_rows takes this pixel as the zero it would hold */
    size_t rows = (size_t)(y1 - y0), o = (size_t)(py - y0) * g.nW + px;
    im2[o] = r; im2[rows * g.nW + o] = gg; im2[2 * rows * g.nW + o] = b;
    wt2[o] = w;
}

/* ------------------------------------------------------------------ BlueNoiseSamplingTile reconstruction
 * Image::SplatRGB (Image.cpp:1203-1236) spreads every sample over the <= 9 x 9 output pixels under a separable
 * Kaiser-windowed sinc (float Kaiser4Filter, Image.cpp:890-918), into an image whose indices wrap by mirroring
 * (NewImage's default, Image.h:157; ML_Wrap, Image.cpp:59-83), then Normalize (Image.cpp:1238-1260).  As with the
 * bilinear splat, every output pixel GATHERS its contributions in the reference's order (tile x outer, tile y inner,
 * sample index, x index, y index) so the float sums come out the same without atomics: a wave owns 8 x 8 pixels,
 * walks the tiles within reach and tests all 1600 samples of each (their positions are wave-uniform). */
RT_DEV float kaiser4_filter_f(float x)
{
    float x1, x2, xTerm, fN, fSinc, fKaiser;
    x1 = x * 1.0f / 4.0f;
    x2 = x1 * x1;
    if (x2 > 1.0) return 0.0;
    x2 = 0.25f * 16.0f * (1.0f - x2);
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
    else fSinc = (float)sin((double)x1) / x1;       /* sinf: the double sine rounded to float */
    return fSinc * fKaiser * (float)(1.0 / 11.3019219521363);
}
RT_DEV int mirror_index(int i, int n)                   /* ML_Wrap, WRAP_MIRROR */
{
    i = i % (2 * n);
    if (i < 0) i += 2 * n;
    if (i >= n) i = 2 * n - i - 1;
    return i;
}
/* indices q of [lo, hi] that land on pixel p of an axis of n pixels, ascending; most samples: none or one */
RT_DEV int axis_matches(int lo, int hi, int p, int n, int *q /*[9]*/)
{
    if (lo >= 0 && hi < n) { if (p < lo || p > hi) return 0; q[0] = p; return 1; }
    int c = 0;
    for (int k = lo; k <= hi && c < 9; k++) if (mirror_index(k, n) == p) q[c++] = k;
    return c;
}
__global__ void __launch_bounds__(256) k_kaiser_gather(const RtDevScene Sv, RtSampleGrid g, const float *sample_rgb, int row0, int row1, float *planes)
{
    __shared__ double s_b[2 * RT_BLUE_N];
    const RtDevScene &S = Sv;
    for (int k = threadIdx.x; k < 2 * RT_BLUE_N; k += blockDim.x) s_b[k] = S.blue[k] * g.bn_sqrt;      /* g_rgvBlueNoise[i]*fSqrt */
    __syncthreads();
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int px = blockIdx.x * 16 + (wv & 1) * 8 + (ln & 7), py = row0 + blockIdx.y * 16 + (wv >> 1) * 8 + (ln >> 3);
    const int W = g.width, H = g.height;
    if (px >= W || py >= row1) return;
    /* tiles whose samples can reach this pixel, directly or through the mirrored border (generous: every sample is tested) */
    const double xt = g.bn_tx[g.tiles_x - 1] + g.bn_sqrt, yt = g.bn_ty[g.tiles_y - 1] + g.bn_sqrt;
    double xlo = (double)px - 5.0, xhi = (double)px + 6.0, ylo = (double)py - 5.0, yhi = (double)py + 6.0;
    if (2.0 * W - 1.0 - px - 5.0 < xt) xhi = 2.0 * W - px + 5.0;
    if (2.0 * H - 1.0 - py - 5.0 < yt) yhi = 2.0 * H - py + 5.0;
    int tx0 = (int)floor(xlo / g.bn_sqrt) - 2, tx1 = (int)floor(xhi / g.bn_sqrt) + 1, ty0 = (int)floor(ylo / g.bn_sqrt) - 2, ty1 = (int)floor(yhi / g.bn_sqrt) + 1;
    if (W < 16) { tx0 = 0; tx1 = g.tiles_x - 1; }
    if (H < 16) { ty0 = 0; ty1 = g.tiles_y - 1; }
    if (tx0 < 0) tx0 = 0;
    if (tx1 > g.tiles_x - 1) tx1 = g.tiles_x - 1;
    if (ty0 < g.tile_y0) ty0 = g.tile_y0;
    if (ty1 > g.tile_y0 + g.tiles_y_local - 1) ty1 = g.tile_y0 + g.tiles_y_local - 1;
    float r = 0.0f, gg = 0.0f, b = 0.0f, w = 0.0f;
    for (int tx = tx0; tx <= tx1; tx++) {
        const double vtx = g.bn_tx[tx];
        for (int ty = ty0; ty <= ty1; ty++) {
            const double vty = g.bn_ty[ty];
            const float *c0 = sample_rgb + 3 * (((int64_t)tx * g.tiles_y_local + (ty - g.tile_y0)) * RT_BLUE_N);
            for (int i = 0; i < RT_BLUE_N; i++) {
                const double vx = s_b[2 * i] + vtx, vy = s_b[2 * i + 1] + vty;
                const int iLow = (int)floor((vx - 4.0) + 0.5), iHigh = (int)floor((vx + 4.0) - 0.5);
                int qx[9], qy[9];
                const int nx = axis_matches(iLow, iHigh, px, W, qx);
                if (nx == 0) continue;
                const int jLow = (int)floor((vy - 4.0) + 0.5), jHigh = (int)floor((vy + 4.0) - 0.5);
                const int ny = axis_matches(jLow, jHigh, py, H, qy);
                if (ny == 0) continue;
                const float cr = c0[3 * i], cg = c0[3 * i + 1], cb = c0[3 * i + 2];
                for (int a = 0; a < nx; a++) {
                    const double fWeightX = kaiser4_filter_f((float)((((double)qx[a] + 0.5) - vx) * 1.0));
                    for (int e = 0; e < ny; e++) {
                        const double fWeight = fWeightX * kaiser4_filter_f((float)((((double)qy[e] + 0.5) - vy) * 1.0));
                        r += (float)((double)cr * fWeight);
                        gg += (float)((double)cg * fWeight);
                        b += (float)((double)cb * fWeight);
                        w += (float)fWeight;
                    }
                }
            }
        }
    }
    if (w) { r /= w; gg /= w; b /= w; }                 /* Image::Normalize */
    const size_t o = (size_t)py * W + px, plane = (size_t)W * H;
    planes[o] = r; planes[plane + o] = gg; planes[2 * plane + o] = b;
}

/* ------------------------------------------------------------------ AdaptiveBlueSamplingTile, Sampling.cpp:619-679
 * After the base pass (a FastBlue pass) the reference computes the normalised luminance of the supersampled image,
 * its Roberts-cross gradient relative to the local sum, and refines every 20 x 20 block whose (24 x 24, mirrored)
 * neighbourhood holds a gradient above 0.30 with 1600 more samples splat into the same image.  Rows are local:
 * arrays hold supersample rows [y0, y0 + rows). */
__global__ void __launch_bounds__(256) k_adapt_luminance(const float *im2, const float *wt2, int nW, int rows, float *lum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, jl = blockIdx.y;
    if (i >= nW) return;
    const size_t o = (size_t)jl * nW + i, plane = (size_t)rows * nW;
    float l = 0.212649390101432f * im2[o] + 0.715169131755828f * im2[plane + o] + 0.0721815153956413f * im2[2 * plane + o];    /* DisplayRGB::Luminance, Image.h:75 */
    const float w = wt2[o];
    if (w) l /= w;                                      /* imDetect.Normalize(&im2Weight) */
    lum[o] = l;
}
/* the reference overwrites imDetect in place, row by row; every value it reads (right, below, diagonal; the mirrored
   index n maps to n - 1) is still a luminance at that moment, so out of place with clamped reads is the same */
__global__ void __launch_bounds__(256) k_adapt_roberts(const float *lum, int nW, int nH, int y0, int rows, float *grad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, jl = blockIdx.y;
    if (i >= nW) return;
    const int j = y0 + jl;
    int j1 = j + 1 < nH ? j + 1 : nH - 1;
    const int i1 = i + 1 < nW ? i + 1 : nW - 1;
    float out = 0.0f;
    if (j1 - y0 < rows) {                               /* the last local row of a band has no row below it and is never needed */
        const double f00 = lum[(size_t)jl * nW + i], f01 = lum[(size_t)(j1 - y0) * nW + i], f10 = lum[(size_t)jl * nW + i1], f11 = lum[(size_t)(j1 - y0) * nW + i1];
        const double g1 = f00 - f11, g2 = f01 - f10, sum = f00 + f01 + f10 + f11;
        double gv;
        if (sum > 0.0) gv = sqrt(g1 * g1 + g2 * g2) / sum * 1.0; else gv = 0.0;
        out = (float)gv;
    }
    grad[(size_t)jl * nW + i] = out;
}
/* one wave per block: the maximum over its neighbourhood, flag = maxg > THRESHOLD1 */
__global__ void __launch_bounds__(64) k_adapt_blocks(const float *grad, int nW, int nH, int y0, int rows, int bx, int bj0, int *flags)
{
    const int bi = blockIdx.x, bj = bj0 + blockIdx.y, span = RT_ADAPT_BLOCK + 2 * RT_ADAPT_HYST;
    float m = 0.0f;
    for (int e = threadIdx.x; e < span * span; e += 64) {
        const int is = mirror_index(bi * RT_ADAPT_BLOCK - RT_ADAPT_HYST + e % span, nW), js = mirror_index(bj * RT_ADAPT_BLOCK - RT_ADAPT_HYST + e / span, nH);
        const int jl = js - y0;
        if (jl < 0 || jl >= rows) continue;             /* cannot happen: the host sizes the row range for it */
        const float v = grad[(size_t)jl * nW + is];
        if (v > m) m = v;
    }
    for (int o = 32; o > 0; o >>= 1) { const float v = __shfl_xor(m, o, 64); if (v > m) m = v; }
    if (threadIdx.x == 0) flags[(size_t)bj * bx + bi] = ((double)m > 0.30) ? 1 : 0;
}
/* refinement splats: FastSplatRGB of the refined blocks' samples, added to the base pass's sums in the reference's
   order (block row outer, block column inner, sample index) */
__global__ void __launch_bounds__(256) k_splat_gather_refine(const RtDevScene Sv, RtSampleGrid g, RtSplatTable T, const float *sample_rgb,
                                                             int y0, int y1, float *im2 /*[3][rows][nW]*/, float *wt2 /*[rows][nW]*/)
{
    const RtDevScene &S = Sv;
    const int px = blockIdx.x * blockDim.x + threadIdx.x, py = y0 + blockIdx.y, B = RT_ADAPT_BLOCK;
    if (px >= g.nW || py >= y1) return;
    const size_t rows = (size_t)(y1 - y0), o = (size_t)(py - y0) * g.nW + px;
    float r = im2[o], gg = im2[rows * g.nW + o], b = im2[2 * rows * g.nW + o], w = wt2[o];
    const int bcx = px / B, bcy = py / B;
    bool any = false;
    for (int bj = bcy - 1; bj <= bcy + 1; bj++) {
        if (bj < g.ad_bj0 || bj > g.ad_bj1) continue;
        const int ly = py - bj * B;
        if (ly < -1 || ly > B) continue;
        for (int bi = bcx - 1; bi <= bcx + 1; bi++) {
            if (bi < 0 || bi >= g.ad_bx) continue;
            const int lx = px - bi * B;
            if (lx < -1 || lx > B) continue;
            const int ord = g.ad_ord[bj * g.ad_bx + bi];
            if (ord < 0) continue;
            const int cell = (ly + 1) * (B + 2) + (lx + 1);
            const int e0 = T.offsets[cell], e1 = T.offsets[cell + 1];
            const int64_t sbase = (int64_t)ord * RT_BLUE_N;
            for (int e = e0; e < e1; e++) {
                const int k = T.entries[e];
                double x = S.blue[2 * k] * (double)RT_BLUE_NSQRT / 2.0 + (double)(bi * B);
                double y = S.blue[2 * k + 1] * (double)RT_BLUE_NSQRT / 2.0 + (double)(bj * B);
                x -= 0.5; y -= 0.5;
                const double fx = floor(x), fy = floor(y);
                const int i1 = (int)fx, j1 = (int)fy;
                const double dx = x - fx, dy = y - fy;
                double wx, wy;
                if (px == i1) wx = 1.0 - dx; else if (px == i1 + 1) wx = dx; else continue;
                if (py == j1) wy = 1.0 - dy; else if (py == j1 + 1) wy = dy; else continue;
                const float *c = sample_rgb + 3 * (sbase + k);
                r += (float)((double)(float)((double)c[0] * wx) * wy);
                gg += (float)((double)(float)((double)c[1] * wx) * wy);
                b += (float)((double)(float)((double)c[2] * wx) * wy);
                w += (float)(wx * wy);
                any = true;
            }
        }
    }
    if (any) { im2[o] = r; im2[rows * g.nW + o] = gg; im2[2 * rows * g.nW + o] = b; wt2[o] = w; }
}
/* fSamplesPerPixel == 1.0 (Sampling.cpp:687-692): the supersampled image IS the output; Normalize by its weights */
__global__ void __launch_bounds__(256) k_adapt_direct(const float *im2, const float *wt2, int width, int height, int y0, int rows, int row0, int row1, float *planes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = row0 + blockIdx.y;
    if (i >= width || j >= row1) return;
    const size_t o = (size_t)(j - y0) * width + i, plane = (size_t)rows * width;
    const float w = wt2[o];
    for (int k = 0; k < 3; k++) {
        float v = im2[k * plane + o];
        if (w) v /= w;
        planes[((size_t)k * height + j) * width + i] = v;
    }
}

/* Resample (Sampling.cpp:131-184), WRAP_CLAMP.  src: [planes][src_rows][src_n] -> dst laid out [planes][src_rows][dst_n] */
/* One workgroup = 256 consecutive destinations of one source row.  The source span they read (256 * ratio + taps
   floats) is staged in LDS once, padded by one word per 32 so that the lanes' strided reads (stride = the
   decimation ratio) spread over the banks; with an integer ratio every destination has the same taps and those
   are staged too.  The sums run in the reference's order (Sampling.cpp:158-176), in double. */
#define RT_RS_SPAN  4608        /* floats of source per workgroup that fit the staging buffer */
#define RT_RS_TAPS  520
#define RT_RS_IDX(j) ((j) + ((j) >> 5))
/* need (optional, FastBlue): RtSampleGrid::tile_need with its geometry.  A destination all of whose source pixels lie in
   tiles that cannot matter reads zeros only: the zero it would compute (+0 / wsum) is written without the tap loop. */
struct RtNeedMap { const unsigned char *need; int tiles_x, tiles_y_local, tile_row0 /* tile_y0 * 40 - src_y0: local source row where tile row 0 begins */;
                   int lazy /* pixels of tiles with need == 0 were not written by the gather: they are zeros by definition, never loaded */; };
/* is supersample pixel x of local source row `row` one the splat gather left unwritten (k_splat_gather's own rule)? */
RT_DEV bool need_unwritten(const RtNeedMap &nd, int x, int row)
{
    if (!nd.lazy || row < nd.tile_row0) return false;
    const int tx = x / RT_BLUE_NSQRT, ty = (row - nd.tile_row0) / RT_BLUE_NSQRT;
    return tx < nd.tiles_x && ty < nd.tiles_y_local && nd.need[tx * nd.tiles_y_local + ty] == 0;
}
RT_DEV bool need_any(const RtNeedMap &nd, int x0, int x1, int n_src, int ty0, int ty1)
{
    if (ty0 < 0 || ty1 >= nd.tiles_y_local) return true;
    const int a = (x0 < 0 ? 0 : x0) / RT_BLUE_NSQRT, b = (x1 >= n_src ? n_src - 1 : x1) / RT_BLUE_NSQRT;
    for (int tx = a; tx <= b; tx++) {
        if (tx >= nd.tiles_x) return true;
        for (int ty = ty0; ty <= ty1; ty++) if (nd.need[tx * nd.tiles_y_local + ty]) return true;
    }
    return false;
}
__global__ void __launch_bounds__(256) k_resample_rows(const float *src, int planes, int rows, int n_src, int n_dst, RtTapTable T, float *dst, RtNeedMap nd)
{
    __shared__ float  s_src[RT_RS_SPAN + RT_RS_SPAN / 32 + 2];
    __shared__ double s_taps[RT_RS_TAPS];
    const int id0 = blockIdx.x * blockDim.x, id = id0 + threadIdx.x;
    const int row = blockIdx.y, pl = blockIdx.z;
    const float *s = src + ((size_t)pl * rows + row) * n_src;
    const int last = (id0 + 255 < n_dst ? id0 + 255 : n_dst - 1);
    const int base = T.first[id0], span = T.first[last] + T.count[last] - base;
    const bool staged = span > 0 && span <= RT_RS_SPAN;
    const bool utaps = T.uniform && T.max_count <= RT_RS_TAPS;
    /* lazy rows: which of the tiles this workgroup's span touches were written at all (one flag per tile, in LDS); when
       none was, every destination is the zero it would compute from zeros: written without staging anything */
    __shared__ unsigned char s_need[RT_RS_SPAN / RT_BLUE_NSQRT + 3];
    const bool lazy = staged && nd.lazy && row >= nd.tile_row0 && (row - nd.tile_row0) / RT_BLUE_NSQRT < nd.tiles_y_local;
    int t0 = 0;
    if (lazy) {
        const int xa = base < 0 ? 0 : (base >= n_src ? n_src - 1 : base), xe = base + span - 1, xb = xe < 0 ? 0 : (xe >= n_src ? n_src - 1 : xe);
        const int ty = (row - nd.tile_row0) / RT_BLUE_NSQRT, nt = xb / RT_BLUE_NSQRT - xa / RT_BLUE_NSQRT + 1;
        t0 = xa / RT_BLUE_NSQRT;
        int mine = 0;
        for (int k = threadIdx.x; k < nt; k += 256) {
            const int tx = t0 + k;
            const unsigned char v = tx < nd.tiles_x ? nd.need[tx * nd.tiles_y_local + ty] : (unsigned char)1;
            s_need[k] = v;
            mine |= v;
        }
        if (!__syncthreads_or(mine)) {
            if (id < n_dst) dst[((size_t)pl * rows + row) * n_dst + id] = 0.0f;
            return;
        }
    }
    if (staged)
        for (int j = threadIdx.x; j < span; j += 256) {
            int is = base + j, i = is;
            if ((unsigned)is >= (unsigned)n_src) i = is < 0 ? 0 : n_src - 1;
            s_src[RT_RS_IDX(j)] = (lazy && !s_need[i / RT_BLUE_NSQRT - t0]) ? 0.0f : s[i];
        }
    if (utaps) {
        const double *p0 = T.taps + T.offset[0];
        for (int k = threadIdx.x; k < T.max_count; k += 256) s_taps[k] = p0[k];
    }
    __syncthreads();
    if (id >= n_dst) return;
    const int i0 = T.first[id], cnt = T.count[id];
    if (nd.need && T.uniform && cnt > 0 && row >= nd.tile_row0) {
        const int ty = (row - nd.tile_row0) / RT_BLUE_NSQRT;
        if (!need_any(nd, i0, i0 + cnt - 1, n_src, ty, ty)) { dst[((size_t)pl * rows + row) * n_dst + id] = 0.0f; return; }
    }
    const double *pf = T.taps + T.offset[id];
    double sum = 0.0, wsum = 0.0;
    if (staged && utaps) {
        const int j0 = i0 - base;
        for (int k = 0; k < cnt; k++) {
            double y = (double)s_src[RT_RS_IDX(j0 + k)], fw = s_taps[k];
            sum += fw * y;
        }
        wsum = T.wsum_uniform;          /* every destination adds the same taps in the same order: summed once on the host */
    } else if (staged) {
        const int j0 = i0 - base;
        for (int k = 0; k < cnt; k++) {
            double y = (double)s_src[RT_RS_IDX(j0 + k)], fw = pf[k];
            sum += fw * y;
            wsum += fw;
        }
    } else {
        for (int k = 0; k < cnt; k++) {
            int is = i0 + k, i = is;
            if ((unsigned)is >= (unsigned)n_src) i = is < 0 ? 0 : n_src - 1;
            double y = need_unwritten(nd, i, row) ? 0.0 : (double)s[i], fw = pf[k];
            sum += fw * y;
            wsum += fw;
        }
    }
    if (wsum != 0.0) sum /= wsum;
    dst[((size_t)pl * rows + row) * n_dst + id] = (float)sum;
}

/* Splat gather and the horizontal resampling pass in one kernel: a workgroup produces the supersample pixels that 128
   output pixels of one supersample row read (their span plus the filter's reach), keeps the four planes in LDS and
   filters them from there -- the supersampled image (8.5 GB at 4K / 64 spp, written once and read once) never exists.
   Same gather (splat_pixel), same taps, same summation order as k_splat_gather + k_resample_rows. */
#define RT_FS_DST   128
#define RT_FS_SPAN  1152        /* source pixels per workgroup: 128 * ratio + taps, ratio <= 8.5 */
__global__ void __launch_bounds__(256) k_splat_resample_rows(const RtDevScene Sv, RtSampleGrid g, RtSplatTable ST, const float *sample_rgb,
                                                             int y0, int y1, int n_dst, RtTapTable T, float *dst3 /*[3][rows][n_dst]*/, float *dstw /*[rows][n_dst]*/)
{
    __shared__ float  s_px[4][RT_FS_SPAN + RT_FS_SPAN / 32 + 2];
    __shared__ double s_taps[RT_RS_TAPS];
    const int id0 = blockIdx.x * RT_FS_DST, row = blockIdx.y, py = y0 + row, n_src = g.nW;
    const int last = (id0 + RT_FS_DST - 1 < n_dst ? id0 + RT_FS_DST - 1 : n_dst - 1);
    const int base = T.first[id0], span = T.first[last] + T.count[last] - base;      /* host: 0 < span <= RT_FS_SPAN */
    const bool utaps = T.uniform && T.max_count <= RT_RS_TAPS;
    for (int j = threadIdx.x; j < span; j += 256) {
        int is = base + j, px = is;
        if ((unsigned)is >= (unsigned)n_src) px = is < 0 ? 0 : n_src - 1;           /* WRAP_CLAMP */
        float r, gg, b, w;
        splat_pixel(Sv, g, ST, sample_rgb, px, py, r, gg, b, w);
        const int q = RT_RS_IDX(j);
        s_px[0][q] = r; s_px[1][q] = gg; s_px[2][q] = b; s_px[3][q] = w;
    }
    if (utaps) {
        const double *p0 = T.taps + T.offset[0];
        for (int k = threadIdx.x; k < T.max_count; k += 256) s_taps[k] = p0[k];
    }
    __syncthreads();
    const int id = id0 + (threadIdx.x & (RT_FS_DST - 1)), half = threadIdx.x >> 7;      /* planes 2 * half, 2 * half + 1 */
    if (id >= n_dst) return;
    const int j0 = T.first[id] - base, cnt = T.count[id];
    const double *pf = T.taps + T.offset[id];
    double sum0 = 0.0, sum1 = 0.0, wsum = 0.0;
    const float *a0 = s_px[2 * half], *a1 = s_px[2 * half + 1];
    for (int k = 0; k < cnt; k++) {
        const double fw = utaps ? s_taps[k] : pf[k];
        const int q = RT_RS_IDX(j0 + k);
        sum0 += fw * (double)a0[q];
        sum1 += fw * (double)a1[q];
        if (!T.uniform) wsum += fw;
    }
    if (T.uniform) wsum = T.wsum_uniform;
    if (wsum != 0.0) { sum0 /= wsum; sum1 /= wsum; }
    const size_t rows = (size_t)(y1 - y0), o = (size_t)row * n_dst + id;
    if (half == 0) { dst3[o] = (float)sum0; dst3[rows * n_dst + o] = (float)sum1; }
    else { dst3[2 * rows * n_dst + o] = (float)sum0; dstw[o] = (float)sum1; }
}

/* The same, honouring the need map, with its tables in LDS and without the gather's idle lanes (option fuse_splat = 2, the default):
   - tiles of the span that cannot matter (RtSampleGrid::tile_need) are neither gathered nor filtered: their pixels are the zeros the
     two-kernel path would read, and a workgroup whose span holds no needed tile writes its zeros and leaves;
   - the gather is a chain of dependent look-ups per splat (cell -> entry -> sample position -> colour) through tables of 58 KB that
     do not stay in the vector L1: measured, 5.4 of this kernel's first 8.4 ms.  A supersample row only ever needs ONE row of cells of
     the splat table (two on a tile's first and last row, for the neighbouring tile's overhang): those cells' offsets and entries --
     with the sample's position spelled out, RtSplatEntry -- are staged in LDS (about 4 KB), and only the colour is read from memory;
   - a thread per supersample pixel leaves lanes idle: the pixels of a row receive between none and ten splats (3.6 on average) and
     the wave runs as long as its busiest lane.  The number of splats a pixel receives depends on its position within its 40 x 40
     tile alone, so the lanes are dealt pixels of EQUAL position from different tiles of the span: thread u takes the pixel of rank
     u / nn in the order `order[ly][.]` (positions of row ly of a tile, most splats first) of needed tile u % nn.
   Same arithmetic and the same summation order as splat_pixel + k_resample_rows: bit-identical planes. */
#define RT_FS_ROWENT 224        /* entries of one cell row the staging buffer holds, twice (the host checks: RtSplatTable::max_row_entries) */
#define RT_FS_OFFS (RT_BLUE_NSQRT + 4)
struct SplatRowLds { const int *off; const RtSplatEntry *ent; int cy1; };       /* off: [2][RT_FS_OFFS], already relative to ent; cy1: the second cell row (-99: none) */
RT_DEV void splat_pixel_lds(const RtSampleGrid &g, const SplatRowLds &L, const float *sample_rgb, int px, int py, float &r, float &gg, float &b, float &w)
{
    r = 0.0f; gg = 0.0f; b = 0.0f; w = 0.0f;
    const int tcx = px / RT_BLUE_NSQRT, tcy = py / RT_BLUE_NSQRT;
    for (int tx = tcx - 1; tx <= tcx + 1; tx++) {
        if (tx < 0 || tx >= g.tiles_x) continue;
        const int lx = px - tx * RT_BLUE_NSQRT;
        if (lx < -1 || lx > RT_BLUE_NSQRT) continue;
        for (int ty = tcy - 1; ty <= tcy + 1; ty++) {
            if (ty < g.tile_y0 || ty >= g.tile_y0 + g.tiles_y_local) continue;
            const int ly = py - ty * RT_BLUE_NSQRT;
            if (ly < -1 || ly > RT_BLUE_NSQRT) continue;
            const int so = ly == L.cy1 ? RT_FS_OFFS : 0;       /* which staged cell row: ly is the row itself or cy1 by construction */
            const int e0 = L.off[so + lx + 1], e1 = L.off[so + lx + 2];
            const int tile = tx * g.tiles_y_local + (ty - g.tile_y0);
#ifdef RT_RECON_DEBUG
            const int64_t sbase = (g.pad_ & 4) ? 0 : (int64_t)tile * RT_BLUE_N;      /* (debug: every lane reads the first tile's colours) */
#else
            const int64_t sbase = (int64_t)tile * RT_BLUE_N;
#endif
            const bool black = g.tile_mask && g.mask_skips_rgb && g.tile_mask[tile] == 0;
            for (int e = e0; e < e1; e++) {
                const RtSplatEntry en = L.ent[e];
                /* v = g_rgvBlueNoise[i]*BLUE_NSQRT + vTile (Sampling.cpp:501), then FastSplatRGB's own arithmetic: as splat_pixel */
                double x = en.bx + (double)tx * (double)RT_BLUE_NSQRT;
                x -= 0.5;
                const double fx = floor(x);
                const int i1 = (int)fx;
                const double dx = x - fx;
                double wx;
                if (px == i1) wx = 1.0 - dx; else if (px == i1 + 1) wx = dx; else continue;
                const double wy = en.by;            /* the row's weight: the same for every tile of the span, formed once when the row was staged */
                if (wy < 0.0) continue;
                w += (float)(wx * wy);
                if (black) continue;
                const float *c = sample_rgb + 3 * (sbase + en.i);
                r += (float)((double)(float)((double)c[0] * wx) * wy);
                gg += (float)((double)(float)((double)c[1] * wx) * wy);
                b += (float)((double)(float)((double)c[2] * wx) * wy);
            }
        }
    }
}
__global__ void __launch_bounds__(256) k_splat_rows_need(RtSampleGrid g, RtSplatTable ST, const float *sample_rgb,
                                                         int y0, int y1, int n_dst, RtTapTable T, float *dst3 /*[3][rows][n_dst]*/, float *dstw /*[rows][n_dst]*/)
{
    __shared__ float  s_px[4][RT_FS_SPAN + RT_FS_SPAN / 32 + 2];
    __shared__ double s_taps[RT_RS_TAPS];
    __shared__ RtSplatEntry s_ent[2 * RT_FS_ROWENT];
    __shared__ int    s_off[2 * RT_FS_OFFS];
    __shared__ unsigned char s_need[64];
    __shared__ short  s_tiles[64];
    __shared__ unsigned char s_ord[RT_BLUE_NSQRT];
    const int id0 = blockIdx.x * RT_FS_DST, row = blockIdx.y, py = y0 + row, n_src = g.nW;
    const int last = (id0 + RT_FS_DST - 1 < n_dst ? id0 + RT_FS_DST - 1 : n_dst - 1);
    const int base = T.first[id0], span = T.first[last] + T.count[last] - base;      /* host: 0 < span <= RT_FS_SPAN */
    const bool utaps = T.uniform && T.max_count <= RT_RS_TAPS;
    const size_t rows = (size_t)(y1 - y0);
    /* the pixels of the span that exist: [xa, xb]; the rest of the span repeats an edge pixel (WRAP_CLAMP) */
    const int xa = base < 0 ? 0 : base, xe = base + span - 1, xb = xe >= n_src ? n_src - 1 : xe;
    const int t0 = xa / RT_BLUE_NSQRT, nt = xb / RT_BLUE_NSQRT - t0 + 1;           /* <= 30 tiles */
    const int tyl = py / RT_BLUE_NSQRT - g.tile_y0, ly = py - (py / RT_BLUE_NSQRT) * RT_BLUE_NSQRT;
    const bool have_need = g.tile_need && tyl >= 0 && tyl < g.tiles_y_local;
    const int id = id0 + (threadIdx.x & (RT_FS_DST - 1)), half = threadIdx.x >> 7;      /* planes 2 * half, 2 * half + 1 */
    /* which tiles of the span can matter: one flag per tile, loaded by the first wave, compacted with a ballot */
    int nn;
    {
        unsigned char v = 0;
        if (threadIdx.x < nt) {
            const int tx = t0 + (int)threadIdx.x;
            v = (have_need && tx < g.tiles_x) ? g.tile_need[tx * g.tiles_y_local + tyl] : (unsigned char)1;
        }
        if (threadIdx.x < 64) {
            const unsigned long long m = __ballot(v != 0);
            s_need[threadIdx.x] = v;
            if (v) s_tiles[__popcll(m & ((1ull << threadIdx.x) - 1ull))] = (short)threadIdx.x;
        }
        nn = __syncthreads_count(v != 0);
    }
    if (nn == 0) {                      /* nothing in reach can matter: the zeros the filter would produce from zeros */
        if (id < n_dst) {
            const size_t o = (size_t)row * n_dst + id;
            if (half == 0) { dst3[o] = 0.0f; dst3[rows * n_dst + o] = 0.0f; }
            else { dst3[2 * rows * n_dst + o] = 0.0f; dstw[o] = 0.0f; }
        }
        return;
    }
    /* the cell rows this supersample row reads: its own, and the overhang row of the tile above (ly == 0) or below (ly == 39) */
    SplatRowLds L;
    L.off = s_off; L.ent = s_ent;
    L.cy1 = ly == 0 ? RT_BLUE_NSQRT : (ly == RT_BLUE_NSQRT - 1 ? -1 : -99);
    {
        int at = 0;
        for (int q = 0; q < (L.cy1 == -99 ? 1 : 2); q++) {
            const int c0 = ((q ? L.cy1 : ly) + 1) * (RT_BLUE_NSQRT + 2);        /* first cell of the row: (cy + 1) * 42 + (cx + 1), cx = -1 */
            const int eb = ST.offsets[c0], ee = ST.offsets[c0 + RT_BLUE_NSQRT + 2];
            for (int k = threadIdx.x; k <= RT_BLUE_NSQRT + 2; k += 256) s_off[q * RT_FS_OFFS + k] = ST.offsets[c0 + k] - eb + at;
            /* ... with `by` replaced by the sample's weight for THIS row (-1: it does not reach the row): the tile row the cell row belongs
               to is the same for every tile of the span, so FastSplatRGB's arithmetic on y (Sampling.cpp:501 and the splat's own, as
               splat_pixel does it) is done once per entry here instead of once per entry and pixel */
            const int ty_q = (py - (q ? L.cy1 : ly)) / RT_BLUE_NSQRT;
            for (int k = threadIdx.x; k < ee - eb; k += 256) {
                RtSplatEntry en = ST.entries_pos[eb + k];
                double y = en.by + (double)ty_q * (double)RT_BLUE_NSQRT;
                y -= 0.5;
                const double fy = floor(y);
                const int j1 = (int)fy;
                const double dy = y - fy;
                en.by = py == j1 ? 1.0 - dy : (py == j1 + 1 ? dy : -1.0);
                s_ent[at + k] = en;
            }
            at += ee - eb;
        }
    }
    if (threadIdx.x < RT_BLUE_NSQRT) s_ord[threadIdx.x] = ST.order[ly * RT_BLUE_NSQRT + threadIdx.x];
    for (int j = threadIdx.x; j < span; j += 256) {         /* zeros everywhere first: pixels of tiles that cannot matter stay zero */
        const int q = RT_RS_IDX(j);
        s_px[0][q] = 0.0f; s_px[1][q] = 0.0f; s_px[2][q] = 0.0f; s_px[3][q] = 0.0f;
    }
    if (utaps) {
        const double *p0 = T.taps + T.offset[0];
        for (int k = threadIdx.x; k < T.max_count; k += 256) s_taps[k] = p0[k];
    }
    __syncthreads();
    for (int u = threadIdx.x; u < nn * RT_BLUE_NSQRT; u += 256) {
        const int rank = u / nn, k = s_tiles[u - rank * nn];
        const int px = (t0 + k) * RT_BLUE_NSQRT + s_ord[rank];
        if (px < xa || px > xb) continue;
        float r, gg, b, w;
#ifdef RT_RECON_DEBUG
        if (g.pad_ & 2) { r = gg = b = w = 1.0f; } else
#endif
        splat_pixel_lds(g, L, sample_rgb, px, py, r, gg, b, w);
        const int q = RT_RS_IDX(px - base);
        s_px[0][q] = r; s_px[1][q] = gg; s_px[2][q] = b; s_px[3][q] = w;
    }
    __syncthreads();
    if (base < 0 || xe >= n_src) {      /* WRAP_CLAMP at the image's edges */
        for (int j = threadIdx.x; j < span; j += 256) {
            const int is = base + j;
            if ((unsigned)is < (unsigned)n_src) continue;
            const int qs = RT_RS_IDX((is < 0 ? 0 : n_src - 1) - base), q = RT_RS_IDX(j);
            s_px[0][q] = s_px[0][qs]; s_px[1][q] = s_px[1][qs]; s_px[2][q] = s_px[2][qs]; s_px[3][q] = s_px[3][qs];
        }
        __syncthreads();
    }
    if (id >= n_dst) return;
    const int i0 = T.first[id], j0 = i0 - base, cnt = T.count[id];
    const size_t o = (size_t)row * n_dst + id;
    if (T.uniform && cnt > 0) {        /* a destination that reads zeros only: the zero it would compute */
        const int a = (i0 < 0 ? 0 : (i0 >= n_src ? n_src - 1 : i0)) / RT_BLUE_NSQRT - t0, e = i0 + cnt - 1;
        const int bq = (e < 0 ? 0 : (e >= n_src ? n_src - 1 : e)) / RT_BLUE_NSQRT - t0;
        bool any = false;
        for (int k = a; k <= bq; k++) any = any || s_need[k] != 0;
        if (!any) {
            if (half == 0) { dst3[o] = 0.0f; dst3[rows * n_dst + o] = 0.0f; }
            else { dst3[2 * rows * n_dst + o] = 0.0f; dstw[o] = 0.0f; }
            return;
        }
    }
    const double *pf = T.taps + T.offset[id];
    double sum0 = 0.0, sum1 = 0.0, wsum = 0.0;
    const float *a0 = s_px[2 * half], *a1 = s_px[2 * half + 1];
#ifdef RT_RECON_DEBUG
    if (!(g.pad_ & 1))
#endif
    for (int k = 0; k < cnt; k++) {
        const double fw = utaps ? s_taps[k] : pf[k];
        const int q = RT_RS_IDX(j0 + k);
        sum0 += fw * (double)a0[q];
        sum1 += fw * (double)a1[q];
        if (!T.uniform) wsum += fw;
    }
    if (T.uniform) wsum = T.wsum_uniform;
    if (wsum != 0.0) { sum0 /= wsum; sum1 /= wsum; }
    if (half == 0) { dst3[o] = (float)sum0; dst3[rows * n_dst + o] = (float)sum1; }
    else { dst3[2 * rows * n_dst + o] = (float)sum0; dstw[o] = (float)sum1; }
}

/* vertical pass: src [planes][src_rows_local][width] (rows src_y0..), dst rows [row0,row1) of [planes][H][width] */
__global__ void __launch_bounds__(256) k_resample_cols(const float *src, int planes, int src_rows_local, int src_y0, int n_src_total, int width,
                                                       int row0, int row1, RtTapTable T, float *dst, size_t dst_plane_stride,
                                                       RtNeedMap nd, const int *first_x, const int *count_x, int n_src_x)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = row0 + blockIdx.y;
    int pl = blockIdx.z;
    if (i >= width || j >= row1 || pl >= planes) return;
    int j0 = T.first[j], cnt = T.count[j];
    if (nd.need && T.uniform && cnt > 0) {
        /* the row-pass values this destination reads come from supersample pixels [first_x[i], +count_x[i]) of rows [j0, j0 + cnt) */
        const int ya = (j0 < 0 ? 0 : j0) - src_y0, yb = (j0 + cnt - 1 >= n_src_total ? n_src_total - 1 : j0 + cnt - 1) - src_y0;
        if (ya >= nd.tile_row0 && !need_any(nd, first_x[i], first_x[i] + count_x[i] - 1, n_src_x, (ya - nd.tile_row0) / RT_BLUE_NSQRT, (yb - nd.tile_row0) / RT_BLUE_NSQRT)) {
            dst[(size_t)pl * dst_plane_stride + (size_t)j * width + i] = 0.0f;
            return;
        }
    }
    const double *pf = T.taps + T.offset[j];
    double sum = 0.0, wsum = 0.0;
    for (int k = 0; k < cnt; k++) {
        int js = j0 + k, jj = js;
        if ((unsigned)js >= (unsigned)n_src_total) jj = js < 0 ? 0 : n_src_total - 1;
        double y = (double)src[((size_t)pl * src_rows_local + (size_t)(jj - src_y0)) * width + i], fw = pf[k];
        sum += fw * y;
        if (!T.uniform) wsum += fw;
    }
    if (T.uniform) wsum = T.wsum_uniform;
    if (wsum != 0.0) sum /= wsum;
    dst[(size_t)pl * dst_plane_stride + (size_t)j * width + i] = (float)sum;
}

/* Image::Normalize (Image.cpp:1238-1260) on rows [row0,row1) */
__global__ void __launch_bounds__(256) k_normalize(float *planes, const float *wt, int width, int height, int row0, int row1)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = row0 + blockIdx.y;
    if (i >= width || j >= row1) return;
    float w = wt[(size_t)j * width + i];
    if (w != 0.0f)
        for (int k = 0; k < 3; k++) planes[((size_t)k * height + j) * width + i] /= w;
}

/* TestSamplingTile: one sample per pixel, SetRGB (Sampling.cpp:335-346) */
__global__ void __launch_bounds__(256) k_store_test(const float *sample_rgb, int width, int height, int row0, int row1, float *planes)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = row0 + blockIdx.y;
    if (i >= width || j >= row1) return;
    const float *c = sample_rgb + 3 * ((size_t)(j - row0) * width + i);
    for (int k = 0; k < 3; k++) planes[((size_t)k * height + j) * width + i] = c[k];
}

/* DebugSamplingTile: im.FillRGB(rgb) with the one sample (Sampling.cpp:369) */
__global__ void __launch_bounds__(256) k_fill_debug(const float *sample_rgb, int width, int height, int row0, int row1, float *planes)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = row0 + blockIdx.y;
    if (i >= width || j >= row1) return;
    for (int k = 0; k < 3; k++) planes[((size_t)k * height + j) * width + i] = sample_rgb[k];
}

/* JitterSamplingTile: im2.SetRGB(rad.sRGB(), i, j) (Sampling.cpp:412): per-sample RGB in row-major sample order -> planes */
__global__ void __launch_bounds__(256) k_store_jitter(const float *sample_rgb, int nW, int rows, float *im2 /*[3][rows][nW]*/)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, jl = blockIdx.y;
    if (i >= nW) return;
    size_t o = (size_t)jl * nW + i, plane = (size_t)rows * nW;
    im2[o] = sample_rgb[3 * o]; im2[plane + o] = sample_rgb[3 * o + 1]; im2[2 * plane + o] = sample_rgb[3 * o + 2];
}

/* ------------------------------------------------------------------ output stage: Image::WriteBMP, Image.cpp:416-474
 * One thread per pixel: ClampToGamut (Image.cpp:293-327) when a channel leaves [0,1], then GammaEncode's
 * eight-step search (Image.h:206-220) of the 255 thresholds SetGamma built (host, Image.cpp:498-517), written as
 * bottom-up BGR lines of `line` bytes.  Comparisons and float arithmetic only: byte-exact against the CPU. */
struct RtGammaTable { float a[256]; };
RT_DEV int gamma_encode(const float *A, double f)
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
__global__ void __launch_bounds__(256) k_bmp_encode(const float *planes, int width, int height, RtGammaTable G, int line, unsigned char *pixels)
{
    __shared__ float A[256];
    A[threadIdx.x] = G.a[threadIdx.x];
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (i >= width) return;
    int j = height - 1 - row;
    size_t plane = (size_t)width * height, o = (size_t)j * width + i;
    float red = planes[o], grn = planes[plane + o], blu = planes[2 * plane + o];
    if (red < 0.0 || red > 1.0 || grn < 0.0 || grn > 1.0 || blu < 0.0 || blu > 1.0) {
        float fLum = 0.2126f * red + 0.7152f * grn + 0.0722f * blu, t;      /* Luminance(), Image.h:110-115 */
        if (fLum < 0.0) red = grn = blu = 0.0f;
        else if (fLum > 1.0) red = grn = blu = 1.0f;
        else {
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
    }
    unsigned char *p = pixels + (size_t)row * line + 3 * (size_t)i;
    p[0] = (unsigned char)gamma_encode(A, blu);
    p[1] = (unsigned char)gamma_encode(A, grn);
    p[2] = (unsigned char)gamma_encode(A, red);
}

/* ------------------------------------------------------------------ output stage: Video::WriteFrame, Video.cpp:49-80
 * One thread per pixel: ClampToGamut when a channel leaves [0,1], pow(c, 1.0f/m_fGamma) in float, the BT.601
 * studio-range matrix in double, int() truncation, three planes Y, Cb, Cr of W*H bytes (C444).  The float pow is
 * the only libm call: here it is the double pow rounded to float (the correctly rounded value in all but
 * astronomically rare cases); a CPU powf may differ from that by one ulp, which moves a byte by one code at most. */
__global__ void __launch_bounds__(256) k_y4m_encode(const float *planes, int width, int height, float inv_gamma, unsigned char *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= width) return;
    size_t plane = (size_t)width * height, o = (size_t)j * width + i;
    float red = planes[o], grn = planes[plane + o], blu = planes[2 * plane + o];
    if (red < 0.0 || red > 1.0 || grn < 0.0 || grn > 1.0 || blu < 0.0 || blu > 1.0) {
        float fLum = 0.2126f * red + 0.7152f * grn + 0.0722f * blu, t;      /* Luminance(), Image.h:110-115 */
        if (fLum < 0.0) red = grn = blu = 0.0f;
        else if (fLum > 1.0) red = grn = blu = 1.0f;
        else {
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
    }
    red = (float)pow((double)red, (double)inv_gamma);
    grn = (float)pow((double)grn, (double)inv_gamma);
    blu = (float)pow((double)blu, (double)inv_gamma);
    double Y  =  16 +  65.481 * red + 128.553 * grn +  24.966 * blu;
    double Cb = 128 -  37.797 * red -  74.203 * grn + 112.000 * blu;
    double Cr = 128 + 112.000 * red -  93.786 * grn -  18.214 * blu;
    out[o] = (unsigned char)(int)Y;
    out[plane + o] = (unsigned char)(int)Cb;
    out[2 * plane + o] = (unsigned char)(int)Cr;
}

#if defined(RT_ONLY_FAST)     /* tools/kernel_usage.sh: a quick look at one kernel's registers without building the whole library */
template __global__ void k_fast<1>(const RtDevScene, RtSampleGrid, RtPipe, int64_t, float *, float *);
template __global__ void k_fast_hit<1>(const RtDevScene, RtSampleGrid, RtPipe, int64_t, float *, float *);
template __global__ void k_fast_shade<1>(const RtDevScene, RtSampleGrid, RtPipe, int64_t, float *, float *);
#elif defined(RT_ONLY_VISUAL)
template __global__ void k_visual<true, 2, false>(const RtDevScene, RtPipe, int, int, int);
template __global__ void k_visual<true, 1, true>(const RtDevScene, RtPipe, int, int, int);
#else
#include "rt_api.inc"
#endif
