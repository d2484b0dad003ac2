/*
 * rt_core.hip.h -- device functions of the render path (gfx950): double-precision geometry, root
 * finders, primitive intersection, the per-ray CSG stack machine, normal transforms + bump noise,
 * Fresnel / BRDF / propagators.  Arithmetic follows the reference expression by expression
 * (file:line cited at each function) and the translation unit is compiled with -ffp-contract=off so
 * that results round exactly as on the host: +,-,*,/ and sqrt are IEEE-exact in f64 and f32 on CDNA4;
 * only exp/acos/cos/sin (ColoredGlass, the three-real-root branch of the torus cubic, Wavy) go through
 * the device math library and may differ from glibc in the last ulp.
 *
 * No MFMA: nothing here is a dense contraction.  The hot loop is branchy f64 VALU work over a scene
 * that fits in L2; hit lists live in per-lane private memory (scratch is lane-interleaved by the
 * hardware, so uniform indices coalesce).
 */
#pragma once
#include <hip/hip_runtime.h>
#include "rt_device.h"

#define RT_DEV __device__ __forceinline__
#define RT_DEVN __device__ __noinline__
#define RT_HD __host__ __device__ inline

struct D3 { double x, y, z; };
RT_DEV D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV D3 operator+(const D3 &a, const D3 &b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV D3 operator-(const D3 &a, const D3 &b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV D3 operator-(const D3 &a) { return d3(-a.x, -a.y, -a.z); }
RT_DEV D3 operator*(const D3 &a, double f) { return d3(a.x * f, a.y * f, a.z * f); }
RT_DEV D3 operator/(const D3 &a, double f) { return d3(a.x / f, a.y / f, a.z / f); }
RT_DEV double dot(const D3 &u, const D3 &v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
RT_DEV D3 cross(const D3 &u, const D3 &v) { return d3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
RT_DEV double norm(const D3 &v) { return sqrt(dot(v, v)); }
RT_DEV D3 normalize(const D3 &v) { return v / norm(v); }
/* Matrix3 * Vector3 (Affine.h:191-205) and CoVector3 * Matrix3 (Affine.h:286-292), m row major */
RT_DEV D3 mat_vec(const double *m, const D3 &v)
{
    return d3(v.x * m[0] + v.y * m[1] + v.z * m[2], v.x * m[3] + v.y * m[4] + v.z * m[5], v.x * m[6] + v.y * m[7] + v.z * m[8]);
}
RT_DEV D3 covec_mat(const D3 &c, const double *m)
{
    return d3(c.x * m[0] + c.y * m[3] + c.z * m[6], c.x * m[1] + c.y * m[4] + c.z * m[7], c.x * m[2] + c.y * m[5] + c.z * m[8]);
}

struct DRay { D3 org, dir; };
RT_DEV D3 ray_at(const DRay &r, double t) { return r.org + r.dir * t; }

struct Spec { float v[RT_NSPEC]; };
RT_DEV Spec spec_const(float f) { Spec s; for (int i = 0; i < RT_NSPEC; i++) s.v[i] = f; return s; }
RT_DEV Spec spec_load(const float *p) { Spec s; for (int i = 0; i < RT_NSPEC; i++) s.v[i] = p[i]; return s; }
RT_DEV Spec operator+(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < RT_NSPEC; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
RT_DEV Spec operator*(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < RT_NSPEC; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
RT_DEV Spec operator*(const Spec &a, double f) { Spec r; for (int i = 0; i < RT_NSPEC; i++) r.v[i] = (float)((double)a.v[i] * f); return r; }
RT_DEV Spec spec_div(const Spec &a, double f) { Spec r; if (f != 0.0) f = 1.0 / f; for (int i = 0; i < RT_NSPEC; i++) r.v[i] = (float)((double)a.v[i] * f); return r; }
RT_DEV bool spec_is_black(const Spec &a) { for (int i = 0; i < RT_NSPEC; i++) if (a.v[i] != 0.0f) return false; return true; }

#define RT_PI    3.14159265358979323846264338327950288419716939937510
#define RT_SQRT3 1.73205080756887729352744634150587236694280525381038

/* ================================================================== root finders (Solids.cpp:277-561) */
RT_DEV int solve_linear(double *r, double a, double b)
{
    if (a == 0.0) {
        if (b == 0.0) { r[0] = 0.0; return 1; }
        return 0;
    }
    r[0] = -b / a;
    return 1;
}

RT_DEV int solve_quadratic(double *r, double a, double b, double c)
{
    if (a == 0.0) return solve_linear(r, b, c);
    double disc = b * b - 4.0 * a * c;
    if (disc <= 0.0) {
        if (disc == 0.0) { r[0] = r[1] = -b / (a + a); return 2; }
        return 0;
    }
    disc = sqrt(disc);
    if (a < 0.0) { a = -a; b = -b; c = -c; }
    if (b > 0.0) {
        r[0] = (-b - disc) / (a + a);
        r[1] = (c + c) / (-b - disc);
    } else {
        r[1] = (-b + disc) / (a + a);
        r[0] = (c + c) / (-b + disc);
    }
    return 2;
}

__device__ double cube_root(double x)
{
    const double c2 = 1.259921049894873164767210607278228350570251, c4 = 1.587401051968199474751705639272308260391493;
    const double c2i = 0.7937005259840997373758528196361541301957467, c4i = 0.6299605249474365823836053036391141752851257;
    if (x == 0) return x;
    int sign = 1, e, rem;
    if (!(x > 0)) { sign = -1; x = -x; }
    double z = x;
    x = frexp(x, &e);
    x = (((((1.3584464340920900529734e-1) * x - (6.3986917220457538402318e-1)) * x + (1.2875551670318751538055e0)) * x
          - (1.4897083391357284957891e0)) * x + (1.3304961236013647092521e0)) * x + (3.7568280825958912391243e-1);
    if (e >= 0) {
        rem = e; e /= 3; rem -= 3 * e;
        if (rem == 1) x *= c2; else if (rem == 2) x *= c4;
    } else {
        e = -e; rem = e; e /= 3; rem -= 3 * e;
        if (rem == 1) x *= c2i; else if (rem == 2) x *= c4i;
        e = -e;
    }
    x = ldexp(x, e);
    x -= (x - (z / (x * x))) / 3.0;
    x -= (x - (z / (x * x))) / 3.0;
    x -= (x - (z / (x * x))) / 3.0;
    return sign < 0 ? -x : x;
}

__device__ int solve_cubic(double *r, double a, double b, double c, double d)
{
    if (a == 0.0) return solve_quadratic(r, b, c, d);
    double A = b / a, B = c / a, C = d / a;
    double yi2 = 0.5 * (A * (2.0 * A * A - 9.0 * B) + 27.0 * C);
    double ypi3 = 3.0 * B - A * A;
    double disc = ypi3 * ypi3 * ypi3 + yi2 * yi2;
    double S, S3, T, U, V, fc, fs;
    if (disc > 0.0) {
        U = cube_root(fabs(yi2) + sqrt(disc));
        if (yi2 > 0.0) U = -U;
        V = -ypi3 / U;
        r[0] = (U + V - A) * 1.0 / 3.0;
        return 1;
    }
    S = sqrt(-ypi3);
    S3 = S * S * S;
    bool one = false;
    if (yi2 >= S3) {
        if (yi2 > S3) one = true;
        fc = 1.0; fs = 0.0;
    } else if (yi2 <= -S3) {
        if (yi2 < -S3) one = true;
        fc = 0.5; fs = 1.5;
    } else {
        T = yi2 / S3;
        double th = acos(T) / 3.0;
        fc = cos(th);
        fs = RT_SQRT3 * sin(th);
    }
    if (!one) {
        r[0] = (-2.0 * S * fc - A) * 1.0 / 3.0;
        r[1] = (S * (fc - fs) - A) * 1.0 / 3.0;
        r[2] = (S * (fc + fs) - A) * 1.0 / 3.0;
        return 3;
    }
    T = yi2 / S3;
    U = cube_root(T + sqrt(T * T - 1.0));
    r[0] = (-S * (U + 1.0 / U) - A) * 1.0 / 3.0;
    return 1;
}

__device__ int solve_quartic(double *r, double a, double b, double c, double d, double e)
{
    if (a == 0.0) return solve_cubic(r, b, c, d, e);
    double A = b / a, B = c / a, C = d / a, D = e / a;
    double P = B, Q = A * C - 4.0 * D, R = D * (A * A - 4.0 * B) + C * C;
    double cr[3];
    (void)solve_cubic(cr, 1.0, P, Q, R);
    double y = cr[0];
    double EE = 0.25 * A * A - B - y;
    double FF = 0.25 * y * y - D;
    double EF = -0.5 * A * y - C;
    double E = EE < 0.0 ? 0.0 : sqrt(EE);
    double F = FF < 0.0 ? 0.0 : sqrt(FF);
    if (EF < 0.0) F = -F;
    double G = 0.5 * A + E, g = 0.5 * A - E, H, h;
    if (y * F >= 0.0) {
        h = -0.5 * y - F;
        if (h != 0.0) H = D / h; else H = -0.5 * y + F;
    } else {
        H = -0.5 * y + F;
        h = D / H;
    }
    int n = solve_quadratic(r, 1.0, G, H);
    n += solve_quadratic(r + n, 1.0, g, h);
    for (int j = 1; j < n; j++) {
        double x = r[j];
        int i;
        for (i = j - 1; i >= 0 && x < r[i]; --i) r[i + 1] = r[i];
        r[i + 1] = x;
    }
    return n;
}

/* ================================================================== per-ray hit storage */
struct __attribute__((aligned(16))) HitRec {
    double   t;
    D3       n;
    uint32_t leaf_affine;   /* index + 1, 0 = none */
    uint16_t surf, mat;
    uint16_t prim;          /* primitive index, or 0x8000 | clip index for an AngularClip wedge */
    uint8_t  ntex, pad_;
    uint32_t texpack;       /* up to RT_MAX_TEX (4) texture call-out indices, one byte each, first queued in the low byte */
};
#define HIT_TEX(h, k) (((h).texpack >> (8 * (k))) & 0xFFu)
RT_DEV void hit_add_tex(HitRec &h, int tex) { if (tex >= 0 && h.ntex < RT_MAX_TEX) { h.texpack |= (uint32_t)tex << (8 * h.ntex); h.ntex++; } }


/* which dynamic limit a ray ran into (Arena::overflow, reported in the error text) */
enum { OVF_ARENA = 1, OVF_LISTS = 2, OVF_REGIONS = 4, OVF_CAND = 8, OVF_BVH_STACK = 16, OVF_MESH_STACK = 32 };
/* Where a ray's hit records live.  POOL = false: private (scratch) memory -- lane-interleaved by the hardware, so lanes
   that use the same index coalesce.  Every kernel of the default build.  POOL = true (-DRT_SCHED_POOL, an A/B build): the
   scheduled FULL kernels keep the records in a pool in global memory (RtPipe::arena_pool), one record per 64-byte sector of the
   lane's own RT_ARENA sectors; list stacks and the interpreter's arrays stay private (their indices follow the pc, which the
   lanes of a round share).  Round 2 ran such a pool (with everything in it) under the class scheduler, where seven lanes ran
   at a time; under the min-pc order (intersect_scene) the private layout is as good: S4 1080p / 4 spp 463 ms private, 455 ms
   with this pool and its 4.3 GB, and the same 9.5 KB per ray through the fabric either way (tools/pmc_s4_final.sh) -- the
   records are not what that traffic is made of. */
template <bool POOL> struct ArenaStore {
    HitRec   h[RT_ARENA];
    uint16_t l_start[RT_LIST_STACK], l_count[RT_LIST_STACK];
    int32_t  l_tag[RT_LIST_STACK];
    int32_t  l_gap[RT_LIST_STACK];      /* union regions: LCA depth between this entry and the one below it */
    uint16_t l_idx[RT_LIST_STACK];      /* union regions: last region child merged into this entry */
};
/* a lane's part of the pool: its hit records alone, one per 64-byte sector (a record is 48 bytes: three 16-byte accesses that
   stay inside one sector, whatever the other lanes' indices are) */
#define RT_POOL_REC     64
#define RT_POOL_STRIDE  (RT_ARENA * RT_POOL_REC)
struct HitPool { char *p; RT_DEV HitRec &operator[](int i) const { return *(HitRec *)(p + (size_t)i * RT_POOL_REC); } };
template <> struct ArenaStore<true> {
    HitPool  h;
    uint16_t l_start[RT_LIST_STACK], l_count[RT_LIST_STACK];
    int32_t  l_tag[RT_LIST_STACK];
    int32_t  l_gap[RT_LIST_STACK];
    uint16_t l_idx[RT_LIST_STACK];
    RT_DEV void bind(void *pool, size_t lane) { h.p = (char *)pool + lane * RT_POOL_STRIDE; }
};
template <bool POOL>
struct ArenaT : ArenaStore<POOL> {
    static constexpr bool pool = POOL;
    int      nlist;
    int      t_start, t_count, t_tag;   /* register mirror of the top list entry (valid when nlist > 0): avoids store -> load round trips */
    int      top;           /* first free hit slot */
    int      overflow;
    int      bail;          /* LITE interpreter met a rare, register-hungry feature: redo this ray in the FULL kernel */
    int      mesh_slot;     /* >= 0: this ray's mesh crossings were found by k_mesh_cross, entry mesh_slot of SceneLds::mx_* */
#ifdef RT_PHASE_PROF
    int      prof[12];      /* cycles: [1] BVH candidate search, [2] primitive tests, [3] whole intersect, [4] chain setup, [5] wrappers, [6] region fold, [7] cull_ray */
#endif
};
typedef ArenaT<false> Arena;
/* The FULL kernels' interpreter is a function of its own and gets the arena by reference: its counters (nlist, top, the mirror of
   the top list entry ...) then live in the caller's frame, every look at them is a load from memory behind whatever stores are in
   flight, and every 48-byte record copy -- which the compiler has to assume may overlap them -- makes it read them again.  The
   function therefore works on a VIEW: the arrays through pointers, the counters by value in a local object whose address never
   leaves the function (intersect_scene_full inlines all users), so they are registers; the two rare callees that are functions of
   their own (torus, mesh walk) get the caller's arena, brought up to date around the call (arena_call_out). */
struct ArenaView {
    static constexpr bool pool = false;
    HitRec   *h;
    uint16_t *l_start, *l_count, *l_idx;
    int32_t  *l_tag, *l_gap;
    ArenaT<false> *home;
    int      nlist, t_start, t_count, t_tag, top, overflow, bail, mesh_slot;
#ifdef RT_PHASE_PROF
    int      prof[12];
#endif
    RT_DEV void load(ArenaT<false> &a)
    {
        nlist = a.nlist; t_start = a.t_start; t_count = a.t_count; t_tag = a.t_tag; top = a.top; overflow = a.overflow; bail = a.bail; mesh_slot = a.mesh_slot;
#ifdef RT_PHASE_PROF
        for (int q = 0; q < 12; q++) prof[q] = a.prof[q];
#endif
    }
    RT_DEV void commit() const
    {
        ArenaT<false> &a = *home;
        a.nlist = nlist; a.t_start = t_start; a.t_count = t_count; a.t_tag = t_tag; a.top = top; a.overflow = overflow; a.bail = bail; a.mesh_slot = mesh_slot;
#ifdef RT_PHASE_PROF
        for (int q = 0; q < 12; q++) a.prof[q] = prof[q];
#endif
    }
};
template <class F> RT_DEV void arena_call_out(ArenaT<false> &A, F f) { f(A); }
template <class F> RT_DEV void arena_call_out(ArenaView &A, F f) { A.commit(); f(*A.home); A.load(*A.home); }
#ifdef RT_PHASE_PROF
#define PROF_BEGIN(v)      long long v = clock64()
#define PROF_END(A, k, v)  (A).prof[k] += (int)(clock64() - v)
#else
#define PROF_BEGIN(v)
#define PROF_END(A, k, v)
#endif

/* HitWork (Solids.h:47-58): up to 4 roots; slot [-1] is addressable */
struct Work {
    double r[6];
    D3     n[6];
    int    count;
    int    inside_inf;
    double eps;
};
#define WR(w, i) (w).r[(i) + 1]
#define WN(w, i) (w).n[(i) + 1]

/* roots and normals are always written before they are read (a slot HitList reads is a root the solver stored or the
   synthetic hit it has just set), so only the scalars are initialised: the arrays live in private memory */
RT_DEV void work_init(Work &w) { w.count = 0; w.inside_inf = 0; w.eps = RT_EPSILON; }

template <class AR> __device__ void arena_compact(AR &A);
template <class AR> RT_DEV void list_push(AR &A, int start, int count, int tag)
{
    A.l_start[A.nlist] = (uint16_t)start; A.l_count[A.nlist] = (uint16_t)count; A.l_tag[A.nlist] = tag;
    A.t_start = start; A.t_count = count; A.t_tag = tag;
    A.nlist++;
}
template <class AR> RT_DEV void list_reload_top(AR &A)
{
    if (A.nlist > 0) { A.t_start = A.l_start[A.nlist - 1]; A.t_count = A.l_count[A.nlist - 1]; A.t_tag = A.l_tag[A.nlist - 1]; }
}
template <class AR> RT_DEV void list_retag_top(AR &A, int tag) { A.l_tag[A.nlist - 1] = tag; A.t_tag = tag; }

/* HitWork::HitList, Solids.cpp:172-227 -> pushes a new list (if not empty) for consumer `tag` */
template <class W, class AR>
__device__ void hit_list(W &w, AR &A, int tag, uint16_t prim, int leaf_affine)
{
    int first, origin_inside;
    for (first = 0; first < w.count; first++)
        if (WR(w, first) >= 0.0) break;
    origin_inside = (first & 1) ^ w.inside_inf;
    bool far_first = (w.count == 0) || (first >= w.count) || (WR(w, first) > w.eps);    /* Q10 */
    if (origin_inside && far_first) {
        --first;
        WR(w, first) = 0.0;
        WN(w, first) = d3(0.0, 0.0, 0.0);
    }
    if ((w.count - first) >= 2 && WR(w, first) == 0.0 && WR(w, first + 1) < w.eps) {
        first += 2;
    } else if ((w.count - first) && WR(w, first) < w.eps) {
        WR(w, first) = 0.0;
        WN(w, first) = d3(0.0, 0.0, 0.0);
    }
    int nout = w.count - first;
    if (nout <= 0) return;
    if (A.top + nout > RT_ARENA_SOFT) arena_compact(A);
    if (A.top + nout > RT_ARENA || A.nlist >= RT_LIST_STACK) { A.overflow |= (A.nlist >= RT_LIST_STACK) ? OVF_LISTS : OVF_ARENA; return; }
    /* materials alternate from the far end: the last hit enters nRayInsideAtInfinity's material */
    int inside = w.inside_inf;
    for (int i = w.count - 1; i >= first; --i) {
        HitRec &h = A.h[A.top + (i - first)];
        h.t = WR(w, i);
        h.n = WN(w, i);
        h.leaf_affine = (uint32_t)(leaf_affine + 1);
        h.mat = inside ? 1 : 0;
        h.surf = (h.t > 0.0) ? 1 : 0;
        h.prim = prim;
        h.ntex = 0; h.pad_ = 0; h.texpack = 0;
        inside = !inside;
    }
    list_push(A, A.top, nout, tag);
    A.top += nout;
}

/* Material::Intersect / Surface::Intersect (Solids.cpp:1466-1519) of a chain's wrappers, innermost first, applied to
   one hit while it is still in registers */
RT_DEV void hit_apply_wraps(const RtChain &ch, HitRec &h)
{
    bool dense = h.mat != 0;            /* a fresh hit enters vacuum (0, density 0) or neutronium (1, density 1e6) */
    for (int w = 0; w < ch.n_wrap; w++) {
        const int wi = ch.wrap[w];
        if (wi >= 0) {
            if (dense) { h.mat = (uint16_t)wi; hit_add_tex(h, ch.wrap_tex[w]); dense = (ch.wrap_dense >> w) & 1; }
        } else if (h.t != 0.0) {
            h.surf = (uint16_t)(~wi); hit_add_tex(h, ch.wrap_tex[w]);
        }
    }
}

/* HitList for solvers that return at most two roots (quadrics, polyhedra, slabs): the same rules as hit_list
   above, written without variable array indices so that the roots and normals stay in registers.
   The list is (optional synthetic t = 0 hit) + (roots >= 0), at most three entries.  With a chain, its wrappers
   are applied before the records are handed over.  Where the records go is the emitter's business: ArenaEmit appends a
   list to the per-lane arena (the interpreter), RegEmit keeps the three records in registers (k_fast, which never
   instantiates an arena). */
#ifdef RT_EXP_NONORMAL      /* experiment (wrong pictures): what would hit records without normals be worth? */
#define HSTORE(dst, src) do { HitRec &d_ = (dst); d_.t = (src).t; d_.leaf_affine = (src).leaf_affine; d_.surf = (src).surf; d_.mat = (src).mat; d_.prim = (src).prim; d_.ntex = (src).ntex; d_.pad_ = 0; d_.texpack = (src).texpack; } while (0)
#else
#define HSTORE(dst, src) (dst) = (src)
#endif
template <class AR>
struct ArenaEmit {
    AR &A; int tag;
    RT_DEV bool begin(int L)
    {
        if (A.top + L > RT_ARENA_SOFT) arena_compact(A);
        if (A.top + L > RT_ARENA || A.nlist >= RT_LIST_STACK) { A.overflow |= (A.nlist >= RT_LIST_STACK) ? OVF_LISTS : OVF_ARENA; return false; }
        return true;
    }
    RT_DEV void put(int i, const HitRec &h) { HSTORE(A.h[A.top + i], h); }
    RT_DEV void end(int L) { if (L > 0) { list_push(A, A.top, L, tag); A.top += L; } }      /* an empty list is never pushed */
};
struct RegEmit {
    HitRec h0, h1, h2; int L;
    const double *minv_own;     /* the chain's minv when it alone transforms the hits' normals (RT_CHAIN_OWN_NORMAL), else null */
    RT_DEV const double *own_minv() const { return minv_own; }
    RT_DEV bool begin(int) { return true; }
    RT_DEV void put(int i, const HitRec &h) { if (i == 0) h0 = h; else if (i == 1) h1 = h; else h2 = h; }
    RT_DEV void end(int n) { L = n; }
    /* field by field and BY VALUE: a conditional between lvalues selects an address and loads through it, which pins
       all three records in private memory */
    template <class T> static RT_DEV T sel3(int i, T a, T b, T c) { return i == 0 ? a : (i == 1 ? b : c); }
#define RT_SEL3(f) sel3(i, h0.f, h1.f, h2.f)
    RT_DEV HitRec get(int i) const
    {
        HitRec r;
        r.t = RT_SEL3(t); r.n.x = RT_SEL3(n.x); r.n.y = RT_SEL3(n.y); r.n.z = RT_SEL3(n.z);
        r.leaf_affine = RT_SEL3(leaf_affine); r.surf = RT_SEL3(surf); r.mat = RT_SEL3(mat); r.prim = RT_SEL3(prim);
        r.ntex = RT_SEL3(ntex); r.pad_ = 0; r.texpack = RT_SEL3(texpack);
        return r;
    }
#undef RT_SEL3
};
template <class Emit>
RT_DEV void hit_list2_core(Work &w, Emit &out, uint16_t prim, int leaf_affine, const RtChain *ch)
{
    const int count = w.count;
    const double r0 = WR(w, 0), r1 = WR(w, 1);
    int first = 0;
    if (count > 0 && !(r0 >= 0.0)) first = (count > 1 && !(r1 >= 0.0)) ? 2 : 1;
    const int m = count - first;                                /* roots kept: 0, 1 or 2 */
    const int origin_inside = (first & 1) ^ w.inside_inf;
    const double ta = (first == 0) ? r0 : r1;                   /* first kept root (m >= 1) */
    const D3 na = (first == 0) ? WN(w, 0) : WN(w, 1);
    const bool far_first = (count == 0) || (m == 0) || (ta > w.eps);        /* Q10 */
    const bool prepend = origin_inside && far_first;
    double t0, t1, t2 = r1;
    D3 n0, n1, n2 = WN(w, 1);
    int L;
    if (prepend) { t0 = 0.0; n0 = d3(0.0, 0.0, 0.0); t1 = ta; n1 = na; L = m + 1; }
    else { t0 = ta; n0 = na; t1 = r1; n1 = WN(w, 1); L = m; }
    if (L >= 2 && t0 == 0.0 && t1 < w.eps) { t0 = t2; n0 = n2; L -= 2; }
    else if (L >= 1 && t0 < w.eps) { t0 = 0.0; n0 = d3(0.0, 0.0, 0.0); }
    if (L <= 0) { out.end(0); return; }
    if (!out.begin(L)) return;
    /* materials alternate from the far end: the last hit enters nRayInsideAtInfinity's material */
    const int last_in = w.inside_inf ? 1 : 0;
    HitRec h;
    h.leaf_affine = (uint32_t)(leaf_affine + 1); h.prim = prim; h.pad_ = 0;
    {
        h.t = t0; h.n = n0; h.ntex = 0; h.texpack = 0;
        h.mat = (uint16_t)(((L - 1) & 1) ? !last_in : last_in);
        h.surf = (t0 > 0.0) ? 1 : 0;
        if (ch) hit_apply_wraps(*ch, h);
        out.put(0, h);
    }
    if (L >= 2) {
        h.t = t1; h.n = n1; h.ntex = 0; h.texpack = 0;
        h.mat = (uint16_t)(((L - 2) & 1) ? !last_in : last_in);
        h.surf = (t1 > 0.0) ? 1 : 0;
        if (ch) hit_apply_wraps(*ch, h);
        out.put(1, h);
    }
    if (L >= 3) {
        h.t = t2; h.n = n2; h.ntex = 0; h.texpack = 0;
        h.mat = (uint16_t)last_in;
        h.surf = (t2 > 0.0) ? 1 : 0;
        if (ch) hit_apply_wraps(*ch, h);
        out.put(2, h);
    }
    out.end(L);
}
template <class AR>
__device__ void hit_list2(Work &w, AR &A, int tag, uint16_t prim, int leaf_affine, const RtChain *ch = 0)
{
    ArenaEmit<AR> out = { A, tag };
    hit_list2_core(w, out, prim, leaf_affine, ch);
}

/* ClipPlane, Solids.cpp:232-273 */
RT_DEV int clip_plane(const DRay &ray, const D3 &nrm, Work &w, int do_clip, double pn)
{
    double B, C, t = 0.0;
    if (do_clip && w.count == 0) return 0;
    B = dot(nrm, ray.dir);
    C = dot(nrm, ray.org) - pn;
    if (B != 0.0) t = -C / B;
    if (!do_clip) {
        w.count = 1;
        WR(w, 0) = 0.0;
        w.inside_inf = 1;
    }
    w.inside_inf &= (B < 0.0) ? 1 : 0;
    if (B < 0.0) {
        if (w.count == 2 && WR(w, 1) <= t) {
            w.count = 0;
        } else if (WR(w, 0) < t) {
            WR(w, 0) = t;
            WN(w, 0) = nrm;
        }
    } else if (B > 0.0) {
        if (t <= WR(w, 0)) {
            w.count = 0;
        } else if (w.count == 1 || t < WR(w, 1)) {
            w.count = 2;
            WR(w, 1) = t;
            WN(w, 1) = nrm;
        }
    } else if (C > 0.0)
        w.count = 0;
    return w.count;
}

/* Extent::TestIntersection, Solids.cpp:1533-1572 (reference semantics, used by BoundingBox nodes and the torus) */
RT_DEV int extent_test(const double *lo, const double *hi, const DRay &ray)
{
    double t0 = 0.0, t1 = RT_INFINITY, tmin = 0.0, tmax = 0.0;
    const double o[3] = { ray.org.x, ray.org.y, ray.org.z }, dd[3] = { ray.dir.x, ray.dir.y, ray.dir.z };
    for (int i = 0; i < 3; i++) {
        double B = dd[i];
        double cmin = +lo[i] - o[i];
        double cmax = -hi[i] + o[i];
        if (B != 0.0) { tmin = +cmin / B; tmax = -cmax / B; }
        if (B < 0.0) {
            if (t1 < tmax || cmin > 0.0 || t0 >= tmin) return 0;
            if (t0 < tmax) t0 = tmax;
            if (t1 > tmin) t1 = tmin;
        } else if (B > 0.0) {
            if (t1 < tmin || cmax > 0.0 || t0 >= tmax) return 0;
            if (t0 < tmin) t0 = tmin;
            if (t1 > tmax) t1 = tmax;
        } else if (cmin > 0.0 || cmax > 0.0)
            return 0;
    }
    return 1;
}

/* this build's culling test: does the world ray, t in [0, tmax], touch the (padded, conservative) box?
   Not part of the reference's arithmetic -- any conservative test gives the same hit lists -- so it may use
   fused multiply-adds (the reference path is compiled with contraction off): per bound t = fma(b, inv, -o*inv).
   Rounding: -o*inv carries |o*inv| * 2^-53, the fma another half ulp of t; `slack` covers both. */
struct CullRay { double o[3], inv[3], noi[3], tmax, slack; int zmask, neg; };
/* reciprocal good to a few ulps (hardware estimate + two Newton steps): culling only, never the reference's arithmetic */
RT_DEV double cull_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return r;
}
RT_DEV CullRay cull_ray(const DRay &r, double tmax)
{
    CullRay c;
    c.o[0] = r.org.x; c.o[1] = r.org.y; c.o[2] = r.org.z;
    const double d[3] = { r.dir.x, r.dir.y, r.dir.z };
    c.zmask = 0; c.neg = 0;
    c.slack = 0.0;
    for (int i = 0; i < 3; i++) {
        /* directions too small to invert safely are treated like zero: the slab test then only asks whether the
           origin lies between the planes, which is conservative for |t * d| far below the box padding */
        if (fabs(d[i]) > 1.0e-280) { c.inv[i] = cull_rcp(d[i]); c.noi[i] = -(c.o[i] * c.inv[i]); c.slack += fabs(c.noi[i]); if (d[i] < 0.0) c.neg |= 1 << i; }
        else { c.inv[i] = 0.0; c.noi[i] = 0.0; c.zmask |= 1 << i; }
    }
    c.slack *= 1.0e-12;
    c.tmax = tmax;
    return c;
}
RT_DEV bool cull_interval_ok(double t0, double t1, const CullRay &c)
{
    return t0 - t1 <= c.slack + (fabs(t0) + fabs(t1)) * 1.0e-12;
}
RT_DEV bool cull_box_hit(const double *lo, const double *hi, const CullRay &c)
{
    double t0 = 0.0, t1 = c.tmax;
    for (int i = 0; i < 3; i++) {
        if (!(c.zmask & (1 << i))) {
            double ta = __builtin_fma(lo[i], c.inv[i], c.noi[i]), tb = __builtin_fma(hi[i], c.inv[i], c.noi[i]);
            t0 = fmax(t0, fmin(ta, tb)); t1 = fmin(t1, fmax(ta, tb));
        } else if (c.o[i] < lo[i] || c.o[i] > hi[i])
            return false;
    }
    return cull_interval_ok(t0, t1, c);
}

/* ================================================================== primitives */
RT_DEV void quadric_roots(const double P, const double Q, const double R, const int nclip, const DRay &ray, Work &w)      /* Solids.cpp:325-358 */
{
    work_init(w);
    const D3 &d = ray.dir, &o = ray.org;
    double a = d.x * d.x + d.y * d.y + d.z * d.z * P;
    double b = 2.0 * (d.x * o.x + d.y * o.y + d.z * o.z * P) + d.z * Q;
    double c = o.x * o.x + o.y * o.y + o.z * o.z * P + o.z * Q + R;
    w.count = solve_quadratic(&WR(w, 0), a, b, c);
    if (w.count > 0) { D3 p = ray_at(ray, WR(w, 0)); WN(w, 0) = d3(p.x, p.y, p.z * P + 0.5 * Q); }
    if (w.count > 1) { D3 p = ray_at(ray, WR(w, 1)); WN(w, 1) = d3(p.x, p.y, p.z * P + 0.5 * Q); }
    if (nclip == 1) {
        clip_plane(ray, d3(0.0, 0.0, 1.0), w, 1, 1.0);
        clip_plane(ray, d3(0.0, 0.0, -1.0), w, 1, 1.0);
    }
    if (nclip == -1) { a = -a; b = -b; c = -c; }
    w.inside_inf = (a < 0.0) ? 1 : 0;
}
RT_DEV void poly_roots(const double *pl, const int nsides, const DRay &ray, Work &w)   /* Solids.cpp:769-781 */
{
    work_init(w);
    int n = clip_plane(ray, d3(pl[0], pl[1], pl[2]), w, 0, 1.0);
    for (int i = 1; n && i < nsides; i++)
        n = clip_plane(ray, d3(pl[3 * i], pl[3 * i + 1], pl[3 * i + 2]), w, 1, 1.0);
}
RT_DEV void slab_roots(const DRay &ray, Work &w)        /* Solids.cpp:859-869 */
{
    work_init(w);
    int n = clip_plane(ray, d3(0.0, 0.0, 1.0), w, 0, 1.0);
    if (n) n = clip_plane(ray, d3(0.0, 0.0, -1.0), w, 1, 1.0);
}
template <class AR>
__device__ void quadric_hits(const RtPrim &q, uint16_t prim, const DRay &ray, AR &A, int tag, const RtChain *ch = 0)
{
    Work w;
    quadric_roots(q.p[0], q.p[1], q.p[2], q.i0, ray, w);
    hit_list2(w, A, tag, prim, q.leaf_affine, ch);
}
template <class AR>
__device__ void poly_hits(const RtPrim &p, uint16_t prim, const double *planes, const DRay &ray, AR &A, int tag, const RtChain *ch = 0)
{
    Work w;
    poly_roots(planes + 3 * (size_t)p.plane0, p.i0, ray, w);
    hit_list2(w, A, tag, prim, p.leaf_affine, ch);
}
template <class AR>
__device__ void slab_hits(const RtPrim &p, uint16_t prim, const DRay &ray, AR &A, int tag, const RtChain *ch = 0)
{
    Work w;
    slab_roots(ray, w);
    hit_list2(w, A, tag, prim, p.leaf_affine, ch);
}

template <class AR>
RT_DEVN void torus_hits(const RtPrim &tp, uint16_t prim, const DRay &ray, AR &A, int tag)      /* Solids.cpp:566-639 */
{
    const double rm = tp.p[0];
    const double lo[3] = { -1.0 - rm, -1.0 - rm, -rm }, hi[3] = { 1.0 + rm, 1.0 + rm, rm };
    if (!extent_test(lo, hi, ray)) return;
    Work w;
    work_init(w);
    double fT = norm(ray.org - d3(0.0, 0.0, 0.0)) - 1.0 - rm - 0.1;
    double fS = norm(ray.dir);
    D3 d = ray.dir, o = ray.org;
    if (fT >= 0.0) {
        d = d / fS;
        o = o + d * fT;
    } else {
        fT = 0.0;
        d = d / fS;
    }
    double r2 = rm * rm;
    double f0 = dot(d, d);
    double f1 = o.x * d.x + o.y * d.y + o.z * d.z;
    double f2 = o.x * o.x + o.y * o.y + o.z * o.z;
    double a = f0 * f0;
    double b = 4.0 * f0 * f1;
    double c = 2.0 * f0 * (f2 - r2 - 1.0) + 4.0 * f1 * f1 + 4.0 * d.z * d.z;
    double dd = 4.0 * (f2 - r2 - 1.0) * f1 + 8.0 * o.z * d.z;
    double e = (f2 - r2 - 1) * (f2 - r2 - 1.0) - 4.0 * (r2 - o.z * o.z);
    w.count = solve_quartic(&WR(w, 0), a, b, c, dd, e);
    double db = 4.0 * a, dc = 3.0 * b, ddd = 2.0 * c, de = dd;
    for (int i = 0; i < w.count; i++) {
        double x = WR(w, i);
        x = x - (e + x * (dd + x * (c + x * (b + x * a)))) / (de + x * (ddd + x * (dc + x * db)));
        x = x - (e + x * (dd + x * (c + x * (b + x * a)))) / (de + x * (ddd + x * (dc + x * db)));
        WR(w, i) = x;
    }
    for (int i = 0; i < w.count; i++) {
        double x = o.x + d.x * WR(w, i), y = o.y + d.y * WR(w, i), z = o.z + d.z * WR(w, i);
        double ax = sqrt(x * x + y * y);
        WN(w, i) = d3(x - x / ax, y - y / ax, z);
    }
    if (fT >= 0.0) {
        for (int i = 0; i < w.count; i++)
            WR(w, i) = (WR(w, i) + fT) / fS;
    }
    w.inside_inf = 0;
    w.eps *= 10.0;
    hit_list(w, A, tag, prim, tp.leaf_affine);
}

/* what a workgroup of a tracing kernel keeps in LDS (scene_lds_load), and where the pre-walked mesh crossings are */
struct SceneLds { const double *bvh; const RtChain *chains; uint16_t *stack; int *mstack; int n_nodes, n_chains, stride, groups /* RT_LDS_GROUP_* staged in LDS (scene_lds_load with SL) */;
                  const int *mx_n; const double *mx_t; const int *mx_q; int mxcap; };     /* the pre-walked mesh crossings (RtPipe::mx_*), null = none */

/* ------------------------------------------------------------------ triangle meshes (policy Q14)
 * A closed mesh is a Solid whose boundary crossings are the ray's triangle hits.  Per triangle: the Moeller-Trumbore
 * test in the form of ObjectFormat.cpp:79-102, with its two slips put right (the second barycentric test is
 * u + v > 1, and t = f * Dot(e2, q)) and its parallel test made scale free (|cos| < EPSILON instead of |a| < EPSILON,
 * which would delete every triangle of a fine mesh).  Crossings ahead of the origin (t >= 0) are gathered through the
 * mesh's BVH, the RT_MESH_MAXHITS nearest kept (MAXHITS of Solids.h:20; more is not an error here, the oracle keeps
 * the same ones) in (t, triangle) order, and handed to HitList as any solver's roots
 * are; an odd number of them means the origin is inside, which HitList is told by one root behind the origin. */
struct MeshWork {
    double r[RT_MESH_MAXHITS + 3];
    D3     n[RT_MESH_MAXHITS + 3];
    int    count;
    int    inside_inf;
    double eps;
};
RT_DEV bool mesh_triangle(const RtMeshTri &T, const DRay &ray, double &t)
{
    const D3 e1 = d3(T.e1[0], T.e1[1], T.e1[2]), e2 = d3(T.e2[0], T.e2[1], T.e2[2]);
    const D3 h = cross(ray.dir, e2);
    const double a = dot(e1, h);
    const D3 nn = cross(e1, e2);
    if (a * a < (RT_EPSILON * RT_EPSILON) * (dot(nn, nn) * dot(ray.dir, ray.dir))) return false;
    const double f = 1.0 / a;
    const D3 s = ray.org - d3(T.v0[0], T.v0[1], T.v0[2]);
    const double u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return false;
    const D3 q = cross(s, e1);
    const double v = f * dot(ray.dir, q);
    if (v < 0.0 || u + v > 1.0) return false;
    t = f * dot(e2, q);
    return t >= 0.0;
}
#ifdef RT_MESH_COUNT        /* instrumentation build (RT_EXTRA_DEFS=-DRT_MESH_COUNT): [0] rays, [1] node visits, [2] leaf visits, [3] triangle tests, [4] wave-iterations of the node loop, [5] of the leaf loop */
__device__ unsigned long long g_mesh_count[32];     /* [8..15] FULL, [16..23] LITE interpreter loop: +0 wave iterations, +1 distinct opcodes summed over them, +2 active lanes summed over them, +3 rays, +4 lane iterations */
/* the loop's counters are kept by one workgroup in 64: same-address atomics of every wave in every round would be what the run measures */
#define RT_COUNT_SAMPLED ((blockIdx.x & 63) == 0)
__device__ unsigned long long g_op_time[32], g_op_rounds[32], g_op_lanes[32];      /* SCHED interpreter: per opcode, clock ticks / rounds / lanes of the rounds it led */
#define MESH_COUNT(k, v) atomicAdd(&g_mesh_count[k], (unsigned long long)(v))
#define MESH_WAVE_COUNT(k) do { if ((__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&g_mesh_count[k], 1ull); } while (0)
#else
#define MESH_COUNT(k, v)
#define MESH_WAVE_COUNT(k)
#endif
struct MeshRay { double o[3], inv[3]; int par[3]; };      /* par: the direction component is zero */
RT_DEV bool mesh_slot_hit(const RtMeshNode &nd, int k, const MeshRay &mr)
{
    double t0 = 0.0, t1 = RT_INFINITY;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double lo = (double)nd.org[a] + (double)nd.qlo[a][k] * (double)nd.scale[a], hi = (double)nd.org[a] + (double)nd.qhi[a][k] * (double)nd.scale[a];
        if (!mr.par[a]) {
            double ta = (lo - mr.o[a]) * mr.inv[a], tb = (hi - mr.o[a]) * mr.inv[a];
            const double tn = fmin(ta, tb), tf = fmax(ta, tb);
            /* one part in 1e9 of slack on top of the padded boxes: this is not the arithmetic of the triangle test */
            t0 = fmax(t0, tn - fabs(tn) * 1e-9);
            t1 = fmin(t1, tf + fabs(tf) * 1e-9);
        } else ok = ok && !(mr.o[a] < lo || mr.o[a] > hi);
    }
    return ok && t0 <= t1;
}
/* The walk.  Two phases that never interleave within a wave: (1) pop a node, test its four boxes, push the inner children
   that are hit and QUEUE the leaf slots that are hit; (2) test the queued triangles, one per lane and iteration.  With the
   triangle tests inside the node loop a wave ran them once per node visit for its busiest lane: measured on the
   10 M-triangle scene, 183 triangle iterations per 64-ray wave for 10.3 tests per ray -- 6 % of the lanes busy.
   Node stack and leaf queue share one column of `cap` LDS words per lane (entry i at [i * stride]: no bank conflicts; in
   private memory every push and pop is a round trip through the vector memory path): the stack grows up from 0, the queue
   down from cap - 1 (one word per leaf slot: first triangle | (count - 1) << 27); a lane whose column fills up tests its
   queued triangles early.  Every crossing goes to the sink, unsorted.  Returns false when the stack alone outgrows the column. */
template <class Sink>
RT_DEV void mesh_test_queued(const RtMeshTri *tris, const DRay &ray, int *stack, const int stride, const int cap, int &lq, Sink &K)
{
#define MSTK(i) stack[(i) * stride]
    int q = 0, left = 0;
    for (;;) {
        if (left == 0) {
            if (lq >= cap) break;
            const int e = MSTK(lq); lq++;
            q = e & 0x07FFFFFF; left = (e >> 27) + 1;
        }
        MESH_COUNT(3, 1);
        MESH_WAVE_COUNT(5);
        double t;
        const bool hit = mesh_triangle(tris[q], ray, t);
        if (hit) { MESH_COUNT(6, 1); K.add(t, q, tris); }
        q++; left--;
    }
#undef MSTK
}
template <class Sink>
RT_DEV bool mesh_walk(const RtMeshNode *nodes, const RtMeshTri *tris, const DRay &ray, int *stack, const int stride, const int cap, Sink &K)
{
#define MSTK(i) stack[(i) * stride]
    MeshRay mr;
    mr.o[0] = ray.org.x; mr.o[1] = ray.org.y; mr.o[2] = ray.org.z;
    {
        const double d[3] = { ray.dir.x, ray.dir.y, ray.dir.z };
#pragma unroll
        for (int a = 0; a < 3; a++) { mr.par[a] = d[a] == 0.0; mr.inv[a] = mr.par[a] ? 0.0 : 1.0 / d[a]; }
    }
    int sp = 0, lq = cap;
    bool ok = true;
    MSTK(sp) = 0; sp++;
    MESH_COUNT(0, 1);
    while (sp > 0) {
        const RtMeshNode &nd = nodes[MSTK(--sp)];
        MESH_COUNT(1, 1);
        MESH_WAVE_COUNT(4);
        bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; k++) hit[k] = mesh_slot_hit(nd, k, mr);
        if (sp + 4 > lq) {                      /* the column is full: make room by testing what is queued */
            mesh_test_queued(tris, ray, stack, stride, cap, lq, K);
            if (sp + 4 > lq) { ok = false; break; }
        }
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const int c = nd.child[k];
            if (!hit[k]) continue;
            if (c >= 0) { MSTK(sp) = c; sp++; }
            else if (c != RT_MESH_EMPTY) { --lq; MSTK(lq) = ~c; }
        }
    }
    mesh_test_queued(tris, ray, stack, stride, cap, lq, K);
    return ok;
#undef MSTK
}
/* keeps the RT_MESH_MAXHITS nearest crossings in (t, triangle id) order of preference, in private arrays */
struct MeshLocalSink {
    double ht[RT_MESH_MAXHITS];
    int    hq[RT_MESH_MAXHITS];         /* position in tris[] */
    int    nh;
    RT_DEV void add(double t, int q, const RtMeshTri *tris)
    {
        if (nh < RT_MESH_MAXHITS) { ht[nh] = t; hq[nh] = q; nh++; }
        else {          /* rare (a ray skimming a faceted saddle): keep the RT_MESH_MAXHITS nearest, as the oracle does */
            int m = 0;
            for (int i = 1; i < RT_MESH_MAXHITS; i++)
                if (ht[i] > ht[m] || (ht[i] == ht[m] && tris[hq[i]].id > tris[hq[m]].id)) m = i;
            if (t < ht[m] || (t == ht[m] && tris[q].id < tris[hq[m]].id)) { ht[m] = t; hq[m] = q; }
        }
    }
};
/* the same rule with the crossings in global memory, entry i of a ray at [i * cap] (k_mesh_cross): a store per crossing
   and no private array; the replacement of a 17th crossing reads the stored ones back */
struct MeshGlobalSink {
    double *ht; int *hq; size_t cap; int nh;
    RT_DEV void add(double t, int q, const RtMeshTri *tris)
    {
        if (nh < RT_MESH_MAXHITS) { ht[nh * cap] = t; hq[nh * cap] = q; nh++; }
        else {
            int m = 0;
            double tm = ht[0]; int qm = hq[0];
            for (int i = 1; i < RT_MESH_MAXHITS; i++) {
                const double ti = ht[i * cap]; const int qi = hq[i * cap];
                if (ti > tm || (ti == tm && tris[qi].id > tris[qm].id)) { m = i; tm = ti; qm = qi; }
            }
            if (t < tm || (t == tm && tris[q].id < tris[qm].id)) { ht[m * cap] = t; hq[m * cap] = q; }
        }
    }
};
/* WALK = false (the LITE kernels of a scene whose rays are all pre-walked): a ray without pre-walked crossings (its walk ran
   out of stack) is handed to the FULL kernel, which walks it in line */
template <bool WALK, class AR>
RT_DEVN void mesh_hits(const RtDevScene &S, const RtPrim &mp, uint16_t prim, const DRay &ray, AR &A, int tag, const SceneLds &L)
{
    const RtMeshInfo mi = S.meshes[mp.plane0];
    const RtMeshTri *tris = S.mesh_tris + mi.tri0;
    MeshLocalSink K;
    K.nh = 0;
    int pre = -1;
    if (A.mesh_slot >= 0 && L.mx_n) pre = L.mx_n[A.mesh_slot];
    if (pre == 0) return;
    if (pre == 1 || pre == 2) {
        /* What nearly every ray through a closed mesh brings: one crossing (the origin is inside or on the surface) or two.  Taken in
           registers -- no private arrays, no sort, and the two-root form of HitList (hit_list2: the same rules, written for at most
           two roots): the general path below wrote 580 bytes of private memory per ray for them.  Order (t, triangle id) as below;
           an odd count puts a root behind the origin in front (Solids.h:47-58's slot [-1]), as below. */
        double ta = L.mx_t[A.mesh_slot];
        int qa = L.mx_q[A.mesh_slot];
        Work w;
        work_init(w);
        if (pre == 1) {
            const RtMeshTri &T = tris[qa];
            WR(w, 0) = -1.0; WN(w, 0) = d3(0.0, 0.0, 0.0);
            WR(w, 1) = ta; WN(w, 1) = cross(d3(T.e1[0], T.e1[1], T.e1[2]), d3(T.e2[0], T.e2[1], T.e2[2]));
        } else {
            double tb = L.mx_t[(size_t)L.mxcap + A.mesh_slot];
            int qb = L.mx_q[(size_t)L.mxcap + A.mesh_slot];
            if (tb < ta || (tb == ta && tris[qb].id < tris[qa].id)) { const double t_ = ta; ta = tb; tb = t_; const int q_ = qa; qa = qb; qb = q_; }
            const RtMeshTri &Ta = tris[qa], &Tb = tris[qb];
            WR(w, 0) = ta; WN(w, 0) = cross(d3(Ta.e1[0], Ta.e1[1], Ta.e1[2]), d3(Ta.e2[0], Ta.e2[1], Ta.e2[2]));
            WR(w, 1) = tb; WN(w, 1) = cross(d3(Tb.e1[0], Tb.e1[1], Tb.e1[2]), d3(Tb.e2[0], Tb.e2[1], Tb.e2[2]));
        }
        w.count = 2;
        hit_list2(w, A, tag, prim, mp.leaf_affine);
        return;
    }
    if (pre >= 0) {                     /* found by k_mesh_cross: the same walk, run in a kernel of its own */
        for (int i = 0; i < pre; i++) { K.ht[i] = L.mx_t[(size_t)i * L.mxcap + A.mesh_slot]; K.hq[i] = L.mx_q[(size_t)i * L.mxcap + A.mesh_slot]; }
        K.nh = pre;
    } else if constexpr (!WALK) { A.bail = 1; return; }
    else if (!mesh_walk(S.mesh_nodes + mi.node0, tris, ray, L.mstack + threadIdx.x, L.stride, RT_MESH_STACK, K)) {
        A.overflow |= OVF_MESH_STACK;
    }
    double *const ht = K.ht;
    int *const hq = K.hq;
    const int nh = K.nh;
    /* insertion sort, order (t, triangle id) */
    for (int i = 1; i < nh; i++) {
        const double t = ht[i];
        const int q = hq[i], id = tris[q].id;
        int j = i;
        for (; j > 0 && (t < ht[j - 1] || (t == ht[j - 1] && id < tris[hq[j - 1]].id)); --j) { ht[j] = ht[j - 1]; hq[j] = hq[j - 1]; }
        ht[j] = t; hq[j] = q;
    }
    if (nh == 0) return;
    MeshWork w;
    w.inside_inf = 0; w.eps = RT_EPSILON;
    const int behind = nh & 1;                  /* odd number of crossings ahead: the origin is inside */
    if (behind) { WR(w, 0) = -1.0; WN(w, 0) = d3(0.0, 0.0, 0.0); }
    for (int i = 0; i < nh; i++) {
        const RtMeshTri &T = tris[hq[i]];
        WR(w, behind + i) = ht[i];
        WN(w, behind + i) = cross(d3(T.e1[0], T.e1[1], T.e1[2]), d3(T.e2[0], T.e2[1], T.e2[2]));
    }
    w.count = nh + behind;
    hit_list(w, A, tag, prim, mp.leaf_affine);
}

/* ================================================================== CSG merge (Solids.cpp:987-1058) */
enum { MERGE_MAX = 0, MERGE_MIN = 1, MERGE_DIFF = 2 };
RT_DEV uint16_t mat_op(int op, uint16_t l, uint16_t r, const RtMaterialDev *mats)
{
    double dl = mats[l].density, dr = mats[r].density;
    if (op == MERGE_MAX) return dl > dr ? l : r;
    if (op == MERGE_MIN) return dl < dr ? l : r;
    return dr <= 0.0 ? l : (uint16_t)0;
}

/* Lists sit in the arena in stack order with increasing start offsets; a merge writes its result above both
   operands and leaves it there (no copy back), so dead records can lie between live lists.  When space runs out
   the live lists are packed down. */
template <class AR>
__device__ void arena_compact(AR &A)
{
    int top = 0;
    for (int l = 0; l < A.nlist; l++) {
        const int s0 = A.l_start[l], c = A.l_count[l];
        if (s0 != top) for (int i = 0; i < c; i++) A.h[top + i] = A.h[s0 + i];
        A.l_start[l] = (uint16_t)top;
        if (l == A.nlist - 1) A.t_start = top;
        top += c;
    }
    A.top = top;
}

/* merge the two top lists (left below right) into one that takes the left list's place on the list stack */
template <class AR>
__device__ void merge_top(AR &A, int op, const RtMaterialDev *mats, int tag)
{
    int li = A.nlist - 2;
    int lc = A.l_count[li], rc = A.t_count;
    if (A.top + lc + rc > RT_ARENA_SOFT) arena_compact(A);
    int ls = A.l_start[li], rs = A.t_start;
    int out = A.top, nout = 0;
    if (out + lc + rc > RT_ARENA) { A.overflow |= OVF_ARENA; A.nlist -= 1; A.top = rs; list_reload_top(A); list_retag_top(A, tag); return; }
    int il = 0, ir = 0;
    uint16_t ml = 0, mr = 0;
    double prev = mats[0].density;
    while (il < lc || ir < rc) {
        bool take_left = (ir >= rc) || (il < lc && A.h[ls + il].t <= A.h[rs + ir].t);
        if (take_left) {
            if (ir < rc && A.h[ls + il].t == A.h[rs + ir].t) { mr = A.h[rs + ir].mat; ir++; }
            ml = A.h[ls + il].mat;
            uint16_t m = mat_op(op, ml, mr, mats);
            double dm = mats[m].density;
            if (dm != prev) { prev = dm; A.h[out + nout] = A.h[ls + il]; A.h[out + nout].mat = m; nout++; }
            il++;
        } else {
            mr = A.h[rs + ir].mat;
            uint16_t m = mat_op(op, ml, mr, mats);
            double dm = mats[m].density;
            if (dm != prev) { prev = dm; A.h[out + nout] = A.h[rs + ir]; A.h[out + nout].mat = m; nout++; }
            ir++;
        }
    }
    if (nout == 1 && A.h[out].t == 0.0 && mats[A.h[out].mat].density == 0.0) nout = 0;     /* Solids.cpp:1053-1056 */
    const int ridx = A.l_idx[A.nlist - 1];
    A.nlist -= 2;
    A.top = ls;
    if (nout > 0) {
        /* the merged entry keeps the left operand's l_gap and takes the right operand's l_idx */
        A.l_idx[A.nlist] = (uint16_t)ridx;
        list_push(A, out, nout, tag);
        A.top = out + nout;
    } else
        list_reload_top(A);
}

template <class AR> RT_DEV bool top_is(const AR &A, int tag) { return A.nlist > 0 && A.t_tag == tag; }
template <class AR> RT_DEV bool second_is(const AR &A, int tag) { return A.nlist > 1 && A.l_tag[A.nlist - 2] == tag; }
template <class AR> RT_DEV void drop_top(AR &A) { A.nlist--; A.top = A.t_start; list_reload_top(A); }

/* ================================================================== the per-ray stack machine
 * Evaluates psModel->Intersect(ray) (RayTracer2020.h:72): returns the hit list as [start, start+count)
 * in the arena.  tmax bounds the culling tests only (shadow rays stop at t = 1, Shade.cpp:328). */
/* the innermost union region being folded lives in registers; enclosing ones are saved in private memory */
struct RegionCur { int ncand, pos, base_lists, nl_before, out_tag, child_pc0, lca0, nchild, uid, end_pc, cur_c, top_idx, top_gap; };
/* the interpreter's per-ray arrays: the rays of the enclosing affine frames, the saved fold states and candidate lists of the
   enclosing regions -- private memory, or the lane's part of the pool (ArenaStore<true>) */
template <bool POOL> struct WalkStore {
    DRay rs[RT_MAX_AFFINE_DEPTH + 1];
    RegionCur saved[RT_MAX_REGION_DEPTH];
    uint16_t cands[RT_MAX_REGION_DEPTH][RT_MAX_CAND];
    template <class AR> RT_DEV void bind(AR &) {}
};
static_assert(sizeof(HitRec) == 48, "RT_POOL_REC is written for this size");
template <> struct WalkStore<true> : WalkStore<false> {};

/* min of lca[a .. b-1], a < b (RtRegion sparse table) */
RT_DEV int region_gap(const RtDevScene &S, const RegionCur &rg, int a, int b)
{
    int len = b - a, l = 31 - __clz(len);
    const int32_t *row = S.region_lca + rg.lca0 + (size_t)l * rg.nchild;
    int x = row[a], y = row[b - (1 << l)];
    return x < y ? x : y;
}

/* workgroup-shared staging in dynamic LDS: the first n_nodes nodes of the scene's BVH, the scene's chain records
   (persistent kernels, when they all fit), then the per-lane traversal stacks (RT_BVH_STACK x blockDim.x x 2 B).
   The host sizes the allocation (rt_scene_upload: lds_small / lds_visual / lds_shadow). */
#define RT_LDS_STACK(L, sp, lane) (L).stack[(sp) * (L).stride + (lane)]
/* every thread of the workgroup must call this before intersect_scene */
/* 8-byte words of LDS the small scene tables take when staged (same arithmetic on the host: rt_scene_upload) */
/* Two groups, staged independently (RT_LDS_GROUP_* bits travel in the upper byte of the kernels' n_chains_lds argument):
   CONTROL = what the interpreter's dependent lookups go through (program, regions, their child and LCA tables): small,
   and every candidate of a union region costs two of them in a row; DATA = boxes, affines, materials, surfaces. */
#define RT_LDS_GROUP_CONTROL 1
#define RT_LDS_GROUP_DATA    2
RT_HD int scene_table_words_control(const RtDevScene &S)
{
    return S.n_prog * (int)(sizeof(RtInstr) / 8) + S.n_regions * (int)(sizeof(RtRegion) / 8) + (S.n_child_pc + 1) / 2 + (S.n_lca + 1) / 2;
}
RT_HD int scene_table_words_data(const RtDevScene &S)
{
    return S.n_boxes * (int)(sizeof(RtBox) / 8) + S.n_affines * (int)(sizeof(RtAffine) / 8) +
           S.n_mats * (int)(sizeof(RtMaterialDev) / 8) + S.n_surfs * (int)(sizeof(RtSurfaceDev) / 8);
}
template <class T> RT_DEV const T *lds_stage(double *&cur, const T *src, int n)
{
    static_assert(sizeof(T) % 4 == 0, "tables are copied as 32-bit words");
    const int words = n * (int)(sizeof(T) / 4);
    uint32_t *d = (uint32_t *)cur;
    const uint32_t *s = (const uint32_t *)src;
    for (int k = threadIdx.x; k < words; k += blockDim.x) d[k] = s[k];
    cur += (words + 1) / 2;
    return (const T *)d;
}
/* SL (optional): a copy of the scene description whose small tables -- program, boxes, affines, materials, surfaces,
   regions and their child / LCA tables -- point at LDS copies.  Their lookups are per-lane gathers (every lane is at
   its own instruction, material, box ...): in LDS they cost bank cycles, through the vector L1 one cache line each. */
RT_DEV SceneLds scene_lds_load(const RtDevScene &S, int n_chains_packed, RtDevScene *SL = 0)
{
    extern __shared__ double rt_dyn_lds[];
    SceneLds L;
    L.mx_n = 0; L.mx_t = 0; L.mx_q = 0; L.mxcap = 0;
    const int groups = (n_chains_packed >> 24) & 3;         /* bit 28 belongs to k_visual (primary level) */
    L.groups = SL ? groups : 0;
    L.n_nodes = S.n_bvh_lds;
    L.n_chains = n_chains_packed & 0xFFFFFF;
    L.stride = blockDim.x;
    const int words_bvh = L.n_nodes * (int)(sizeof(RtBvhNode) / 8), words_ch = L.n_chains * (int)(sizeof(RtChain) / 8);
    const double *src = (const double *)S.bvh, *src2 = (const double *)S.chains;
    for (int k = threadIdx.x; k < words_bvh; k += blockDim.x) rt_dyn_lds[k] = src[k];
    for (int k = threadIdx.x; k < words_ch; k += blockDim.x) rt_dyn_lds[words_bvh + k] = src2[k];
    L.bvh = rt_dyn_lds;
    L.chains = (const RtChain *)(rt_dyn_lds + words_bvh);
    double *cur = rt_dyn_lds + words_bvh + words_ch;
    if (SL && (groups & RT_LDS_GROUP_CONTROL)) {
        SL->prog = lds_stage(cur, S.prog, S.n_prog);
        SL->regions = lds_stage(cur, S.regions, S.n_regions);
        SL->region_child_pc = lds_stage(cur, S.region_child_pc, S.n_child_pc);
        SL->region_lca = lds_stage(cur, S.region_lca, S.n_lca);
    }
    if (SL && (groups & RT_LDS_GROUP_DATA)) {
        SL->boxes = lds_stage(cur, S.boxes, S.n_boxes);
        SL->affines = lds_stage(cur, S.affines, S.n_affines);
        SL->mats = lds_stage(cur, S.mats, S.n_mats);
        SL->surfs = lds_stage(cur, S.surfs, S.n_surfs);
    }
    L.stack = (uint16_t *)cur;
    L.mstack = (int *)(L.stack + RT_BVH_STACK * blockDim.x);    /* present only when the launch was sized for it: scenes with meshes, FULL kernels */
    __syncthreads();
    return L;
}

/* one 4-wide node: test the four child boxes together; the near/far plane of every axis is picked by the sign
   of the ray direction, so a bound costs one fma and one min/max */
RT_DEV unsigned bvh4_test(const double *nb, const CullRay &c, int32_t child[4])
{
    double t0[4], t1[4];
    unsigned alive = 0xF;
    for (int k = 0; k < 4; k++) { t0[k] = 0.0; t1[k] = c.tmax; }
    for (int d = 0; d < 3; d++) {
        const double *lo = nb + 4 * d, *hi = nb + 12 + 4 * d;
        if (!(c.zmask & (1 << d))) {
            const double *nr = (c.neg & (1 << d)) ? hi : lo, *fr = (c.neg & (1 << d)) ? lo : hi;
            for (int k = 0; k < 4; k++) {
                t0[k] = fmax(t0[k], __builtin_fma(nr[k], c.inv[d], c.noi[d]));
                t1[k] = fmin(t1[k], __builtin_fma(fr[k], c.inv[d], c.noi[d]));
            }
        } else {
            for (int k = 0; k < 4; k++) if (c.o[d] < lo[k] || c.o[d] > hi[k]) alive &= ~(1u << k);
        }
    }
    const int32_t *ch = (const int32_t *)(nb + 24);
    unsigned hit = 0;
    for (int k = 0; k < 4; k++) {
        child[k] = ch[k];
        if ((alive & (1u << k)) && child[k] != RT_BVH_EMPTY && cull_interval_ok(t0[k], t1[k], c)) hit |= 1u << k;
    }
    return hit;
}

/* candidates of a union region: children with unbounded boxes plus those whose box the ray touches,
   ascending (= the reference's evaluation order) */
__device__ int region_candidates(const RtDevScene &S, const SceneLds &L, const RtRegion &rg, const CullRay &cr, uint16_t *cand, int &overflow)
{
    int n = 0;
    for (int k = 0; k < rg.n_always; k++) { if (n < RT_MAX_CAND) cand[n++] = (uint16_t)S.region_always[rg.always0 + k]; else overflow |= OVF_CAND; }
    if (rg.bvh_root >= 0) {
        const int lane = threadIdx.x;
        int sp = 0;
        int node = rg.bvh_root;
        for (;;) {
            int32_t child[4];
            unsigned hit;
            if (node < L.n_nodes) hit = bvh4_test(L.bvh + (size_t)node * (sizeof(RtBvhNode) / 8), cr, child);
            else hit = bvh4_test((const double *)(S.bvh + node), cr, child);
            for (int k = 3; k >= 0; k--) {
                if (!(hit & (1u << k))) continue;
                if (child[k] < 0) { if (n < RT_MAX_CAND) cand[n++] = (uint16_t)(~child[k]); else overflow |= OVF_CAND; }
                else if (sp < RT_BVH_STACK) RT_LDS_STACK(L, sp++, lane) = (uint16_t)child[k];
                else overflow |= OVF_BVH_STACK;
            }
            if (sp == 0) break;
            node = RT_LDS_STACK(L, --sp, lane);
        }
    }
    for (int i = 1; i < n; i++) {           /* insertion sort: n is small */
        uint16_t v = cand[i];
        int j = i - 1;
        while (j >= 0 && cand[j] > v) { cand[j + 1] = cand[j]; j--; }
        cand[j + 1] = v;
    }
    return n;
}

/* Can psModel->Intersect(ray) be non-empty at all?  false only when the ray misses every conservative box of a
   scene whose root is one union region (or one boxed chain): then the hit list is empty, the radiance along a
   primary ray is black and the irradiance along a shadow ray is unoccluded -- no need to run the interpreter. */
__device__ bool scene_may_hit(const RtDevScene &S, const SceneLds &L, const DRay &wray, double tmax)
{
    if (S.root_cull == 0) return true;
    const CullRay cr = cull_ray(wray, tmax);
    if (S.root_cull == 2) return cull_box_hit(S.boxes[S.root_cull_arg].lo, S.boxes[S.root_cull_arg].hi, cr);
    const RtRegion &rg = S.regions[S.root_cull_arg];
    if (rg.n_always > 0) return true;
    if (rg.bvh_root < 0) return false;
    const int lane = threadIdx.x;
    int sp = 0, node = rg.bvh_root;
    for (;;) {
        int32_t child[4];
        unsigned hit;
        if (node < L.n_nodes) hit = bvh4_test(L.bvh + (size_t)node * (sizeof(RtBvhNode) / 8), cr, child);
        else hit = bvh4_test((const double *)(S.bvh + node), cr, child);
        for (int k = 3; k >= 0; k--) {
            if (!(hit & (1u << k))) continue;
            if (child[k] < 0) return true;
            if (sp < RT_BVH_STACK) RT_LDS_STACK(L, sp++, lane) = (uint16_t)child[k];
            else return true;
        }
        if (sp == 0) return false;
        node = RT_LDS_STACK(L, --sp, lane);
    }
}

/* One entry of a small scene table.  in_lds (the same for all lanes): the workgroup staged the table in LDS (scene_lds_load) and
   the pointer is a generic one into it -- whose low half is the LDS address.  Read through it, the load is an LDS instruction:
   it counts on the LDS counter alone, where the generic (flat) load the compiler has to emit for a pointer that may be either
   counts on both and so waits for every store and load to memory still in flight. */
template <class T> RT_DEV T tab_get(const T *p, const bool in_lds)
{
    T v;
    if (in_lds) __builtin_memcpy(&v, (const __attribute__((address_space(3))) T *)p, sizeof(T));
    else v = *p;
    return v;
}
/* a pointer every lane holds the same value of, moved to scalar registers */
template <class T> RT_DEV T *uniform_ptr(T *p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (T *)(((uint64_t)hi << 32) | lo);
}

/* LITE = true compiles the rare, register-hungry features out (torus quartics here; texture call-outs, propagators
   and spectral metals in the kernels): a ray that needs one sets A.bail and is redone by the FULL instantiation.
   The split keeps the common kernels small enough for several waves per SIMD. */
/* TILES: the caller may pass the candidate list of a primary ray's sampling tile (k_tile_mask).  A template parameter, so
   that kernels for scenes and levels without such lists are compiled without that code: its few extra live registers
   cost the tightly budgeted LITE kernels spills (S1 / S2 ran 6-8 % slower with the code merely present). */
/* MESHPRE (LITE only): the scene's one mesh instruction is served from pre-walked crossings (k_mesh_cross ran over every ray
   of the level before this kernel) instead of handing the ray over */
/* SCHED (FULL kernels of long programs, levels behind the first): the lanes of a wave run in the order of their pc -- see the loop */
template <bool LITE, bool CH_LDS = false, bool TILES = false, bool MESHPRE = false, bool SCHED = false, class AR>
__device__ int intersect_scene(const RtDevScene &S_in, const SceneLds &L_in, const DRay &wray, double tmax, AR &A, int &start,
                               const unsigned short *tile_list = 0, int tile_n = 0, int tile_one = -1, int mesh_slot = -1)
{
    /* In the FULL kernels this is a function of its own, and scene and LDS descriptions arrive by reference: every `S.boxes[..]`
       would first load the table's pointer from the caller's frame -- a second memory access in front of each table access, whose
       wait (flat loads count on both counters) also drains every store the round before left in flight.  The pointers are the same
       for all lanes: they are read once per ray here and kept in scalar registers.  (The LITE kernels inline the function and
       see the values directly.) */
    RtDevScene S_uni;
    SceneLds L_uni;
    if constexpr (!LITE) {
        S_uni.prog = uniform_ptr(S_in.prog); S_uni.boxes = uniform_ptr(S_in.boxes); S_uni.affines = uniform_ptr(S_in.affines);
        S_uni.prims = uniform_ptr(S_in.prims); S_uni.planes = uniform_ptr(S_in.planes); S_uni.mats = uniform_ptr(S_in.mats);
        S_uni.surfs = uniform_ptr(S_in.surfs); S_uni.clips = uniform_ptr(S_in.clips); S_uni.regions = uniform_ptr(S_in.regions);
        S_uni.bvh = uniform_ptr(S_in.bvh); S_uni.chains = uniform_ptr(S_in.chains); S_uni.region_child_pc = uniform_ptr(S_in.region_child_pc);
        S_uni.region_always = uniform_ptr(S_in.region_always); S_uni.region_lca = uniform_ptr(S_in.region_lca);
        S_uni.region_child_boxes = uniform_ptr(S_in.region_child_boxes); S_uni.root_alt_pc = uniform_ptr(S_in.root_alt_pc);
        S_uni.n_prog = __builtin_amdgcn_readfirstlane(S_in.n_prog); S_uni.n_root_alt = __builtin_amdgcn_readfirstlane(S_in.n_root_alt);
        S_uni.root_cull = __builtin_amdgcn_readfirstlane(S_in.root_cull); S_uni.root_cull_arg = __builtin_amdgcn_readfirstlane(S_in.root_cull_arg);
        S_uni.root_child_pc0 = __builtin_amdgcn_readfirstlane(S_in.root_child_pc0); S_uni.n_bvh_lds = __builtin_amdgcn_readfirstlane(S_in.n_bvh_lds);
        L_uni.bvh = uniform_ptr(L_in.bvh); L_uni.chains = uniform_ptr(L_in.chains); L_uni.stack = uniform_ptr(L_in.stack); L_uni.mstack = uniform_ptr(L_in.mstack);
        L_uni.n_nodes = __builtin_amdgcn_readfirstlane(L_in.n_nodes); L_uni.n_chains = __builtin_amdgcn_readfirstlane(L_in.n_chains);
        L_uni.stride = __builtin_amdgcn_readfirstlane(L_in.stride);
        L_uni.groups = __builtin_amdgcn_readfirstlane(L_in.groups);
    }
    const RtDevScene &S = LITE ? S_in : S_uni;      /* (the mesh walk, a function of its own as well, gets the caller's: S_in, L_in) */
    const SceneLds &L = LITE ? L_in : L_uni;
#ifndef RT_TAB_LDS_FULL
#define RT_TAB_LDS_FULL 0
#endif
    const bool tab_lds_on = LITE || RT_TAB_LDS_FULL != 0;
    const bool ctl_lds = tab_lds_on && (L.groups & RT_LDS_GROUP_CONTROL) != 0, data_lds = tab_lds_on && RT_TAB_LDS_FULL != 2 && (L.groups & RT_LDS_GROUP_DATA) != 0;
    WalkStore<AR::pool> wst;
    wst.bind(A);
    auto &rs = wst.rs; auto &saved = wst.saved; auto &cands = wst.cands;
    A.mesh_slot = mesh_slot;
    RegionCur cf = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1 };
    int nrf = 0;
    int depth = 0;
    DRay cur = wray;            /* the ray in the current frame; rs[] holds the enclosing frames' rays (touched only by non-chain affines) */
#ifdef RT_PHASE_PROF
    long long t_cr = clock64();
#endif
    const CullRay cr = cull_ray(wray, tmax);
    A.nlist = 0; A.top = 0; A.overflow = 0; A.bail = 0; A.t_start = 0; A.t_count = 0; A.t_tag = -1;
#ifdef RT_PHASE_PROF
    for (int q = 0; q < 12; q++) A.prof[q] = 0;
#endif
    PROF_BEGIN(t_all);
#ifdef RT_PHASE_PROF
    A.prof[7] = (int)(clock64() - t_cr);
#endif
    int pc = 0;
    RtInstr in;
    bool fetch = true;     /* (SCHED) this lane ran its instruction in the round before */
    const int pc_bits = 31 - __clz((S.n_prog + S.n_root_alt * 2 + 2) | 1);      /* top bit a pc can have (the root's single-chain copies sit behind the program) */
#ifdef RT_MESH_COUNT
    long long cnt_t0 = 0; int cnt_op = 0;
#endif
    if (TILES && S.n_root_alt > 0 && (tile_one >= 0 || (tile_list && tile_n == 1))) {
        /* the sampling tile of this primary ray can touch ONE child of the root region: if that is a bare chain, its
           box is tested here and the chain runs alone, for the root consumer (what the region would have made of it).
           The two table reads do not wait for each other. */
        const int c = tile_one >= 0 ? tile_one : (int)tile_list[0];
        const int apc = S.root_alt_pc[c];
        const RtBox cb = S.region_child_boxes[S.root_child_pc0 + c];
        if (apc >= 0) {
            if (!cull_box_hit(cb.lo, cb.hi, cr)) goto done;
            pc = apc;
        }
    }
#ifdef RT_MESH_COUNT
    if (RT_COUNT_SAMPLED) MESH_COUNT(LITE ? 19 : 11, 1);
#endif
    for (;;) {
#ifdef RT_MESH_COUNT
        {   /* lanes still in the loop this round, running or waiting for their class: against the lanes that run (+2), the share of
               class serialisation; against 64, the share of rays of unequal length */
            const unsigned long long alive = __ballot(1);
            if (RT_COUNT_SAMPLED && (__ffsll((long long)alive) - 1) == (int)(threadIdx.x & 63)) { atomicAdd(&g_mesh_count[(LITE ? 16 : 8) + 4], (unsigned long long)__popcll(alive)); atomicAdd(&g_mesh_count[(LITE ? 16 : 8) + 5], 1ull); }
        }
#endif
        if (!SCHED || fetch) { in = tab_get(S.prog + pc, ctl_lds); pc++; }
#ifdef RT_MESH_COUNT
        if constexpr (SCHED) {
            if (fetch && cnt_t0) {          /* the round before: charged to the opcode of the first lane that ran in it */
                const unsigned long long ran = __ballot(1);
                if (RT_COUNT_SAMPLED && (__ffsll((long long)ran) - 1) == (int)(threadIdx.x & 63)) {
                    atomicAdd(&g_op_time[cnt_op], (unsigned long long)(clock64() - cnt_t0)); atomicAdd(&g_op_rounds[cnt_op], 1ull); atomicAdd(&g_op_lanes[cnt_op], (unsigned long long)__popcll(ran));
                }
            }
        }
#endif
        if constexpr (SCHED) {
            /* The lanes of a wave sit at different instructions.  Every jump of a program goes FORWARD (a box that is missed, an empty
               operand, the next candidate of a region, a region's end), so lanes that part ways meet again further down -- if the
               ones ahead wait.  A round therefore runs the lanes at the SMALLEST pc and nobody else: they execute one and the same
               instruction (one opcode, one box, one affine, one chain record for the whole round), and the others join them when
               the round's lanes reach their pc.  Round 2 ran the cheapest CLASS of opcodes present instead (lanes at different
               instructions of one class together): S4 took 166.6 rounds per 64 rays with 27.5 lanes running; this order takes 88.6
               with 51.7 (tools/diverge_count.py) -- and because a round's lanes now touch the same slots of their hit arenas, those
               live in private memory again, whose lane-interleaved layout coalesces such accesses (the pool of round 2, built for
               seven running lanes, is gone: S4 1080p / 4 spp 757 -> 613 ms by the order alone, -> 460 with private arenas).
               A lane still executes its own instructions in its own order: results do not depend on any of this.  The lane at
               the smallest pc always runs, so every lane finishes whatever the program looks like. */
            fetch = true;
            if (__ballot(pc != __builtin_amdgcn_readfirstlane(pc))) {          /* (a wave of like rays skips all this) */
                int mp = 0;             /* the smallest pc among the lanes in the loop, bit by bit from the top (finished lanes hold no value) */
                for (int bit = pc_bits; bit >= 0; --bit) {
                    const bool cand = (pc >> (bit + 1)) == (mp >> (bit + 1));
                    if (!__ballot(cand && !((pc >> bit) & 1))) mp |= 1 << bit;
                }
                fetch = pc == mp;
                if (!fetch) continue;
            }
        }
#ifdef RT_MESH_COUNT
        cnt_t0 = clock64(); cnt_op = in.op;
        {
            const unsigned long long act = __ballot(1);
            unsigned long long rest = act;
            int nd = 0;
            const int op = in.op;
            while (rest) { const int o = __shfl(op, __ffsll((long long)rest) - 1); rest &= ~__ballot(op == o); nd++; }
            if (RT_COUNT_SAMPLED && (__ffsll((long long)act) - 1) == (int)(threadIdx.x & 63)) {
                const int b = LITE ? 16 : 8;
                atomicAdd(&g_mesh_count[b], 1ull); atomicAdd(&g_mesh_count[b + 1], (unsigned long long)nd); atomicAdd(&g_mesh_count[b + 2], (unsigned long long)__popcll(act));
            }
        }
#endif
        switch (in.op) {
        case OP_END:
            goto done;
        case OP_BOUND:
            { const RtBox bx = tab_get(S.boxes + in.a, data_lds); if (!cull_box_hit(bx.lo, bx.hi, cr)) pc = in.b; }
            break;
        case OP_REFBOX:
            { const RtBox bx = tab_get(S.boxes + in.a, data_lds); if (!extent_test(bx.lo, bx.hi, cur)) pc = in.b; }
            break;
        case OP_PUSH_AFFINE: {                                  /* AffineSolid::Intersect, Solids.cpp:899-913 */
            const RtAffine a = tab_get(S.affines + in.a, data_lds);
            DRay nr;
            nr.dir = mat_vec(a.minv, cur.dir);
            nr.org = mat_vec(a.minv, cur.org - d3(a.tr[0], a.tr[1], a.tr[2]));
            rs[depth] = cur;
            depth++;
            cur = nr;
            break;
        }
        case OP_POP_AFFINE:
            depth--;
            cur = rs[depth];
            break;
        case OP_QUADRIC: { PROF_BEGIN(tp); quadric_hits(S.prims[in.a], (uint16_t)in.a, cur, A, in.tag); PROF_END(A, 2, tp); break; }
        case OP_POLY:    { PROF_BEGIN(tp); poly_hits(S.prims[in.a], (uint16_t)in.a, S.planes, cur, A, in.tag); PROF_END(A, 2, tp); break; }
        case OP_SLAB:    { PROF_BEGIN(tp); slab_hits(S.prims[in.a], (uint16_t)in.a, cur, A, in.tag); PROF_END(A, 2, tp); break; }
        case OP_TORUS:
            if constexpr (LITE) { A.bail = 1; goto done; }
            else { PROF_BEGIN(tp); arena_call_out(A, [&](ArenaT<false> &X) { torus_hits(S.prims[in.a], (uint16_t)in.a, cur, X, in.tag); }); PROF_END(A, 2, tp); break; }
        case OP_MESH:
            if constexpr (LITE && !MESHPRE) { A.bail = 1; goto done; }
            else if constexpr (LITE) { mesh_hits<false>(S, S.prims[in.a], (uint16_t)in.a, cur, A, in.tag, L); if (A.bail) goto done; break; }
            else { PROF_BEGIN(tp); arena_call_out(A, [&](ArenaT<false> &X) { mesh_hits<true>(S_in, S.prims[in.a], (uint16_t)in.a, cur, X, in.tag, L_in); }); PROF_END(A, 2, tp); break; }
        case OP_MATERIAL: {                                     /* Material::Intersect, Solids.cpp:1466-1487 */
            if (!top_is(A, 2 * in.b)) break;
            int s = A.t_start, c = A.t_count;
            const RtMaterialDev &m = S.mats[in.a];
            for (int i = 0; i < c; i++) {
                HitRec &h = A.h[s + i];
                if (S.mats[h.mat].density > 0.0) {
                    h.mat = (uint16_t)in.a;
                    hit_add_tex(h, m.tex);
                }
            }
            list_retag_top(A, in.tag);
            break;
        }
        case OP_SURFACE: {                                      /* Surface::Intersect, Solids.cpp:1498-1519 */
            if (!top_is(A, 2 * in.b)) break;
            int s = A.t_start, c = A.t_count;
            const RtSurfaceDev &sf = S.surfs[in.a];
            for (int i = 0; i < c; i++) {
                HitRec &h = A.h[s + i];
                if (h.t == 0.0) continue;
                h.surf = (uint16_t)in.a;
                hit_add_tex(h, sf.tex);
            }
            list_retag_top(A, in.tag);
            break;
        }
        case OP_UNION: {                                        /* Union::Intersect, Solids.cpp:1060-1075 */
            bool r = top_is(A, 2 * in.b + 1);
            bool l = r ? second_is(A, 2 * in.b) : top_is(A, 2 * in.b);
            if (l && r) { PROF_BEGIN(tm); merge_top(A, MERGE_MAX, S.mats, in.tag); PROF_END(A, 9, tm); }
            else if (l || r) list_retag_top(A, in.tag);
            break;
        }
        case OP_INTER: {                                        /* Intersection::Intersect, Solids.cpp:1139-1158 */
            bool r = top_is(A, 2 * in.b + 1);
            bool l = r ? second_is(A, 2 * in.b) : top_is(A, 2 * in.b);
            if (l && r) { PROF_BEGIN(tm); merge_top(A, MERGE_MIN, S.mats, in.tag); PROF_END(A, 9, tm); }
            else if (l || r) drop_top(A);
            break;
        }
        case OP_DIFF: {                                         /* Difference::Intersect, Solids.cpp:1177-1196 */
            bool r = top_is(A, 2 * in.b + 1);
            bool l = r ? second_is(A, 2 * in.b) : top_is(A, 2 * in.b);
            if (l && r) { PROF_BEGIN(tm); merge_top(A, MERGE_DIFF, S.mats, in.tag); PROF_END(A, 9, tm); }
            else if (l) list_retag_top(A, in.tag);
            else if (r) drop_top(A);
            break;
        }
        case OP_SKIP_EMPTY:
            if (!top_is(A, 2 * in.a)) pc = in.b;
            break;
        case OP_REGION: {
            const RtRegion rg = tab_get(S.regions + in.a, ctl_lds);
            if (nrf >= RT_MAX_REGION_DEPTH) { A.overflow |= OVF_REGIONS; pc = rg.end_pc; break; }
            uint16_t *cd = cands[nrf];
            int ov = 0;
            PROF_BEGIN(tb);
            int nc;
            if (TILES && tile_list && nrf == 0 && S.root_cull == 1 && in.a == S.root_cull_arg) {
                /* a primary ray whose sampling tile came with the complete list of root-region children it can touch
                   (k_tile_mask): their boxes -- the very boxes the BVH's leaves hold -- are tested directly, ascending */
                const RtBox *cb = S.region_child_boxes + rg.child_pc0;
                nc = 0;
                for (int q = 0; q < tile_n; q++) {
                    const int c = tile_list[q];
                    if (cull_box_hit(cb[c].lo, cb[c].hi, cr)) cd[nc++] = (uint16_t)c;
                }
            } else
                nc = region_candidates(S, L, rg, cr, cd, ov);
            PROF_END(A, 1, tb);
            A.overflow |= ov;
            if (nc == 0) { pc = rg.end_pc; break; }
            if (nrf > 0) saved[nrf - 1] = cf;
            cf.ncand = nc; cf.pos = 0; cf.base_lists = A.nlist; cf.nl_before = A.nlist; cf.out_tag = in.tag;
            cf.child_pc0 = rg.child_pc0; cf.lca0 = rg.lca0; cf.nchild = rg.n_children; cf.uid = rg.uid; cf.end_pc = rg.end_pc;
            cf.cur_c = cd[0]; cf.top_idx = 0; cf.top_gap = -1;
            nrf++;
            pc = tab_get(S.region_child_pc + (cf.child_pc0 + cf.cur_c), ctl_lds);
            break;
        }
        case OP_CHAIN: {
            PROF_BEGIN(tc);
            /* CH_LDS: every chain record of the scene is staged in LDS (decided per launch, so that the loads below
               are LDS loads and not generic ones) */
            const RtChain &ch = CH_LDS ? L.chains[in.a] : S.chains[in.a];
            if (ch.test_box & RT_CHAIN_TEST_BOX) {
                const RtBox bx = tab_get(S.boxes + ch.box, data_lds);
                if (!cull_box_hit(bx.lo, bx.hi, cr)) { if (in.b & 1) goto region_ret; break; }
            }
            DRay r = cur;
            if (ch.n_affine) {                                      /* AffineSolid::Intersect, Solids.cpp:899-913 */
                DRay nr;
                nr.dir = mat_vec(ch.minv, r.dir);
                nr.org = mat_vec(ch.minv, r.org - d3(ch.tr[0], ch.tr[1], ch.tr[2]));
                r = nr;
            }
            RtPrim pr;
            pr.p[0] = ch.p[0]; pr.p[1] = ch.p[1]; pr.p[2] = ch.p[2]; pr.p[3] = 0.0;
            pr.i0 = ch.i0; pr.plane0 = ch.plane0; pr.leaf_affine = ch.leaf_affine; pr.id = 0;
            const int before = A.nlist;
            PROF_END(A, 4, tc);
            {
                PROF_BEGIN(tp);
                switch (ch.prim_op) {
                case OP_QUADRIC: quadric_hits(pr, (uint16_t)ch.prim, r, A, in.tag, &ch); break;
                case OP_POLY:    poly_hits(pr, (uint16_t)ch.prim, S.planes, r, A, in.tag, &ch); break;
                case OP_SLAB:    slab_hits(pr, (uint16_t)ch.prim, r, A, in.tag, &ch); break;
                default:
                    if constexpr (LITE && !MESHPRE) { A.bail = 1; goto done; }
                    else if constexpr (LITE) {
                        if (ch.prim_op != OP_MESH) { A.bail = 1; goto done; }
                        mesh_hits<false>(S, pr, (uint16_t)ch.prim, r, A, in.tag, L);
                        if (A.bail) goto done;
                        if (A.nlist > before)
                            for (int i = 0; i < A.t_count; i++) { HitRec h = A.h[A.t_start + i]; hit_apply_wraps(ch, h); A.h[A.t_start + i] = h; }
                        break;
                    } else {
                        if (ch.prim_op == OP_MESH) arena_call_out(A, [&](ArenaT<false> &X) { mesh_hits<true>(S_in, pr, (uint16_t)ch.prim, r, X, in.tag, L_in); });
                        else arena_call_out(A, [&](ArenaT<false> &X) { torus_hits(pr, (uint16_t)ch.prim, r, X, in.tag); });
                        if (A.nlist > before)
                            for (int i = 0; i < A.t_count; i++) { HitRec h = A.h[A.t_start + i]; hit_apply_wraps(ch, h); A.h[A.t_start + i] = h; }
                        break;
                    }
                }
                PROF_END(A, 2, tp);
            }
            if (in.b & 1) goto region_ret;
            break;
        }
        case OP_RET:
        region_ret: {
            PROF_BEGIN(tr);
            /* one region child done: fold the non-empty lists in the union tree's own order */
            const int ctag = 2 * cf.uid;
            const int c = cf.cur_c;
            if (A.nlist > cf.nl_before) {           /* the child left a list on top */
                const int e = A.nlist - 1;
                const int gap = (e > cf.base_lists) ? region_gap(S, cf, cf.top_idx, c) : -1;
                A.l_idx[e] = (uint16_t)c;
                A.l_gap[e] = gap;
                cf.top_idx = c; cf.top_gap = gap;
            }
            cf.pos++;
            if (cf.pos < cf.ncand) {
                const int next = cands[nrf - 1][cf.pos];
                while (A.nlist - cf.base_lists >= 2 && cf.top_gap > region_gap(S, cf, cf.top_idx, next)) {
                    merge_top(A, MERGE_MAX, S.mats, ctag);
                    if (A.nlist > cf.base_lists) { cf.top_gap = A.l_gap[A.nlist - 1]; cf.top_idx = A.l_idx[A.nlist - 1]; }
                }
                cf.nl_before = A.nlist;
                cf.cur_c = next;
                pc = tab_get(S.region_child_pc + (cf.child_pc0 + next), ctl_lds);
            } else {
                while (A.nlist - cf.base_lists >= 2) merge_top(A, MERGE_MAX, S.mats, ctag);
                if (A.nlist - cf.base_lists == 1) list_retag_top(A, cf.out_tag);
                nrf--;
                pc = cf.end_pc;
                if (nrf > 0) cf = saved[nrf - 1];
            }
            PROF_END(A, 6, tr);
            break;
        }
        case OP_ANGCLIP: {                                      /* AngularClip::Intersect, Solids.cpp:1224-1241 */
            if (!top_is(A, 2 * in.b)) break;
            const RtClip &c = S.clips[in.a];
            Work w;
            work_init(w);
            clip_plane(cur, d3(c.cv1[0], c.cv1[1], c.cv1[2]), w, 0, 0.0);
            clip_plane(cur, d3(c.cv2[0], c.cv2[1], c.cv2[2]), w, 1, 0.0);
            int before = A.nlist;
            hit_list2(w, A, 2 * in.b + 1, (uint16_t)(0x8000 | in.a), c.leaf_affine);
            if (A.nlist > before) merge_top(A, c.intersect ? MERGE_MIN : MERGE_DIFF, S.mats, in.tag);
            else {
                /* Merge(body, empty): every body hit is still filtered through op(pmLeft, vacuum) */
                if (A.top + 1 <= RT_ARENA && A.nlist < RT_LIST_STACK) {
                    list_push(A, A.top, 0, 2 * in.b + 1);
                    merge_top(A, c.intersect ? MERGE_MIN : MERGE_DIFF, S.mats, in.tag);
                } else A.overflow |= OVF_ARENA;
            }
            break;
        }
        default:
            break;
        }
    }
done:
    PROF_END(A, 3, t_all);
    if (A.bail) { start = 0; return 0; }
    if (A.nlist > 0 && A.t_tag == 0) { start = A.t_start; return A.t_count; }
    start = 0;
    return 0;
}

/* the FULL kernels' entry: a function of its own (inlined into those kernels it costs the second wave per SIMD) that runs the
   interpreter, with everything it calls but the torus and the mesh walk inlined, on a view of the caller's arena (ArenaView) */
template <bool CH_LDS, bool TILES, bool SCHED>
__device__ __noinline__ __attribute__((flatten)) int intersect_scene_full(const RtDevScene &S, const SceneLds &L, const DRay &wray, double tmax, ArenaT<false> &A, int &start,
                                                                          const unsigned short *tile_list, int tile_n, int tile_one, int mesh_slot)
{
    ArenaView V;
    V.h = uniform_ptr(&A.h[0]);
    V.l_start = uniform_ptr(&A.l_start[0]); V.l_count = uniform_ptr(&A.l_count[0]); V.l_idx = uniform_ptr(&A.l_idx[0]);
    V.l_tag = uniform_ptr(&A.l_tag[0]); V.l_gap = uniform_ptr(&A.l_gap[0]);
    V.home = uniform_ptr(&A);
    int st = 0;
    const int count = intersect_scene<false, CH_LDS, TILES, false, SCHED>(S, L, wray, tmax, V, st, tile_list, tile_n, tile_one, mesh_slot);
    V.commit();
    start = st;
    return count;
}
template <bool LITE, bool CH_LDS, bool TILES, bool MESHPRE, bool SCHED, class AR>
RT_DEV int trace_scene(const RtDevScene &S, const SceneLds &L, const DRay &wray, double tmax, AR &A, int &start,
                       const unsigned short *tile_list, int tile_n, int tile_one, int mesh_slot)
{
#ifndef RT_SCHED_POOL
    if constexpr (!LITE) return intersect_scene_full<CH_LDS, TILES, SCHED>(S, L, wray, tmax, A, start, tile_list, tile_n, tile_one, mesh_slot);
    else
#endif
    return intersect_scene<LITE, CH_LDS, TILES, MESHPRE, SCHED>(S, L, wray, tmax, A, start, tile_list, tile_n, tile_one, mesh_slot);
}

/* ================================================================== lattice noise (Texture.cpp:11-191) */
RT_DEV uint32_t lattice(uint32_t n)
{
    n = 1099087573u * n;
    uint32_t s = n & 31u;
    return s ? ((n >> s) | (n << (32u - s))) : n;
}
struct F4 { float x, y, z, w; };
RT_DEV float dot4(const F4 &a, const F4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
#define RT_HALF32 (0.5f * 4294967296.0f)

struct NoiseSetup { F4 coef[4], grad[4]; uint32_t flr[4]; float gain; };
RT_DEV void noise_setup(NoiseSetup &s, const float *p, int ndim)
{
    s.gain = 1.0f;
    for (int i = 0; i < ndim; i++) {
        s.gain *= 1.27f;
        int d = ndim - i;
        int fl = (int)p[i] - ((p[i] < 0.0f && p[i] != (float)(int)p[i]) ? 1 : 0);
        float x = p[i] - (float)fl;
        s.flr[d] = (uint32_t)fl;
        s.coef[d].x = 1.0f / 6.0f + x * (x * (0.5f - 1.0f / 6.0f * x) - 0.5f);
        s.coef[d].y = 4.0f / 6.0f + x * x * (0.5f * x - 1.0f);
        s.coef[d].z = 1.0f / 6.0f + 0.5f * x * (1.0f + x * (1.0f - x));
        s.coef[d].w = 1.0f / 6.0f * x * x * x;
        s.grad[d].x = -0.5f + x * (1.0f - 0.5f * x);
        s.grad[d].y = x * (1.5f * x - 2.0f);
        s.grad[d].z = 0.5f + x * (1.0f - 1.5f * x);
        s.grad[d].w = 0.5f * x * x;
    }
}
RT_DEV F4 knots4(uint32_t f, uint32_t lat, uint32_t *h)
{
    h[0] = lattice((f - 1) ^ lat); h[1] = lattice(f ^ lat); h[2] = lattice((f + 1) ^ lat); h[3] = lattice((f + 2) ^ lat);
    F4 k;
    k.x = (float)h[0] - RT_HALF32; k.y = (float)h[1] - RT_HALF32; k.z = (float)h[2] - RT_HALF32; k.w = (float)h[3] - RT_HALF32;
    return k;
}
/* GradientSpline at dimension 1 / 2 / 3 (Texture.cpp:127-157), recursion unrolled.
   g1(lat): value, d/dx1.  g2: value, d/dx2 (own), d/dx1.  g3: value, d/dx3, d/dx2, d/dx1 -> stored as grad[0..2]
   in the reference's order: rgfGradient[0] = own dimension, [1] = next lower, ... */
RT_DEV float gspline1(const NoiseSetup &s, uint32_t lat, float *g0)
{
    uint32_t h[4];
    F4 k = knots4(s.flr[1], lat, h);
    *g0 = dot4(s.grad[1], k);
    return dot4(s.coef[1], k);
}
RT_DEV float gspline2(const NoiseSetup &s, uint32_t lat, float *g0, float *g1)
{
    uint32_t f = s.flr[2];
    uint32_t h0 = lattice((f - 1) ^ lat), h1 = lattice(f ^ lat), h2 = lattice((f + 1) ^ lat), h3 = lattice((f + 2) ^ lat);
    F4 k, gk;
    k.x = gspline1(s, h0, &gk.x); k.y = gspline1(s, h1, &gk.y); k.z = gspline1(s, h2, &gk.z); k.w = gspline1(s, h3, &gk.w);
    *g1 = dot4(s.coef[2], gk);
    *g0 = dot4(s.grad[2], k);
    return dot4(s.coef[2], k);
}
RT_DEV float gspline3(const NoiseSetup &s, uint32_t lat, float *g0, float *g1, float *g2)
{
    uint32_t f = s.flr[3];
    uint32_t h0 = lattice((f - 1) ^ lat), h1 = lattice(f ^ lat), h2 = lattice((f + 1) ^ lat), h3 = lattice((f + 2) ^ lat);
    F4 k, ga, gb;
    k.x = gspline2(s, h0, &ga.x, &gb.x); k.y = gspline2(s, h1, &ga.y, &gb.y); k.z = gspline2(s, h2, &ga.z, &gb.z); k.w = gspline2(s, h3, &ga.w, &gb.w);
    *g1 = dot4(s.coef[3], ga);
    *g2 = dot4(s.coef[3], gb);
    *g0 = dot4(s.grad[3], k);
    return dot4(s.coef[3], k);
}
/* GradientBandNoise (Texture.cpp:159-191), nPow == 1, ndim 2 or 3 */
__device__ float gradient_band_noise(float *g, const float *p, int ndim)
{
    NoiseSetup s;
    noise_setup(s, p, ndim);
    float v;
    if (ndim == 3) v = gspline3(s, 123456789u, &g[0], &g[1], &g[2]);
    else { v = gspline2(s, 123456789u, &g[0], &g[1]); g[2] = 0.0f; }
    v = s.gain * v * (0.9f / 4294967296.0f) + 0.5f;
    for (int i = 0; i < ndim; i++) g[i] *= s.gain * (0.9f / 4294967296.0f);
    return v;
}

/* ================================================================== shading helpers (Shade.cpp) */
struct ShadeHit {           /* a hit as the shader sees it after TransformNormals */
    double t;
    D3 n, np;
    int perturbed;
    int surf, mat;
    /* Q13: a solid colour texture sets the colour (Spots: the luster) of the material this hit enters; the reference
       writes them into the shared Material, here they belong to the hit (the oracle does the same) */
    int has_color, luster_ov;       /* luster_ov < 0: the material's own */
    int rough_ov;                   /* PlanetMapping writes ps->nRoughness = SURF_SHINY (TextureMaps.cpp:1041, 1049): the same policy; < 0: the surface's own */
    Spec color;
};
RT_DEV int hit_roughness(const RtDevScene &S, const ShadeHit &h) { return h.rough_ov >= 0 ? h.rough_ov : S.surfs[h.surf].roughness; }

RT_DEV void perturb(ShadeHit &h, const D3 &d)      /* PerturbNormal, TextureMaps.cpp:189-198 */
{
    if (h.perturbed) h.np = h.np + d;
    else { h.np = normalize(h.n) + d; h.perturbed++; }
}

__device__ const double kDodeca[20][3] = {
#define T_  (1.0 / RT_SQRT3)
#define P1_ (1.61803398874989484820458683436563811772030917980576 / RT_SQRT3)
#define P2_ ((1.61803398874989484820458683436563811772030917980576 - 1.0) / RT_SQRT3)
    { T_, T_, T_ }, { T_, T_, -T_ }, { T_, -T_, T_ }, { T_, -T_, -T_ }, { -T_, T_, T_ }, { -T_, T_, -T_ }, { -T_, -T_, T_ }, { -T_, -T_, -T_ },
    { 0.0, P2_, P1_ }, { 0.0, P2_, -P1_ }, { 0.0, -P2_, P1_ }, { 0.0, -P2_, -P1_ },
    { P1_, 0.0, P2_ }, { P1_, 0.0, -P2_ }, { -P1_, 0.0, P2_ }, { -P1_, 0.0, -P2_ },
    { P2_, P1_, 0.0 }, { P2_, -P1_, 0.0 }, { -P2_, P1_, 0.0 }, { -P2_, -P1_, 0.0 }
#undef T_
#undef P1_
#undef P2_
};


/* VHash3 + GradientVoronoi, Texture.cpp:318-332, :380-431: nearest feature point among the lattice cells the current
   best distance can reach; the gradient quirk for nPow > 2 (scaled by fmin only nPow - 2 times) is kept */
RT_DEV void vhash3(int nx, int ny, int nz, float &x, float &y, float &z)
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
__device__ float gradient_voronoi(float *grad, const float *p, int npow)
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

/* GradientFractalNoise, Texture.cpp:243-297: fractal dimension 1, lacunarity 2.1, whole number of levels; omega =
   powf(2.1f, -1.0f) comes from the host */
__device__ float gradient_fractal_noise(float *grad, const float *p, int ndim, int levels, float omega)
{
    const float lac = 2.1f;
    float wl = 1.0f, scale = 1.0f, noise = 0.0f, gain = 0.0f, sc[3] = { 0.0f, 0.0f, 0.0f }, bg[3] = { 0.0f, 0.0f, 0.0f };
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

/* BandNoise (Texture.cpp:98-123, nPow == 1, three dimensions): the value-only spline over its own lattice seed */
RT_DEV float band_noise3(const float *p)
{
    NoiseSetup s;
    noise_setup(s, p, 3);
    float g0, g1, g2;
    const float v = gspline3(s, 1099087573u, &g0, &g1, &g2);       /* Spline(nDim, 1099087573): the same knots and coefficients */
    return s.gain * v * (0.9f / 4294967296.0f) + 0.5f;
}
/* FractalNoise, Texture.cpp:200-241: fractal dimension 1, lacunarity 1.9, whole levels */
RT_DEV float fractal_noise3(const float *p, int levels, int absolute, float omega19)
{
    const float lac = 1.9f;
    float wl = 1.0f, noise = 0.0f, gain = 0.0f, sc[3] = { p[0], p[1], p[2] };
    for (int i = 0; i < levels; i++) {
        float fb = wl * (band_noise3(sc) - 0.5f);
        if (absolute) fb = fabsf(fb);
        noise += fb;
        gain += wl;
        wl *= omega19;
        for (int j = 0; j < 3; j++) sc[j] *= lac;
    }
    return noise / gain + 0.5f;
}
RT_DEV double sawtooth_wave(double x)                    /* TextureMaps.cpp:546-555 */
{
    x = (0.5 / RT_PI) * x + 0.25;
    x = 2.0 * (x - floor(x));
    if (x < 1.0) return 2.0 * x - 1.0;
    return 3.0 - 2.0 * x;
}
RT_DEV double wood_wave(double x)                        /* TextureMaps.cpp:590-604 */
{
    x = (0.5 / RT_PI) * x + 0.25;
    x = 2.0 * (x - floor(x));
    if (x < 1.0) return 0.0;
    else if (x < 1.5) { x = 2.0 * (x - 1.0); return 0.5 * x * x; }
    else { x = 2.0 * (x - 2.0); return -0.5 * x * x + 1.0; }
}
/* solid colour textures, TextureMaps.cpp:518-683 (Q13: the colour belongs to the hit) */
RT_DEVN void colour_texture(const RtDevScene &S, int tex, const D3 &p, ShadeHit &h)
{
    const double sc = tex == RT_TEX_WOOD ? 3.0 : 7.0;
    float q[3] = { (float)(p.x * sc), (float)(p.y * sc), (float)(p.z * sc) };
    const Spec green = spec_load(S.tex_palette[0]), white = spec_load(S.tex_palette[1]), orange = spec_load(S.tex_palette[2]),
               red = spec_load(S.tex_palette[3]), blue = spec_load(S.tex_palette[4]);
    h.has_color = 1;
    switch (tex) {
    case RT_TEX_VORONOI: {
        float g[3];
        double fBlend = gradient_voronoi(g, q, 1) / 1.6;
        if (fBlend > 1.0) fBlend = 1.0;
        h.color = green * fBlend + white * (1.0 - fBlend);
        return;
    }
    case RT_TEX_AGATE: {
        double fBlend = band_noise3(q);
        fBlend = 0.5 * (sin(100.0 * fBlend) + 1.0);
        h.color = orange * fBlend + white * (1.0 - fBlend);
        return;
    }
    case RT_TEX_MARBLE: {
        const Spec pink = white * 0.67 + red * 0.33;
        double d = sawtooth_wave(q[2] + 12.0 * fractal_noise3(q, 5, 1, S.fractal_omega19));
        d = d * d;
        h.color = pink * (1.0 - d);
        return;
    }
    case RT_TEX_CLOUDS: {
        double fBlend = 2.0 * fractal_noise3(q, 5, 0, S.fractal_omega19) - 1.0;
        if (fBlend < 0.0) fBlend = 0.0; else if (fBlend > 1.0) fBlend = 1.0;
        h.color = white * fBlend + blue * (1.0 - fBlend);
        return;
    }
    case RT_TEX_WOOD: {
        const double fWarp = 3.0f * fractal_noise3(q, 2, 0, S.fractal_omega19);
        q[0] += (float)fWarp;
        double fBlend = 0.5 + 0.5 * wood_wave(50.0 * sqrt(q[0] * q[0] + q[1] * q[1]));
        fBlend = 0.9 * fBlend + 0.1 * sawtooth_wave(250.0 * (1.0 + fBlend) * sqrt(q[0] * q[0] + q[1] * q[1]));
        fBlend = 1.0 - fBlend;
        h.color = (orange + red) * (0.25 * fBlend) + white * 0.03;
        return;
    }
    case RT_TEX_SPOTS: {
        const double fBlend = band_noise3(q);
        if (fBlend > 0.5) { h.luster_ov = RT_MAT_COPPER; h.has_color = 0; }
        else { h.luster_ov = RT_MAT_PLASTIC; h.color = spec_const(0.0f); }
        return;
    }
    default: {      /* RT_TEX_VEINS */
        const double fBlend = fractal_noise3(q, 5, 0, S.fractal_omega19) - 0.5;
        if (fabs(fBlend) > 0.02) h.color = green * 0.125; else h.color = white;
        return;
    }
    }
}

/* ---- the planet image maps: Image::Get with its wrap modes (Image.h:226-236, Image.cpp:50-89), SampleLerp / SampleLerpRGB
   (Image.cpp:1262-1307), SampleGradient (Image.h:187-189), PlanetMapping (TextureMaps.cpp:955-1084) */
RT_DEV int img_wrap(int i, int len, int wrap)
{
    if (wrap == RT_WRAP_AROUND) { i = i % len; if (i < 0) i += len; return i; }
    if (wrap == RT_WRAP_MIRROR) { i = i % (2 * len); if (i < 0) i += 2 * len; if (i >= len) i = 2 * len - i - 1; return i; }
    return i < 0 ? 0 : (i >= len ? len - 1 : i);
}
RT_DEV float img_get(const RtImageDev &im, int i, int j, int k)
{
    if ((unsigned)i >= (unsigned)im.w || (unsigned)j >= (unsigned)im.h) { i = img_wrap(i, im.w, im.wrap_x); j = img_wrap(j, im.h, im.wrap_y); }
    return im.px[(size_t)i + (size_t)im.w * ((size_t)j + (size_t)im.h * (size_t)k)];
}
RT_DEV double img_sample_lerp(const RtImageDev &im, double x, double y, int k = 0)
{
    x -= 0.5; y -= 0.5;
    const double fx = floor(x), fy = floor(y);
    const int i1 = (int)(long long)fx, j1 = (int)(long long)fy;     /* "unsigned(floor(x))", then Get(int): the same 32 bits */
    const double dx = x - fx, dy = y - fy;
    double a = img_get(im, i1, j1, k), b = img_get(im, i1 + 1, j1, k), c = img_get(im, i1, j1 + 1, k), d = img_get(im, i1 + 1, j1 + 1, k);
    a = a + dx * (b - a);
    c = c + dx * (d - c);
    return a + dy * (c - a);
}
RT_DEV void img_sample_lerp_rgb(const RtImageDev &im, double x, double y, float rgb[3])
{
    x -= 0.5; y -= 0.5;
    const double fx = floor(x), fy = floor(y);
    const int i1 = (int)(long long)fx, j1 = (int)(long long)fy;
    const double dx = x - fx, dy = y - fy;
    for (int k = 0; k < 3; k++) {
        const int kc = im.ch < 3 ? 0 : k;               /* GetRGB of a grey image: the one plane three times (Image.h:275-276) */
        float a = img_get(im, i1, j1, kc), b = img_get(im, i1 + 1, j1, kc), c = img_get(im, i1, j1 + 1, kc), d = img_get(im, i1 + 1, j1 + 1, kc);
        /* DisplayRGB arithmetic (Image.h:67-69): float sums and differences, products with the double rounded to float */
        a = a + (float)((double)(b - a) * dx);
        c = c + (float)((double)(d - c) * dx);
        rgb[k] = a + (float)((double)(c - a) * dy);
    }
}
RT_DEV void img_sample_gradient(const RtImageDev &im, double x, double y, float g[2])
{
    g[0] = (float)(img_sample_lerp(im, x + 0.5, y) - img_sample_lerp(im, x - 0.5, y));
    g[1] = (float)(img_sample_lerp(im, x, y + 0.5) - img_sample_lerp(im, x, y - 0.5));
}
#define RT_SPHEREMAPWIDTH (8 * 1024)        /* TextureMaps.cpp:943: the maps are addressed as if 8192 wide, whatever their size */
#define RT_MAPSELECT 32.70                  /* :691 */
RT_DEVN void planet_texture(const RtDevScene &S, int tex, const D3 &p, ShadeHit &h)
{
    /* colour, bump, water: the Image arguments of the wrappers (TextureMaps.cpp:1104-1242); the bump map's slot follows its colour map's */
    int sl[3];
    sl[0] = tex == RT_TEX_SKY ? RT_IMG_SKY : (tex <= RT_TEX_EARTH2 ? RT_IMG_EARTH : RT_IMG_MARS + 2 * (tex - RT_TEX_MARS));
    sl[1] = tex == RT_TEX_SKY ? -1 : sl[0] + 1;
    sl[2] = tex == RT_TEX_EARTH ? RT_IMG_EARTHWATER : -1;
    const RtImageDev &imColor = S.images[sl[0]];
    const D3 cvN = normalize(h.n);                  /* "cvN = Normalize(ph->cvNormal)": the normal as it stands at this call-out */
    double x = cvN.x, y = cvN.y, z = cvN.z;
    const double fLat = asin(z);
    double fLong = RT_PI - atan2(p.y, p.x);
    const double w = (double)RT_SPHEREMAPWIDTH, fKmPerPixel = 2.0 * RT_PI * 6378.137 / w;
    if (fLong < 0.0) fLong += 2.0 * RT_PI;
    const double fSinLat = z, fCosLat = sqrt(1.0 - z * z), r = sqrt(x * x + y * y);
    const double fSinLong = y / r, fCosLong = x / r;
    double fScale;
    bool polar;
    if (fabs(fLat * 180.0 / RT_PI) > RT_MAPSELECT) {            /* stereographic projection of the polar regions */
        if (fabs(fLat) < 0.5 * RT_PI - 0.001) fScale = (0.5 * RT_PI - fabs(fLat)) / r;
        else fScale = 1 / fabs(z);
        double xp, yp;
        if (fLat > 0.0) {
            xp = +x / (1.0 + z) / RT_PI + 0.25;
            yp = +y / (1.0 + z) / RT_PI + 0.25;
            if (xp < 0.0 || xp >= 0.5) fScale = 0.0;
        } else {
            xp = +x / (1.0 - z) / RT_PI + 0.75;
            yp = -y / (1.0 - z) / RT_PI + 0.25;
            if (xp < 0.5 || xp >= 1.0) fScale = 0.0;
        }
        if (yp < 0.0 || yp >= 0.5) fScale = 0.0;
        x = xp; y = yp; polar = true;
    } else {                                                    /* Mercator projection of the equatorial region */
        double xc = fLong / (2.0 * RT_PI);
        if (xc >= 1.0) xc -= 1.0; else if (xc < 0.0) xc += 1.0;
        const double yc = -log(tan(0.25 * RT_PI + 0.5 * fLat)) / (2.0 * RT_PI) + 0.75;
        if (yc >= 0.5 && yc < 1.0) fScale = 1.0 / r; else fScale = 0.0;
        x = xc; y = yc; polar = false;
    }
    double xGrad, yGrad, zGrad;
    float g[2];
    if (sl[2] >= 0 && img_sample_lerp(S.images[sl[2]], w * x, w * y) > 0.0) {
        h.has_color = 1; h.color = spec_load(S.planet_spec + 5 * RT_NSPEC);        /* s_sOcean */
        h.rough_ov = RT_SURF_SHINY;
        img_sample_gradient(S.images[sl[2]], w * x, w * y, g);
        xGrad = g[0] * 256.0 * fScale / fKmPerPixel;
        yGrad = g[1] * 256.0 * fScale / fKmPerPixel;
        zGrad = 1.0;
    } else {
        float rgb[3];
        img_sample_lerp_rgb(imColor, w * x, w * y, rgb);
        /* RGBtoSpectrum(rgb, 1), RadiometryColor.cpp:947-963 */
        const Spec sw = spec_load(S.planet_spec), sr = spec_load(S.planet_spec + RT_NSPEC), sg = spec_load(S.planet_spec + 2 * RT_NSPEC),
                   sb = spec_load(S.planet_spec + 3 * RT_NSPEC);
        Spec s;
        if (rgb[0] < rgb[1] && rgb[0] < rgb[2]) s = sw * (double)rgb[0] + sg * (double)(rgb[1] - rgb[0]) + sb * (double)(rgb[2] - rgb[0]);
        else if (rgb[1] < rgb[0] && rgb[1] < rgb[2]) s = sw * (double)rgb[1] + sr * (double)(rgb[0] - rgb[1]) + sb * (double)(rgb[2] - rgb[1]);
        else s = sw * (double)rgb[2] + sr * (double)(rgb[0] - rgb[2]) + sg * (double)(rgb[1] - rgb[2]);
        const float *d65 = S.planet_spec + 4 * RT_NSPEC;
        for (int i = 0; i < RT_NSPEC; i++) s.v[i] = d65[i] != 0.0f ? s.v[i] / d65[i] : 0.0f;
        h.has_color = 1; h.color = s;
        if (sl[1] >= 0) {
            h.rough_ov = RT_SURF_SHINY;
            img_sample_gradient(S.images[sl[1]], w * x, w * y, g);
            xGrad = g[0] * 256.0 * fScale / fKmPerPixel;
            yGrad = g[1] * 256.0 * fScale / fKmPerPixel;
            zGrad = 0.0;
        } else {
            /* sky domes: the reference passes three uninitialised doubles to PerturbNormal here (TextureMaps.cpp:1055-1083).  Policy Q15:
               the normal stays as it is */
            h.luster_ov = RT_MAT_NOSHADE;
            return;
        }
    }
    double xN, yN, zN;
    if (polar) {
        x = +fCosLong * xGrad - -fSinLong * yGrad;
        y = -fSinLong * xGrad + +fCosLong * yGrad;
        z = zGrad;
        xN = fCosLat * z - fSinLat * x;
        yN = y;
        zN = fSinLat * z + fCosLat * x;
        x = fCosLong * xN - fSinLong * yN;
        y = fSinLong * xN + fCosLong * yN;
        z = zN;
    } else {
        y = xGrad;
        x = fCosLat * zGrad - fSinLat * yGrad;
        z = fSinLat * zGrad + fCosLat * yGrad;
        zN = z;
        xN = fCosLong * x - fSinLong * y;
        yN = fSinLong * x + fCosLong * y;
        z = zN; x = xN; y = yN;
    }
    perturb(h, d3(x, y, z));
}

RT_DEVN void apply_texture(const RtDevScene &S, int tex, const D3 &p, ShadeHit &h, const double *aux, float omega)        /* TextureMaps.cpp:209-369 */
{
    float q[3], g[3];
    if (tex >= RT_TEX_VORONOI && tex <= RT_TEX_VEINS) { colour_texture(S, tex, p, h); return; }
    if (tex >= RT_TEX_EARTH && tex <= RT_TEX_SKY) { planet_texture(S, tex, p, h); return; }
    switch (tex) {
    case RT_TEX_BUMPY: case RT_TEX_MICROFACETS: case RT_TEX_MICROFACETS2: {
        double sc = tex == RT_TEX_BUMPY ? 50.0 : (tex == RT_TEX_MICROFACETS ? 500.0 : 5000.0);
        q[0] = (float)(p.x * sc); q[1] = (float)(p.y * sc); q[2] = (float)(p.z * sc);
        gradient_band_noise(g, q, 3);
        perturb(h, d3(g[0], g[1], g[2]) * 0.3);
        return;
    }
    case RT_TEX_FIBERS:
        q[0] = (float)(p.x * 250.0); q[1] = (float)(p.y * 250.0); q[2] = 0.0f;
        gradient_band_noise(g, q, 2);
        perturb(h, d3(g[0], g[1], 0.0) * 0.3);
        return;
    case RT_TEX_BLISTERED: {
        q[0] = (float)(p.x * 25.0); q[1] = (float)(p.y * 25.0); q[2] = (float)(p.z * 25.0);
        float f = gradient_band_noise(g, q, 3);
        if (f > 0.5) f = -0.5f; else f = 0.0f;
        perturb(h, d3(g[0], g[1], g[2]) * (double)f);
        return;
    }
    case RT_TEX_FACETED20: {
        q[0] = (float)(p.x * 50.0); q[1] = (float)(p.y * 50.0); q[2] = (float)(p.z * 50.0);
        gradient_band_noise(g, q, 3);
        D3 cv = d3(g[0], g[1], g[2]), best = d3(0, 0, 0);
        double bestd = -1000000.0;
        for (int i = 0; i < 20; i++) {
            D3 c = d3(kDodeca[i][0], kDodeca[i][1], kDodeca[i][2]);
            double dd = dot(cv, c);
            if (dd > bestd) { bestd = dd; best = c; }
        }
        perturb(h, best * 0.15);
        return;
    }
    case RT_TEX_WRINKLED:
        q[0] = (float)(p.x * 50.0); q[1] = (float)(p.y * 50.0); q[2] = (float)(p.z * 50.0);
        gradient_fractal_noise(g, q, 3, 6, omega);
        perturb(h, d3(g[0], g[1], g[2]) * 0.3);
        return;
    case RT_TEX_DENTED: {
        D3 p2 = p + d3(13.0, 17.0, 31.0);
        q[0] = (float)(p2.x * 40.0); q[1] = (float)(p2.y * 40.0); q[2] = (float)(p2.z * 40.0);
        float f = gradient_fractal_noise(g, q, 3, 5, omega);
        if (f < 0.5f) { h.np = h.n; return; }               /* cvPerturbed = cvNormal, nPerturbed untouched */
        f = 3.0f * f * f;
        perturb(h, d3(g[0], g[1], g[2]) * (1.0 * (double)f));
        return;
    }
    case RT_TEX_HAMMERED: {
        q[0] = (float)(p.x * 8.0); q[1] = (float)(p.y * 8.0); q[2] = (float)(p.z * 8.0);
        gradient_band_noise(g, q, 3);
        D3 cv = d3(g[0], g[1], g[2]);
        gradient_fractal_noise(g, q, 3, 5, omega);
        D3 cv2 = d3(g[0], g[1], g[2]);
        perturb(h, cv * 0.1 + cv2 * 0.1);
        return;
    }
    case RT_TEX_TECTONIC: {
        q[0] = (float)(p.x * 3.0); q[1] = (float)(p.y * 3.0); q[2] = (float)(p.z * 3.0);
        float f = gradient_fractal_noise(g, q, 3, 6, omega);
        if (f < 0.5f) f = 0.0f; else f = -0.4f;
        perturb(h, d3(g[0], g[1], g[2]) * (double)f);
        return;
    }
    case RT_TEX_VORONOIBUMP: case RT_TEX_VORONOIANTIBUMP: case RT_TEX_VORONOICRINKLED:
        q[0] = (float)(p.x * 7.0); q[1] = (float)(p.y * 7.0); q[2] = (float)(p.z * 7.0);
        gradient_voronoi(g, q, tex == RT_TEX_VORONOICRINKLED ? 4 : 1);
        if (tex == RT_TEX_VORONOIBUMP) perturb(h, d3(g[0], g[1], g[2]) * 0.3);
        else perturb(h, d3(-g[0], -g[1], -g[2]) * 0.3);
        return;
    case RT_TEX_WAVY: {
        double f = 0.3 * cos(p.z / (2.0 * RT_PI));
        perturb(h, d3(f, f, f));
        return;
    }
    case RT_TEX_QUASIMICROFACET: {
        D3 n = h.perturbed ? h.np : h.n;
        D3 cu = normalize(cross(n, d3(0.0, 1.0, 0.0)));
        D3 cvv = normalize(cross(n, cu));
        double u = aux[4], v = aux[5];
        u = u * u * u * u * u;
        v = v * v * v * v * v;
        perturb(h, (cu * u + cvv * v) * 0.25);
        return;
    }
    default:
        return;
    }
}

/* the ray in the frame of affine k: apply the enclosing transforms root-first (Solids.cpp:905-906) */
__device__ DRay ray_in_frame(const RtDevScene &S, const DRay &wray, int k)
{
    int chain[RT_MAX_AFFINE_DEPTH], n = 0;
    for (; k >= 0 && n < RT_MAX_AFFINE_DEPTH; k = S.affines[k].tree_parent) chain[n++] = k;
    DRay r = wray;
    for (int i = n - 1; i >= 0; --i) {
        const RtAffine &a = S.affines[chain[i]];
        DRay nr;
        nr.dir = mat_vec(a.minv, r.dir);
        nr.org = mat_vec(a.minv, r.org - d3(a.tr[0], a.tr[1], a.tr[2]));
        r = nr;
    }
    return r;
}

/* TransformNormals, Shade.cpp:264-284.  LITE: a hit with queued texture call-outs sets bail instead. */
/* own_minv (LITE callers that hold the hit's chain record, RT_CHAIN_OWN_NORMAL): the whole transform is that one matrix */
/* (inlined: as a function of its own the compiler parks two values in accumulation registers, and every FULL kernel that calls it
   then needs 258 registers -- one wave per SIMD instead of the two its launch bounds ask for; S4's k_shadow ran that way) */
template <bool LITE>
__device__ __forceinline__ void transform_normals(const RtDevScene &S, const DRay &wray, const HitRec &hr, ShadeHit &h, const double *aux, bool &bail, const double *own_minv = 0)
{
    h.t = hr.t; h.n = hr.n; h.np = d3(0, 0, 0); h.perturbed = 0; h.surf = hr.surf; h.mat = hr.mat; h.has_color = 0; h.luster_ov = -1; h.rough_ov = -1;
    if constexpr (LITE) {
        if (hr.ntex) { bail = true; return; }
        if (own_minv) { h.n = covec_mat(h.n, own_minv); return; }
        for (int pa = (int)hr.leaf_affine - 1; pa >= 0; pa = S.affines[pa].link) h.n = covec_mat(h.n, S.affines[pa].minv);
        return;
    }
    int it = 0, nt = hr.ntex;
    int pa;
    for (pa = (int)hr.leaf_affine - 1; pa >= 0; pa = S.affines[pa].link) {
        while (it < nt && S.texnodes[HIT_TEX(hr, it)].pa == pa) {
            const RtTexNode &tn = S.texnodes[HIT_TEX(hr, it)];
            DRay lr = ray_in_frame(S, wray, tn.tree_affine);
            apply_texture(S, tn.tex, ray_at(lr, hr.t), h, aux, S.fractal_omega);
            it++;
        }
        h.n = covec_mat(h.n, S.affines[pa].minv);
        if (h.perturbed) h.np = covec_mat(h.np, S.affines[pa].minv);
    }
    while (it < nt && S.texnodes[HIT_TEX(hr, it)].pa == -1) {
        const RtTexNode &tn = S.texnodes[HIT_TEX(hr, it)];
        DRay lr = ray_in_frame(S, wray, tn.tree_affine);
        apply_texture(S, tn.tex, ray_at(lr, hr.t), h, aux, S.fractal_omega);
        it++;
    }
}

/* ConsistentSurfaceNormal, Shade.cpp:62-86 */
RT_DEV void consistent_normal(const DRay &r, const ShadeHit &h, D3 &vE, D3 &cvN, double &fNE)
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
    if (fNE < 0.0) { cvN = -cvN; fNE = -fNE; }
}

/* FresnelLaw_Real, Shade.cpp:90-113 */
RT_DEV void fresnel_real(double fNE, double n1, double n2, double &F, double &fNT)
{
    double eta = n1 / n2;
    double crit = 1.0 - 1.0 / (eta * eta);
    if (crit < 0.0) crit = 0.0; else crit = sqrt(crit);
    if (fNE < crit) {
        F = 1.0;
    } else {
        fNT = sqrt(1.0 - eta * eta * (1.0 - fNE * fNE));
        if (n1 == n2) F = 0.0;
        else {
            double Rs = (n1 * fNE - n2 * fNT) / (n1 * fNE + n2 * fNT);
            double Rp = (n1 * fNT - n2 * fNE) / (n1 * fNT + n2 * fNE);
            F = 0.5 * (Rs * Rs + Rp * Rp);
        }
    }
}

struct Cx { double re, im; };
RT_DEV Cx cx(double a, double b) { Cx c; c.re = a; c.im = b; return c; }
RT_DEV Cx cadd(Cx a, Cx b) { return cx(a.re + b.re, a.im + b.im); }
RT_DEV Cx csub(Cx a, Cx b) { return cx(a.re - b.re, a.im - b.im); }
RT_DEV Cx cscale(Cx a, double f) { return cx(a.re * f, a.im * f); }
RT_DEV double cabs2(Cx a) { return a.re * a.re + a.im * a.im; }
RT_DEV Cx cdiv(Cx a, Cx c)          /* Complex.h:21-33 */
{
    double den, q;
    if (fabs(c.re) < fabs(c.im)) {
        q = c.re / c.im;
        den = c.re * q + c.im;
        return cx((a.re * q + a.im) / den, (a.im * q - a.re) / den);
    }
    q = c.im / c.re;
    den = c.im * q + c.re;
    return cx((a.im * q + a.re) / den, (a.im - a.re * q) / den);
}
/* FresnelLaw_Complex + Fresnel_Spectral, Shade.cpp:117-153 */
RT_DEVN Spec fresnel_spectral(double fNE, double n1r, const float *nn, const float *kk)
{
    Spec s;
    double F = 0.0;
    for (int i = 0; i < RT_NSPEC; i++) {
        Cx n1 = cx(n1r, 0.0), n2 = cx((double)nn[i], (double)kk[i]);
        Cx eta = cdiv(n1, n2);
        double crit = 1.0 - 1.0 / cabs2(eta);
        if (crit < 0.0) crit = 0.0; else crit = sqrt(crit);
        if (fNE < crit) {
            F = 1.0;
        } else {
            double fNT = sqrt(1.0 - cabs2(eta) * (1.0 - fNE * fNE));
            Cx Rs = cdiv(csub(cscale(n1, fNE), cscale(n2, fNT)), cadd(cscale(n1, fNE), cscale(n2, fNT)));
            Cx Rp = cdiv(csub(cscale(n1, fNT), cscale(n2, fNE)), cadd(cscale(n1, fNT), cscale(n2, fNE)));
            F = 0.5 * (cabs2(Rs) + cabs2(Rp));
            if (n1.re == n2.re && n1.im == n2.im) F = 0.0;
        }
        s.v[i] = (float)F;
    }
    return s;
}

RT_DEV bool spectral_metal(int luster) { return luster >= RT_MAT_COPPER && luster <= RT_MAT_TITANIUM; }

/* BidirectionalReflectance, Shade.cpp:155-234.  hn = the hit's unperturbed normal (SURF_WET) */
__device__ Spec brdf(int rough, int luster, const Spec &color, const D3 &hn, const D3 &vL, const D3 &vE, const D3 &cvN, double F)
{
    const Spec white = spec_const(1.0f);
    double lambert = 1.0 / RT_PI, kS = 1.0, fBRDF;
    int logk = 0;
    D3 nspec = cvN;
    switch (rough) {
    case RT_SURF_MATTE:     return color * lambert;
    case RT_SURF_BURNISHED: logk = 3; break;
    case RT_SURF_SMOOTH:    logk = 5; kS = 1.0; break;
    case RT_SURF_WET:       nspec = normalize(hn);     /* falls through */
    case RT_SURF_SHINY:     logk = 7; kS = 0.60; break;
    case RT_SURF_POLISHED:  logk = 9; kS = 0.35; break;
    default: break;
    }
    if (logk) {
        double fNL = dot(cvN, vL), fNLs = dot(nspec, vL), fNE = dot(nspec, vE);
        D3 vH = normalize(vE + vL);
        double fNH = dot(nspec, vH);
        int kappa = 1;
        double blinn = fNH;
        while (logk > 0) { blinn = blinn * blinn; kappa = kappa + kappa; logk--; }
        blinn = kS * ((double)kappa + 2.0) * blinn / (2.0 * RT_PI);
        double fEH = dot(vE, vH);
        double f1 = fabs(2.0 * fNH * fNE / fEH), f2 = fabs(2.0 * fNH * fNLs / fEH);
        double geo = f1 < f2 ? f1 : f2;
        geo = 1.0 < geo ? 1.0 : geo;
        fBRDF = blinn * geo / (4.0 * fNL * fNE);
    } else
        fBRDF = 0.0;
    switch (luster) {
    case RT_MAT_PLASTIC:     return color * lambert + white * (F * fBRDF);
    case RT_MAT_METAL:       return color * ((1.0 - F) * fBRDF) + white * (F * fBRDF);
    case RT_MAT_TRANSPARENT: return white * (F * fBRDF);
    default:
        if (spectral_metal(luster)) return color * fBRDF;
    }
    return spec_const(0.0f);
}

/* RayReflectance, Shade.cpp:236-256 */
RT_DEV Spec ray_reflectance(int luster, const Spec &color, double F)
{
    const Spec white = spec_const(1.0f);
    switch (luster) {
    case RT_MAT_PLASTIC:     return white * F;
    case RT_MAT_METAL:       return color * (1.0 - F) + white * F;
    case RT_MAT_TRANSPARENT: return white * F;
    default:
        if (spectral_metal(luster)) return color;
    }
    return spec_const(0.0f);
}

/* ColoredGlass / Foggy, TextureMaps.cpp:13-57; Atmosphere: the slot the reference leaves empty (TextureMaps.cpp:162-168),
   filled with Rayleigh extinction and ambient airlight -- DESIGN.md 7b, "Atmosphere" */
RT_DEVN void propagate(int id, const D3 &p0, const D3 &p1, const float *color, double refr, Spec &tr, Spec &ster)
{
    if (id == RT_PROP_ATMOSPHERE) {
        /* Rayleigh() (TextureMaps.cpp:123-135): K = 1e36 * 1e3 * 24 pi^3 * ((n^2-1)/(n^2+2))^2 / N, divided by the wavelength in
           nanometres to the fourth = scattering coefficient per kilometre; number density N from n = sqrt(1 + alpha N)
           (RefractiveIndex, :117-121) with POLARIZABILITY_AIR (:107).  Scene units are metres, as in ColoredGlass. */
        const double n2 = refr * refr;
        double K = 0.0;
        if (n2 > 1.0) {
            const double ll = (n2 - 1.0) / (n2 + 2.0), N = (n2 - 1.0) / 2.133e-29;
            K = 1.0e36 * 1.0e3 * (24.0 * RT_PI * RT_PI * RT_PI) * (ll * ll) / N;
        }
        const double km = norm(p1 - p0) / 1000.0;
        for (int i = 0; i < RT_NSPEC; i++) {
            const double lam = 380.0 + 40.0 * i, l2 = lam * lam;
            double x = (double)color[i] * K / (l2 * l2) * km, T;
            if (x > 87.0) T = 0.0; else T = exp(-x);
            tr.v[i] = (float)T;
            ster.v[i] = (float)(1.0 - T);       /* times the ambient radiance: propagate_in() below */
        }
        return;
    }
    if (id == RT_PROP_FOGGY) {
        double t = norm(p1 - p0);
        for (int i = 0; i < RT_NSPEC; i++) {
            double sc = (double)color[i] * t, T;
            if (sc > 87.0) T = 0.0; else T = exp(-sc);
            tr.v[i] = (float)T;
            ster.v[i] = (float)((1.0 - T) * 0.1);
        }
        return;
    }
    ster = spec_const(0.0f);
    double t = norm(p1 - p0) * 100.0;
    for (int i = 0; i < RT_NSPEC; i++) {
        double mu = (double)color[i] * t, T;
        if (mu > 87.0) T = 0.0; else T = exp(-mu);
        tr.v[i] = (float)T;
    }
}

/* the propagator call of Shade.cpp:333-336, :438-442, :512-516; `scene` is what the slot's RayTracer* argument gives access to:
   Atmosphere scatters the scene's ambient radiance into the path (flux of plAmbient; black without one) */
template <class SCENE>
RT_DEV void propagate_in(const SCENE &S, int id, const D3 &p0, const D3 &p1, const float *color, double refr, Spec &tr, Spec &ster)
{
    propagate(id, p0, p1, color, refr, tr, ster);
    if (id == RT_PROP_ATMOSPHERE) {
#pragma unroll
        for (int i = 0; i < RT_NSPEC; i++) ster.v[i] = S.has_ambient ? (float)((double)ster.v[i] * (double)S.ambient[i]) : 0.0f;
    }
}

/* ReflectionDirection / RefractionDirection, Shade.cpp:361-391 */
RT_DEV D3 reflect_dir(const D3 &cvN, const D3 &vE, const D3 &hn, int perturbed, double fNE)
{
    D3 vR = cvN * 2.0 * fNE - vE;
    if (perturbed && dot(hn, vR) < RT_EPSILON) {
        D3 n2 = normalize(hn);
        double fNE2 = dot(n2, vE);
        vR = n2 * 2.0 * fNE2 - vE;
    }
    return vR;
}
RT_DEV D3 refract_dir(const D3 &cvN, const D3 &vE, const D3 &hn, int perturbed, double eta, double fNE, double fNT)
{
    D3 vT = cvN * (eta * fNE - fNT) - vE * eta;
    if (perturbed && dot(hn, vT) < RT_EPSILON) {
        D3 n2 = normalize(hn);
        double fNE2 = dot(n2, vE);
        vT = n2 * (eta * fNE2 - fNT) - vE * eta;
    }
    return vT;
}

/* Camera::Emit, Cameras.h:8-67 */
__device__ DRay camera_emit(const RtCamera &c, double vx, double vy, double uC, double vC)
{
    DRay r;
    D3 org = d3(c.org[0], c.org[1], c.org[2]);
    double x, y, d;
    if (c.type == RT_CAM_PERSPECTIVE) {
        r.org = org;
        r.dir = d3((vx - 0.5) * 35.0 / c.focal, (vy - 0.5) * 35.0 / c.focal, -1.0);
        return r;
    }
    x = (vx - 0.5) * 35.0 / c.focal;
    y = (vy - 0.5) * 35.0 / c.focal;
    if (c.type == RT_CAM_ORTHOGRAPHIC) {
        r.org = d3(x, y, org.z);
        r.dir = d3(0.0, 0.0, -1.0);
        return r;
    }
    d = x * x + y * y + 4.0;
    r.org = org;
    r.dir = d3(4.0 * x / d, 4.0 * y / d, 1.0 - 8.0 / d);
    if (c.type == RT_CAM_APERTURE) {
        DRay r2;
        r.dir = normalize(r.dir);
        D3 vX = d3(c.vX[0], c.vX[1], c.vX[2]), vY = d3(c.vY[0], c.vY[1], c.vY[2]);
        r2.org = r.org + vX * uC + vY * vC;
        r2.dir = ray_at(r, c.focus_distance) - r2.org;
        return r2;
    }
    return r;
}
