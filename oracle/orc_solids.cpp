/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 * CPU restatement of the reference's ray/solid intersection and CSG hit-list algebra (Solids.cpp).
 */
#include "orc_solids.h"

namespace orc {

Builtins g_b;

/* ------------------------------------------------------------------------------------------------
 * Root finders.  Solids.cpp:277-323 (linear, quadratic), :387-434 (cube root), :436-497 (cubic),
 * :501-561 (quartic, Ferrari + Cardano).
 * ---------------------------------------------------------------------------------------------- */
int solve_linear(double *r, double a, double b)
{
    if (a == 0.0) {
        if (b == 0.0) { r[0] = 0.0; return 1; }
        return 0;
    }
    r[0] = -b / a;
    return 1;
}

int solve_quadratic(double *r, double a, double b, double c)
{
    if (a == 0.0)
        return solve_linear(r, b, c);
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

double cube_root(double x)
{
    static const double c2 = 1.259921049894873164767210607278228350570251, c4 = 1.587401051968199474751705639272308260391493;
    static const double c2i = 0.7937005259840997373758528196361541301957467, c4i = 0.6299605249474365823836053036391141752851257;
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
    x = ldexp(x, e);                       /* Solids.cpp:427 uses ldexpl on a double: exact either way */
    x -= (x - (z / (x * x))) / 3.0;
    x -= (x - (z / (x * x))) / 3.0;
    x -= (x - (z / (x * x))) / 3.0;
    return sign < 0 ? -x : x;
}

int solve_cubic(double *r, double a, double b, double c, double d)
{
    if (a == 0.0)
        return solve_quadratic(r, b, c, d);
    double A = b / a, B = c / a, C = d / a;
    double yi2 = 0.5 * (A * (2.0 * A * A - 9.0 * B) + 27.0 * C);
    double ypi3 = 3.0 * B - A * A;
    double disc = ypi3 * ypi3 * ypi3 + yi2 * yi2;
    double S = 0, S3 = 0, T, U, V, fc, fs;
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
        fs = kSqrt3 * sin(th);
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

int solve_quartic(double *r, double a, double b, double c, double d, double e)
{
    if (a == 0.0)
        return solve_cubic(r, b, c, d, e);
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
        if (h) H = D / h; else H = -0.5 * y + F;
    } else {
        H = -0.5 * y + F;
        h = D / H;
    }
    int n = solve_quadratic(r, 1.0, G, H);
    n += solve_quadratic(r + n, 1.0, g, h);
    for (int j = 1; j < n; j++) {               /* insertion sort, Solids.cpp:554-559 */
        double x = r[j];
        int i;
        for (i = j - 1; i >= 0 && x < r[i]; --i) r[i + 1] = r[i];
        r[i + 1] = x;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * HitWork + HitList: Solids.h:47-58, Solids.cpp:172-227
 * ---------------------------------------------------------------------------------------------- */
struct Work {
    double store_r[RT_MAXROOTS_ORC + 2];
    V3     store_n[RT_MAXROOTS_ORC + 2];
    double *r;          /* r[-1] is addressable */
    V3     *n;
    int     count;
    int     inside_inf;
    double  eps;
    Work() : count(0), inside_inf(0), eps(kEps) { r = store_r + 1; n = store_n + 1; for (int i = 0; i < RT_MAXROOTS_ORC + 2; i++) store_r[i] = 0.0; }
};

static void hit_list(Work &w, Node *leaf, Ctx &cx, HitList &out)
{
    int first, origin_inside;
    for (first = 0; first < w.count; first++)
        if (w.r[first] >= 0.0) break;
    origin_inside = (first & 1) ^ w.inside_inf;
    /* Solids.cpp:188 reads rgfRoots[iFirst] even when iFirst == nHits (indeterminate storage in the
       reference).  Stated policy: "no root left" counts as "> eps". */
    bool far_first = (w.count == 0) || (first >= w.count) || (w.r[first] > w.eps);
    if (origin_inside && far_first) {
        --first;
        w.r[first] = 0.0;
        w.n[first] = V3(0.0, 0.0, 0.0);
    }
    if ((w.count - first) >= 2 && w.r[first] == 0.0 && w.r[first + 1] < w.eps) {
        first += 2;
    } else if ((w.count - first) && w.r[first] < w.eps) {
        w.r[first] = 0.0;
        w.n[first] = V3(0.0, 0.0, 0.0);
    }
    /* the reference builds the list back to front, toggling nRayInsideAtInfinity per hit */
    int nout = w.count - first;
    if (nout <= 0) return;
    out.resize(nout);
    int inside = w.inside_inf;
    for (int i = w.count - 1; i >= first; --i) {
        Hit &h = out[i - first];
        h.t = w.r[i];
        h.n = w.n[i];
        h.np = V3(0, 0, 0);
        h.perturbed = 0;
        h.affine = cx.last_affine;
        h.mat = inside ? &g_b.neutronium : &g_b.vacuum;
        h.surf = (h.t > 0.0) ? &g_b.matte : 0;
        h.leaf = leaf;
        h.tex.clear();
        inside = !inside;
    }
}

/* Liang-Barsky clip of the ray against the half space n.p <= pn: Solids.cpp:232-273 */
static int clip_plane(const Ray &ray, const V3 &nrm, Work &w, int do_clip = 1, double pn = 1.0)
{
    double B, C, t = 0.0;
    if (do_clip && w.count == 0)
        return 0;
    B = dot(nrm, ray.dir);
    C = dot(nrm, ray.org) - pn;
    if (B) t = -C / B;
    if (!do_clip) {
        w.count = 1;
        w.r[0] = 0.0;
        w.inside_inf = 1;
    }
    w.inside_inf &= (B < 0.0);
    if (B < 0.0) {
        if (w.count == 2 && w.r[1] <= t) {
            w.count = 0;
        } else if (w.r[0] < t) {
            w.r[0] = t;
            w.n[0] = nrm;
        }
    } else if (B > 0.0) {
        if (t <= w.r[0]) {
            w.count = 0;
        } else if (w.count == 1 || t < w.r[1]) {
            w.count = 2;
            w.r[1] = t;
            w.n[1] = nrm;
        }
    } else if (C > 0.0)
        w.count = 0;
    return w.count;
}

/* Extent::TestIntersection, Solids.cpp:1533-1572 (ray.tMin is ignored there, and here) */
int extent_test(const Extent &e, const Ray &ray)
{
    double t0 = 0.0, t1 = kInf, tmin = 0, tmax = 0;
    for (int i = 0; i < 3; i++) {
        double B = ray.dir.at(i);
        double cmin = +e.lo.at(i) - ray.org.at(i);
        double cmax = -e.hi.at(i) + ray.org.at(i);
        if (B) { tmin = +cmin / B; tmax = -cmax / B; }
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

static inline double dmax(double a, double b) { return a > b ? a : b; }   /* Affine.h:79-101 */
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline V3 vmax(const V3 &a, const V3 &b) { return V3(dmax(a.x, b.x), dmax(a.y, b.y), dmax(a.z, b.z)); }
static inline V3 vmin(const V3 &a, const V3 &b) { return V3(dmin(a.x, b.x), dmin(a.y, b.y), dmin(a.z, b.z)); }
static Extent cube_extent(double w) { Extent e; e.lo = V3(-w, -w, -w); e.hi = V3(w, w, w); return e; }

/* virtual GetExtent(): Solids.h:69,143,210,289,496; Solids.cpp:360-371,641-645,783-796,922-947,1077-1249,1580-1584 */
Extent node_extent(const Node *n)
{
    Extent e, l, r;
    if (!n) return cube_extent(0.0);
    switch (n->type) {
    case RT_NODE_QUADRIC:
    case RT_NODE_POLYHEDRON:
        return cube_extent(1.0);
    case RT_NODE_CONEHYPERCYL: {
        double rr = dmax(1.0, n->Rneck), z;
        if (n->Rneck <= 1.0) z = 1.0; else z = sqrt(n->Rneck * n->Rneck / (n->Rneck * n->Rneck - 1.0));
        e.lo = V3(-rr, -rr, -z); e.hi = V3(rr, rr, +z);
        return e;
    }
    case RT_NODE_SLAB:
        e.lo = V3(-kInf, -kInf, -1.0); e.hi = V3(kInf, kInf, 1.0);
        return e;
    case RT_NODE_TORUS:
        e.lo = V3(-1.0 - n->minor, -1.0 - n->minor, -n->minor);
        e.hi = V3(1.0 + n->minor, 1.0 + n->minor, n->minor);
        return e;
    case RT_NODE_MESH:
        return n->ex;
    case RT_NODE_AFFINE: {
        if (!n->left) return cube_extent(0.0);
        e = node_extent(n->left);
        V3 c[2] = { e.lo, e.hi };
        for (int i = 0; i < 8; i++) {
            V3 p(c[(i >> 0) & 1].x, c[(i >> 1) & 1].y, c[(i >> 2) & 1].z);
            V3 q = mul(n->m, p) + n->tr;
            if (i == 0) e.lo = e.hi = q;
            else { e.lo = vmin(e.lo, q); e.hi = vmax(e.hi, q); }
        }
        return e;
    }
    case RT_NODE_UNION:
    case RT_NODE_TOPUNION:
        if (n->left) l = node_extent(n->left);
        if (n->right) r = node_extent(n->right);
        if (n->left) {
            if (n->right) { e.lo = vmin(l.lo, r.lo); e.hi = vmax(l.hi, r.hi); return e; }
            return l;
        }
        if (n->right) return r;
        return cube_extent(0.0);
    case RT_NODE_INTERSECTION:
        if (n->left) l = node_extent(n->left);
        if (n->right) r = node_extent(n->right);
        if (n->left && n->right) { e.lo = vmax(l.lo, r.lo); e.hi = vmin(l.hi, r.hi); return e; }
        return cube_extent(0.0);
    case RT_NODE_DIFFERENCE:
        if (n->left) return node_extent(n->left);
        return cube_extent(0.0);
    case RT_NODE_ANGULARCLIP:
    case RT_NODE_MATERIAL:
    case RT_NODE_SURFACE:
        return node_extent(n->left);
    case RT_NODE_BOUNDINGBOX:
        return n->ex;
    default:    /* Solid::GetExtent, NullSolid */
        e.lo = V3(0, 0, 0); e.hi = V3(0, 0, 0);
        return e;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Primitive intersections
 * ---------------------------------------------------------------------------------------------- */
static void quadric_hits(Node *q, const Ray &ray, Ctx &cx, HitList &out)     /* Solids.cpp:325-358 */
{
    Work w;
    const V3 &d = ray.dir, &o = ray.org;
    double A = d.x * d.x + d.y * d.y + d.z * d.z * q->P;
    double B = 2.0 * (d.x * o.x + d.y * o.y + d.z * o.z * q->P) + d.z * q->Q;
    double C = o.x * o.x + o.y * o.y + o.z * o.z * q->P + o.z * q->Q + q->R;
    w.count = solve_quadratic(w.r, A, B, C);
    for (int i = 0; i < w.count; i++) {
        V3 p = at(ray, w.r[i]);
        w.n[i].x = p.x;
        w.n[i].y = p.y;
        w.n[i].z = p.z * q->P + 0.5 * q->Q;
    }
    if (q->i0 == 1) {
        clip_plane(ray, V3(0.0, 0.0, 1.0), w);
        clip_plane(ray, V3(0.0, 0.0, -1.0), w);
    }
    if (q->i0 == -1) { A = -A; B = -B; C = -C; }
    w.inside_inf = (A < 0.0);
    hit_list(w, q, cx, out);
}

static void polyhedron_hits(Node *p, const Ray &ray, Ctx &cx, HitList &out)  /* Solids.cpp:769-781 */
{
    Work w;
    int n = 0;
    w.inside_inf = 0;
    int sides = (int)p->planes.size();
    if (sides) n = clip_plane(ray, p->planes[0], w, 0);
    for (int i = 1; n && i < sides; i++)
        n = clip_plane(ray, p->planes[i], w);
    hit_list(w, p, cx, out);
}

static void slab_hits(Node *s, const Ray &ray, Ctx &cx, HitList &out)        /* Solids.cpp:859-869 */
{
    Work w;
    int n = clip_plane(ray, V3(0.0, 0.0, 1.0), w, 0);
    if (n) n = clip_plane(ray, V3(0.0, 0.0, -1.0), w);
    hit_list(w, s, cx, out);
}

/* Triangle mesh as a Solid (policy Q14).  The reference has the pieces -- vertex and triangle arrays read
   from an OBJ file (ObjectFormat.cpp:12-75) and a Moeller-Trumbore test (:79-102) -- but never a Solid made of them.
   Stated behaviour: every triangle is tested (no hierarchy here: the device's BVH may only skip triangles the ray
   misses), with the reference's test put right in two places (second barycentric bound u + v > 1; t = f * Dot(e2, q))
   and the parallel test made scale free; crossings with t >= 0 are kept, the RT_MAXROOTS_ORC nearest of them in
   (t, triangle index) order; an odd number of crossings ahead means the origin is inside, which HitList learns from
   one root behind the origin; normals are Cross(e1, e2), transformed late like any other.
   PARITY UNPINNED by the reference: it holds no fixture, test or output for ray-traced meshes.  What
   pins this routine is indirect: a cube of twelve triangles must give Cube's hit lists and a fine sphere mesh must
   converge to Sphere (tests/test_cpu_oracle.py). */
static void mesh_hits(Node *mn, const Ray &ray, Ctx &cx, HitList &out)
{
    out.clear();
    double ht[RT_MAXROOTS_ORC];
    int    hi[RT_MAXROOTS_ORC];
    int    nh = 0;
    const int nt = (int)(mn->mtris.size() / 3);
    for (int k = 0; k < nt; k++) {
        const V3 &v0 = mn->mverts[mn->mtris[3 * k]];
        const V3 e1 = mn->mverts[mn->mtris[3 * k + 1]] - v0;
        const V3 e2 = mn->mverts[mn->mtris[3 * k + 2]] - v0;
        const V3 h = cross(ray.dir, e2);
        const double a = dot(e1, h);
        const V3 nn = cross(e1, e2);
        if (a * a < (kEps * kEps) * (dot(nn, nn) * dot(ray.dir, ray.dir))) continue;
        const double f = 1.0 / a;
        const V3 s = ray.org - v0;
        const double u = f * dot(s, h);
        if (u < 0.0 || u > 1.0) continue;
        const V3 q = cross(s, e1);
        const double v = f * dot(ray.dir, q);
        if (v < 0.0 || u + v > 1.0) continue;
        const double t = f * dot(e2, q);
        if (!(t >= 0.0)) continue;
        int j = nh;
        if (nh == RT_MAXROOTS_ORC) {
            const int l = nh - 1;
            if (t > ht[l] || (t == ht[l] && k > hi[l])) continue;
            j = l;
        } else nh++;
        for (; j > 0 && (t < ht[j - 1] || (t == ht[j - 1] && k < hi[j - 1])); --j) { ht[j] = ht[j - 1]; hi[j] = hi[j - 1]; }
        ht[j] = t; hi[j] = k;
    }
    if (nh == 0) return;
    Work w;
    const int behind = nh & 1;
    if (behind) { w.r[0] = -1.0; w.n[0] = V3(0.0, 0.0, 0.0); }
    for (int i = 0; i < nh; i++) {
        const int k = hi[i];
        const V3 &v0 = mn->mverts[mn->mtris[3 * k]];
        w.r[behind + i] = ht[i];
        w.n[behind + i] = cross(mn->mverts[mn->mtris[3 * k + 1]] - v0, mn->mverts[mn->mtris[3 * k + 2]] - v0);
    }
    w.count = nh + behind;
    hit_list(w, mn, cx, out);
}

static void torus_hits(Node *tn, const Ray &ray, Ctx &cx, HitList &out)      /* Solids.cpp:566-639 */
{
    Work w;
    double rm = tn->minor;
    if (!extent_test(node_extent(tn), ray))
        return;
    double fT = norm(ray.org - V3(0.0, 0.0, 0.0)) - 1.0 - rm - 0.1;
    double fS = norm(ray.dir);
    V3 d = ray.dir, o = ray.org;
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
    w.count = solve_quartic(w.r, a, b, c, dd, e);
    double db = 4.0 * a, dc = 3.0 * b, ddd = 2.0 * c, de = dd;
    for (int i = 0; i < w.count; i++) {
        double x = w.r[i];
        x = x - (e + x * (dd + x * (c + x * (b + x * a)))) / (de + x * (ddd + x * (dc + x * db)));
        x = x - (e + x * (dd + x * (c + x * (b + x * a)))) / (de + x * (ddd + x * (dc + x * db)));
        w.r[i] = x;
    }
    for (int i = 0; i < w.count; i++) {
        double x = o.x + d.x * w.r[i], y = o.y + d.y * w.r[i], z = o.z + d.z * w.r[i];
        double ax = sqrt(x * x + y * y);
        w.n[i] = V3(x - x / ax, y - y / ax, z);
    }
    if (fT >= 0.0) {
        for (int i = 0; i < w.count; i++)
            w.r[i] = (w.r[i] + fT) / fS;
    }
    w.inside_inf = 0;
    w.eps *= 10.0;
    hit_list(w, tn, cx, out);
}

/* ------------------------------------------------------------------------------------------------
 * CSG: lattice algebra on material density.  Solids.cpp:987-1058
 * ---------------------------------------------------------------------------------------------- */
typedef Node *(*MatOp)(Node *, Node *);
static Node *op_max(Node *l, Node *r) { return l->density > r->density ? l : r; }
static Node *op_min(Node *l, Node *r) { return l->density < r->density ? l : r; }
static Node *op_diff(Node *l, Node *r) { return r->density <= 0.0 ? l : &g_b.vacuum; }

static void merge(HitList &L, HitList &R, MatOp op, HitList &out)
{
    size_t il = 0, ir = 0;
    Node *ml = &g_b.vacuum, *mr = &g_b.vacuum, *prev = &g_b.vacuum;
    out.clear();
    while (il < L.size() || ir < R.size()) {
        if (ir >= R.size() || (il < L.size() && L[il].t <= R[ir].t)) {
            if (ir < R.size() && L[il].t == R[ir].t) {
                mr = R[ir].mat;
                ir++;
            }
            ml = L[il].mat;
            Node *m = op(ml, mr);
            if (m->density != prev->density) {
                prev = m;
                out.push_back(L[il]);
                out.back().mat = m;
            }
            il++;
        } else {
            mr = R[ir].mat;
            Node *m = op(ml, mr);
            if (m->density != prev->density) {
                prev = m;
                out.push_back(R[ir]);
                out.back().mat = m;
            }
            ir++;
        }
    }
    if (out.size() == 1 && out[0].t == 0.0 && out[0].mat->density == 0.0)   /* Solids.cpp:1053-1056 */
        out.clear();
}

/* virtual Intersect() */
void node_hits(Node *n, const Ray &ray, Ctx &cx, HitList &out)
{
    out.clear();
    if (!n) return;
    switch (n->type) {
    case RT_NODE_QUADRIC:
    case RT_NODE_CONEHYPERCYL: quadric_hits(n, ray, cx, out); return;
    case RT_NODE_POLYHEDRON:   polyhedron_hits(n, ray, cx, out); return;
    case RT_NODE_SLAB:         slab_hits(n, ray, cx, out); return;
    case RT_NODE_TORUS:        torus_hits(n, ray, cx, out); return;
    case RT_NODE_MESH:         mesh_hits(n, ray, cx, out); return;
    case RT_NODE_AFFINE: {                                              /* Solids.cpp:899-913 */
        Ray local;
        local.dir = mul(n->mi, ray.dir);
        local.org = mul(n->mi, ray.org - n->tr);
        cx.last_affine = n;
        node_hits(n->left, local, cx, out);
        return;
    }
    case RT_NODE_UNION:
    case RT_NODE_TOPUNION: {                                            /* Solids.cpp:1060-1075, 1103-1131 */
        /* TopUnion's tMin bookkeeping and per-primary-ray child swap change no hit list
           (Extent::TestIntersection ignores tMin); it is evaluated as a Union. */
        HitList l, r;
        if (n->left) node_hits(n->left, ray, cx, l);
        if (n->right) node_hits(n->right, ray, cx, r);
        if (!l.empty()) {
            if (!r.empty()) { merge(l, r, op_max, out); return; }
            out.swap(l);
            return;
        }
        out.swap(r);
        return;
    }
    case RT_NODE_INTERSECTION: {                                        /* Solids.cpp:1139-1158 */
        HitList l, r;
        if (!n->left) return;
        node_hits(n->left, ray, cx, l);
        if (!n->right) return;
        node_hits(n->right, ray, cx, r);
        if (!l.empty() && !r.empty()) merge(l, r, op_min, out);
        return;
    }
    case RT_NODE_DIFFERENCE: {                                          /* Solids.cpp:1177-1196 */
        HitList l, r;
        if (!n->left) return;
        node_hits(n->left, ray, cx, l);
        if (!n->right) { out.swap(l); return; }
        node_hits(n->right, ray, cx, r);
        if (!l.empty()) {
            if (!r.empty()) { merge(l, r, op_diff, out); return; }
            out.swap(l);
        }
        return;
    }
    case RT_NODE_ANGULARCLIP: {                                         /* Solids.cpp:1224-1241 */
        HitList body, wedge;
        node_hits(n->left, ray, cx, body);
        if (!body.empty()) {
            Work w;
            clip_plane(ray, n->cv1, w, 0, 0.0);
            clip_plane(ray, n->cv2, w, 1, 0.0);
            hit_list(w, n, cx, wedge);
            merge(body, wedge, n->i0 ? op_min : op_diff, out);
        }
        return;
    }
    case RT_NODE_BOUNDINGBOX:                                           /* Solids.h:495 */
        if (extent_test(n->ex, ray)) node_hits(n->left, ray, cx, out);
        return;
    case RT_NODE_MATERIAL: {                                            /* Solids.cpp:1466-1487 */
        node_hits(n->left, ray, cx, out);
        for (size_t i = 0; i < out.size(); i++) {
            Hit &h = out[i];
            if (h.mat->density > 0.0) {
                h.mat = n;
                if (n->i1) { TexRef t; t.p = at(ray, h.t); t.tex = n->i1; t.pa = n->link; t.images = n->images; h.tex.push_back(t); }
            }
        }
        return;
    }
    case RT_NODE_SURFACE: {                                             /* Solids.cpp:1498-1519 */
        node_hits(n->left, ray, cx, out);
        for (size_t i = 0; i < out.size(); i++) {
            Hit &h = out[i];
            if (h.t == 0.0) continue;
            h.surf = n;
            if (n->i1) { TexRef t; t.p = at(ray, h.t); t.tex = n->i1; t.pa = n->link; t.images = n->images; h.tex.push_back(t); }
        }
        return;
    }
    default:
        return;     /* NullSolid */
    }
}

/* virtual Optimize(): Solids.cpp:957-983 (affine chain merge + parent links), :1489-1528, Solids.h:378-418,497 */
Node *optimize(Node *n, Node *enclosing)
{
    if (!n) return 0;
    switch (n->type) {
    case RT_NODE_AFFINE:
        if (!n->left) return 0;
        while (n->left->type == RT_NODE_AFFINE) {
            Node *c = n->left;
            M3 m = mul(n->m, c->m);
            M3 mi = mul(c->mi, n->mi);
            V3 v = n->tr + mul(n->m, c->tr);
            n->m = m; n->mi = mi; n->tr = v;
            n->left = c->left;
            if (!n->left) return n;     /* the reference would dereference null here */
        }
        n->link = enclosing;
        n->left = optimize(n->left, n);
        return n;
    case RT_NODE_UNION: case RT_NODE_TOPUNION: case RT_NODE_INTERSECTION: case RT_NODE_DIFFERENCE:
        if (n->left) n->left = optimize(n->left, enclosing);
        if (n->right) n->right = optimize(n->right, enclosing);
        return n;
    case RT_NODE_ANGULARCLIP:
        if (n->left) n->left = optimize(n->left, enclosing);
        return n;
    case RT_NODE_BOUNDINGBOX:
        n->left = optimize(n->left, enclosing);
        return n;
    case RT_NODE_MATERIAL: case RT_NODE_SURFACE:
        n->link = enclosing;
        n->left = optimize(n->left, enclosing);
        return n;
    default:
        return n;
    }
}

} // namespace orc

namespace orc {
/* Q13 (orc_solids.h): private material copies for hits whose colour / luster a solid texture sets */
Node *Ctx::clone_material(const Node *m) const
{
    Node *c = new Node(*m);
    scratch.push_back(c);
    return c;
}
void Ctx::drop_scratch() const
{
    for (size_t i = 0; i < scratch.size(); i++) delete scratch[i];
    scratch.clear();
}
} // namespace orc
