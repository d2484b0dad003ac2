/*
 * rt_flatten.cpp -- turn the optimised object graph (RtSceneDesc) into the per-ray program and the
 * SoA tables the kernels read (rt_device.h).  Host only; runs once per scene upload.
 *
 * Culling boxes are this build's own CONSERVATIVE bounds, not the reference's GetExtent() (which
 * returns the unit cube for every quadric and polyhedron, Solids.h:69, Solids.cpp:789): a box may
 * only be used to skip a subtree if missing it implies the reference's hit list is empty.  The
 * reference's own BoundingBox nodes keep their (possibly too small) extents and their semantics
 * (OP_REFBOX), because dropping hits outside them is visible in the reference's images.
 */
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include "rt_flatten.h"
#include "../../include/rt_amd.h"

namespace {

const double kHuge = 1.0e300;

struct Box {
    double lo[3], hi[3];
    bool empty;
    Box() : empty(true) { for (int i = 0; i < 3; i++) { lo[i] = kHuge; hi[i] = -kHuge; } }
    static Box infinite() { Box b; b.empty = false; for (int i = 0; i < 3; i++) { b.lo[i] = -kHuge; b.hi[i] = kHuge; } return b; }
    static Box cube(double lx, double ly, double lz, double hx, double hy, double hz)
    { Box b; b.empty = false; b.lo[0] = lx; b.lo[1] = ly; b.lo[2] = lz; b.hi[0] = hx; b.hi[1] = hy; b.hi[2] = hz; return b; }
    bool finite() const { if (empty) return true; for (int i = 0; i < 3; i++) if (lo[i] < -1e200 || hi[i] > 1e200) return false; return true; }
};

Box box_union(const Box &a, const Box &b)
{
    if (a.empty) return b;
    if (b.empty) return a;
    Box r; r.empty = false;
    for (int i = 0; i < 3; i++) { r.lo[i] = fmin(a.lo[i], b.lo[i]); r.hi[i] = fmax(a.hi[i], b.hi[i]); }
    return r;
}
Box box_intersect(const Box &a, const Box &b)
{
    if (a.empty || b.empty) return Box();
    Box r; r.empty = false;
    for (int i = 0; i < 3; i++) { r.lo[i] = fmax(a.lo[i], b.lo[i]); r.hi[i] = fmin(a.hi[i], b.hi[i]); if (r.lo[i] > r.hi[i]) return Box(); }
    return r;
}
/* image of a box under p -> M p + T (forward transform = inverse of minv); any infinite side makes it infinite */
Box box_transform(const Box &b, const double *m, const double *t)
{
    if (b.empty) return b;
    if (!b.finite()) return Box::infinite();
    Box r; r.empty = false;
    for (int i = 0; i < 3; i++) { r.lo[i] = kHuge; r.hi[i] = -kHuge; }
    for (int c = 0; c < 8; c++) {
        double p[3] = { (c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1], (c & 4) ? b.hi[2] : b.lo[2] };
        for (int i = 0; i < 3; i++) {
            double q = m[3 * i] * p[0] + m[3 * i + 1] * p[1] + m[3 * i + 2] * p[2] + t[i];
            r.lo[i] = fmin(r.lo[i], q); r.hi[i] = fmax(r.hi[i], q);
        }
    }
    return r;
}

/* solid { x^2 + y^2 + P z^2 + Q z + R <= 0 }, optionally capped to |z| <= 1 */
Box quadric_bound(double P, double Q, double R, int clip)
{
    if (clip == -1) return Box::infinite();
    double z0, z1;
    if (clip == 1) { z0 = -1.0; z1 = 1.0; }
    else {
        if (!(P > 0.0)) return Box::infinite();
        double disc = Q * Q - 4.0 * P * R;
        if (disc < 0.0) return Box();       /* no real points */
        double s = sqrt(disc);
        z0 = (-Q - s) / (2.0 * P); z1 = (-Q + s) / (2.0 * P);
    }
    /* r^2(z) = -(P z^2 + Q z + R), maximised over [z0,z1] */
    double zs[3] = { z0, z1, z0 }; int n = 2;
    if (P != 0.0) { double zv = -Q / (2.0 * P); if (zv > z0 && zv < z1) { zs[2] = zv; n = 3; } }
    double r2 = 0.0;
    for (int i = 0; i < n; i++) r2 = fmax(r2, -(P * zs[i] * zs[i] + Q * zs[i] + R));
    if (!(r2 < 1e200)) return Box::infinite();
    double r = sqrt(r2);
    return Box::cube(-r, -r, z0, r, r, z1);
}

/* convex polyhedron { n_i . p <= 1 } */
Box polyhedron_bound(const double *pl, int n)
{
    if (n < 4) return Box::infinite();
    /* bounded iff every direction has a plane facing it; tested on a dense direction set */
    const int ND = 4000;
    for (int k = 0; k < ND; k++) {
        double z = 1.0 - 2.0 * (k + 0.5) / ND, r = sqrt(fmax(0.0, 1.0 - z * z)), a = k * 2.399963229728653;
        double d[3] = { r * cos(a), r * sin(a), z };
        double best = -1.0;
        for (int i = 0; i < n; i++) best = fmax(best, pl[3 * i] * d[0] + pl[3 * i + 1] * d[1] + pl[3 * i + 2] * d[2]);
        if (best < 1e-6) return Box::infinite();
    }
    Box b;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            for (int k = j + 1; k < n; k++) {
                const double *a = pl + 3 * i, *bb = pl + 3 * j, *c = pl + 3 * k;
                double det = a[0] * (bb[1] * c[2] - bb[2] * c[1]) - a[1] * (bb[0] * c[2] - bb[2] * c[0]) + a[2] * (bb[0] * c[1] - bb[1] * c[0]);
                if (fabs(det) < 1e-12) continue;
                /* a.p = b.p = c.p = 1  =>  p = (b x c + c x a + a x b) / det */
                double x = ((bb[1] * c[2] - bb[2] * c[1]) + (c[1] * a[2] - c[2] * a[1]) + (a[1] * bb[2] - a[2] * bb[1])) / det;
                double y = ((bb[2] * c[0] - bb[0] * c[2]) + (c[2] * a[0] - c[0] * a[2]) + (a[2] * bb[0] - a[0] * bb[2])) / det;
                double z = ((bb[0] * c[1] - bb[1] * c[0]) + (c[0] * a[1] - c[1] * a[0]) + (a[0] * bb[1] - a[1] * bb[0])) / det;
                bool in = true;
                for (int q = 0; q < n && in; q++) if (pl[3 * q] * x + pl[3 * q + 1] * y + pl[3 * q + 2] * z > 1.0 + 1e-7) in = false;
                if (!in) continue;
                Box p = Box::cube(x, y, z, x, y, z);
                b = box_union(b, p);
            }
    if (b.empty) return Box::infinite();    /* should not happen for a bounded solid: stay conservative */
    return b;
}

struct Flattener {
    const RtSceneDesc *d;
    RtFlatScene &o;
    std::string &err;
    int rc;
    bool with_bounds;
    int next_uid;
    int last_affine;                /* static image of s_paLastHitAffine (Solids.cpp:170) */
    std::vector<int> node_affine;   /* desc node index -> affine table index (for link resolution) */
    std::vector<int> affine_stack;  /* enclosing affine table indices */
    struct Fwd { double m[9], t[3]; };
    std::vector<Fwd> fwd_stack;     /* forward transforms of enclosing affines (for world boxes) */
    int tex_depth;
    int region_depth;

    Flattener(const RtSceneDesc *dd, RtFlatScene &oo, std::string &e, bool wb)
        : d(dd), o(oo), err(e), rc(0), with_bounds(wb), next_uid(1), last_affine(-1), tex_depth(0), region_depth(0) {}

    bool fail(int code, const std::string &m) { if (!rc) { rc = code; err = m; } return false; }
    const RtNode &N(int i) const { return d->nodes[i]; }
    const double *D(int i) const { return d->data + d->nodes[i].data; }

    /* conservative bound of node i in ITS OWN frame */
    Box local_bound(int i)
    {
        if (i < 0) return Box();
        const RtNode &n = N(i);
        const double *p = D(i);
        switch (n.type) {
        case RT_NODE_QUADRIC: case RT_NODE_CONEHYPERCYL: return quadric_bound(p[0], p[1], p[2], n.i0);
        case RT_NODE_POLYHEDRON: return polyhedron_bound(p, n.i0);
        case RT_NODE_SLAB: return Box::cube(-kHuge, -kHuge, -1.0, kHuge, kHuge, 1.0);
        case RT_NODE_TORUS: return Box::cube(-1.0 - p[0], -1.0 - p[0], -p[0], 1.0 + p[0], 1.0 + p[0], p[0]);
        case RT_NODE_MESH: {
            if (n.i0 < 1 || n.i1 < 1) return Box();
            double lo[3] = { p[0], p[1], p[2] }, hi[3] = { p[0], p[1], p[2] };
            for (int v = 1; v < n.i0; v++)
                for (int k = 0; k < 3; k++) { double x = p[3 * v + k]; if (x < lo[k]) lo[k] = x; if (x > hi[k]) hi[k] = x; }
            return Box::cube(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
        }
        case RT_NODE_AFFINE: return box_transform(local_bound(n.left), p, p + 18);
        case RT_NODE_UNION: case RT_NODE_TOPUNION: return box_union(local_bound(n.left), local_bound(n.right));
        case RT_NODE_INTERSECTION: return box_intersect(local_bound(n.left), local_bound(n.right));
        case RT_NODE_DIFFERENCE: return local_bound(n.left);
        case RT_NODE_ANGULARCLIP: case RT_NODE_BOUNDINGBOX: case RT_NODE_MATERIAL: case RT_NODE_SURFACE: return local_bound(n.left);
        default: return Box();
        }
    }
    Box to_world(Box b)
    {
        for (int k = (int)fwd_stack.size() - 1; k >= 0 && !b.empty; --k) b = box_transform(b, fwd_stack[k].m, fwd_stack[k].t);
        return b;
    }
    int add_box(const Box &b, bool pad)
    {
        RtBox r;
        for (int i = 0; i < 3; i++) {
            double e = pad ? 1e-6 * (fabs(b.hi[i] - b.lo[i]) + fabs(b.lo[i]) + fabs(b.hi[i])) + 1e-9 : 0.0;
            r.lo[i] = b.lo[i] - e; r.hi[i] = b.hi[i] + e;
        }
        o.boxes.push_back(r);
        return (int)o.boxes.size() - 1;
    }
    int push(int op, int a, int b, int tag)
    {
        RtInstr in; in.op = op; in.a = a; in.b = b; in.tag = tag;
        o.prog.push_back(in);
        return (int)o.prog.size() - 1;
    }
    bool can_hit(int i)     /* false when the subtree is statically empty */
    {
        if (i < 0) return false;
        const RtNode &n = N(i);
        switch (n.type) {
        case RT_NODE_NULL: return false;
        case RT_NODE_AFFINE: case RT_NODE_ANGULARCLIP: case RT_NODE_BOUNDINGBOX: case RT_NODE_MATERIAL: case RT_NODE_SURFACE: return can_hit(n.left);
        case RT_NODE_UNION: case RT_NODE_TOPUNION: return can_hit(n.left) || can_hit(n.right);
        case RT_NODE_INTERSECTION: return can_hit(n.left) && can_hit(n.right);
        case RT_NODE_DIFFERENCE: return can_hit(n.left);
        default: return true;
        }
    }
    int leaf(int type_op, int i)
    {
        const RtNode &n = N(i);
        const double *p = D(i);
        RtPrim pr;
        memset(&pr, 0, sizeof pr);
        pr.id = n.id; pr.leaf_affine = last_affine; pr.i0 = n.i0; pr.plane0 = -1;
        int enclosing = affine_stack.empty() ? -1 : affine_stack.back();
        if (last_affine != enclosing) o.irregular_leaves++;
        if (type_op == OP_QUADRIC) { pr.p[0] = p[0]; pr.p[1] = p[1]; pr.p[2] = p[2]; }
        else if (type_op == OP_TORUS) pr.p[0] = p[0];
        else if (type_op == OP_POLY) {
            pr.plane0 = (int)o.planes.size() / 3;
            o.planes.insert(o.planes.end(), p, p + 3 * n.i0);
            if (n.i0 < 1) fail(RT_ERR_INVALID, "polyhedron without planes");
        }
        else if (type_op == OP_MESH) pr.plane0 = build_mesh(i);
        o.prims.push_back(pr);
        return (int)o.prims.size() - 1;
    }
    /* 4-wide BVH over the triangles of mesh node i: the range is halved at the median of the centroids along its
       widest axis, and each half again, which gives the four children of a node */
    struct MeshBuild {
        const double *v; const double *f;
        std::vector<int> order;
        std::vector<float> cen;         /* 3 per triangle */
        std::vector<RtMeshNode> *nodes;
        size_t node0;
        void tri_box(int t, double lo[3], double hi[3]) const
        {
            for (int k = 0; k < 3; k++) { lo[k] = 1e300; hi[k] = -1e300; }
            for (int c = 0; c < 3; c++) {
                const double *q = v + 3 * (size_t)f[3 * (size_t)t + c];
                for (int k = 0; k < 3; k++) { if (q[k] < lo[k]) lo[k] = q[k]; if (q[k] > hi[k]) hi[k] = q[k]; }
            }
        }
        void split(int first, int count, int &half)
        {
            float clo[3] = { 3e38f, 3e38f, 3e38f }, chi[3] = { -3e38f, -3e38f, -3e38f };
            for (int q = first; q < first + count; q++)
                for (int k = 0; k < 3; k++) {
                    float c = cen[3 * (size_t)order[q] + k];
                    if (c < clo[k]) clo[k] = c;
                    if (c > chi[k]) chi[k] = c;
                }
            int ax = 0;
            if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
            if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
            half = count / 2;
            const std::vector<float> &C = cen;
            std::nth_element(order.begin() + first, order.begin() + first + half, order.begin() + first + count,
                             [&C, ax](int x, int y) { float cx = C[3 * (size_t)x + ax], cy = C[3 * (size_t)y + ax]; return cx < cy || (cx == cy && x < y); });
        }
        void range_box(int first, int count, float flo[3], float fhi[3]) const
        {
            double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
            for (int q = first; q < first + count; q++) {
                double a[3], b[3];
                tri_box(order[q], a, b);
                for (int k = 0; k < 3; k++) { if (a[k] < lo[k]) lo[k] = a[k]; if (b[k] > hi[k]) hi[k] = b[k]; }
            }
            for (int k = 0; k < 3; k++) {       /* outward: a relative pad, then the float on the far side */
                double e = 1e-6 * (hi[k] - lo[k]) + 1e-7 * (fabs(lo[k]) + fabs(hi[k])) + 1e-12;
                float l = (float)(lo[k] - e), h = (float)(hi[k] + e);
                if ((double)l > lo[k] - e) l = nextafterf(l, -INFINITY);
                if ((double)h < hi[k] + e) h = nextafterf(h, INFINITY);
                flo[k] = l; fhi[k] = h;
            }
        }
        /* the node's grid and the children's 8-bit corners on it: every decoded corner (evaluated as the device does, in
           double) lies outside the child's float box */
        static void quantize(RtMeshNode &nd, int nr, const float (*blo)[3], const float (*bhi)[3])
        {
            for (int a = 0; a < 3; a++) {
                double ulo = 1e300, uhi = -1e300;
                for (int k = 0; k < nr; k++) { if (blo[k][a] < ulo) ulo = blo[k][a]; if (bhi[k][a] > uhi) uhi = bhi[k][a]; }
                float org = (float)ulo;
                if ((double)org > ulo) org = nextafterf(org, -INFINITY);
                float sc = (float)((uhi - (double)org) / 255.0);
                if (!(sc > 0.0f)) sc = 1e-30f;
                while ((double)org + 255.0 * (double)sc < uhi) sc = nextafterf(sc, INFINITY);
                nd.org[a] = org; nd.scale[a] = sc;
                for (int k = 0; k < 4; k++) {
                    if (k >= nr) { nd.qlo[a][k] = 255; nd.qhi[a][k] = 0; continue; }       /* unused slot: an inverted box */
                    int ql = (int)floor(((double)blo[k][a] - (double)org) / (double)sc), qh = (int)ceil(((double)bhi[k][a] - (double)org) / (double)sc);
                    ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                    while (ql > 0 && (double)org + (double)ql * (double)sc > (double)blo[k][a]) ql--;
                    while (qh < 255 && (double)org + (double)qh * (double)sc < (double)bhi[k][a]) qh++;
                    nd.qlo[a][k] = (uint8_t)ql; nd.qhi[a][k] = (uint8_t)qh;
                }
            }
        }
        int build(int first, int count)
        {
            const int me = (int)(nodes->size() - node0);
            nodes->push_back(RtMeshNode());
            RtMeshNode nd;
            memset(&nd, 0, sizeof nd);
            for (int k = 0; k < 4; k++) nd.child[k] = RT_MESH_EMPTY;
            int r0[4], rc[4], nr = 0;
            if (count <= RT_MESH_LEAF) { r0[0] = first; rc[0] = count; nr = 1; }
            else {
                int h;
                split(first, count, h);
                const int part0[2] = { first, first + h }, partc[2] = { h, count - h };
                for (int s = 0; s < 2; s++) {
                    if (partc[s] <= RT_MESH_LEAF) { r0[nr] = part0[s]; rc[nr] = partc[s]; nr++; }
                    else {
                        int h2;
                        split(part0[s], partc[s], h2);
                        r0[nr] = part0[s]; rc[nr] = h2; nr++;
                        r0[nr] = part0[s] + h2; rc[nr] = partc[s] - h2; nr++;
                    }
                }
            }
            float blo[4][3], bhi[4][3];
            for (int k = 0; k < nr; k++) {
                range_box(r0[k], rc[k], blo[k], bhi[k]);
                if (rc[k] <= RT_MESH_LEAF) nd.child[k] = ~(r0[k] | ((rc[k] - 1) << 27));
                else nd.child[k] = build(r0[k], rc[k]);
            }
            quantize(nd, nr, blo, bhi);
            (*nodes)[node0 + me] = nd;
            return me;
        }
    };
    int build_mesh(int i)
    {
        const RtNode &n = N(i);
        const double *p = D(i);
        const int nv = n.i0, nt = n.i1;
        if (nt >= (1 << 27)) { fail(RT_ERR_UNSUPPORTED, "mesh of 2^27 triangles or more"); return -1; }
        if (nv < 3 || nt < 1 || n.ndata != 3 * nv + 3 * nt) { fail(RT_ERR_INVALID, "mesh node with inconsistent sizes"); return -1; }
        const double *f = p + 3 * (size_t)nv;
        for (size_t k = 0; k < 3 * (size_t)nt; k++)
            if (!(f[k] >= 0.0 && f[k] < (double)nv) || f[k] != floor(f[k])) { fail(RT_ERR_INVALID, "mesh vertex index out of range"); return -1; }
        for (size_t k = 0; k < 3 * (size_t)nv; k++) if (!(fabs(p[k]) < 1e30)) { fail(RT_ERR_INVALID, "mesh vertex is not finite"); return -1; }
        MeshBuild mb;
        mb.v = p; mb.f = f; mb.nodes = &o.mesh_nodes; mb.node0 = o.mesh_nodes.size();
        mb.order.resize(nt); mb.cen.resize(3 * (size_t)nt);
        for (int t = 0; t < nt; t++) {
            mb.order[t] = t;
            double a[3], b[3];
            mb.tri_box(t, a, b);
            for (int k = 0; k < 3; k++) mb.cen[3 * (size_t)t + k] = (float)(0.5 * (a[k] + b[k]));
        }
        RtMeshInfo mi;
        mi.node0 = (int64_t)o.mesh_nodes.size(); mi.tri0 = (int64_t)o.mesh_tris.size(); mi.n_tris = nt;
        mb.build(0, nt);
        mi.n_nodes = (int)(o.mesh_nodes.size() - mb.node0);
        o.mesh_tris.resize((size_t)mi.tri0 + nt);
        for (int q = 0; q < nt; q++) {
            const int t = mb.order[q];
            const double *v0 = p + 3 * (size_t)f[3 * (size_t)t], *v1 = p + 3 * (size_t)f[3 * (size_t)t + 1], *v2 = p + 3 * (size_t)f[3 * (size_t)t + 2];
            RtMeshTri &T = o.mesh_tris[(size_t)mi.tri0 + q];
            for (int k = 0; k < 3; k++) { T.v0[k] = v0[k]; T.e1[k] = v1[k] - v0[k]; T.e2[k] = v2[k] - v0[k]; }
            T.id = t; T.pad_ = 0;
        }
        o.meshes.push_back(mi);
        return (int)o.meshes.size() - 1;
    }
    int tex_node(int tex_id, int link_node)
    {
        if (tex_id <= 0) return -1;
        if (tex_id >= RT_TEX__COUNT) { fail(RT_ERR_UNSUPPORTED, "texture call-out without a device routine (id " + std::to_string(tex_id) + ")"); return -1; }
        if (tex_id >= RT_TEX_EARTH && tex_id <= RT_TEX_SKY) {       /* the planet maps: colour, bump, water (PlanetMapping's three Image arguments) */
            static const int slots[7][3] = { { RT_IMG_EARTH, RT_IMG_EARTHBUMP, RT_IMG_EARTHWATER }, { RT_IMG_EARTH, RT_IMG_EARTHBUMP, -1 }, { RT_IMG_MARS, RT_IMG_MARSBUMP, -1 },
                                             { RT_IMG_MOON, RT_IMG_MOONBUMP, -1 }, { RT_IMG_VENUS, RT_IMG_VENUSBUMP, -1 }, { RT_IMG_MERCURY, RT_IMG_MERCURYBUMP, -1 }, { RT_IMG_SKY, -1, -1 } };
            for (int k = 0; k < 3; k++) {
                const int sl = slots[tex_id - RT_TEX_EARTH][k];
                if (sl < 0) continue;
                if (d->n_images != RT_IMG__COUNT || !d->images) { fail(RT_ERR_INVALID, "a planet image-map texture without RtSceneDesc.images"); return -1; }
                const RtImageMap &m = d->images[sl];
                if (!m.pixels || m.width <= 0 || m.height <= 0 || (m.channels != 1 && m.channels != 3) || m.wrap_x < 0 || m.wrap_x > 2 || m.wrap_y < 0 || m.wrap_y > 2) {
                    fail(RT_ERR_INVALID, "planet image-map texture " + std::to_string(tex_id) + ": image slot " + std::to_string(sl) + " is missing or malformed"); return -1;
                }
            }
        }
        RtTexNode t;
        t.tex = tex_id;
        t.pa = (link_node >= 0 && link_node < (int)node_affine.size()) ? node_affine[link_node] : -1;
        t.tree_affine = affine_stack.empty() ? -1 : affine_stack.back();
        t.pad_ = 0;
        o.texnodes.push_back(t);
        return (int)o.texnodes.size() - 1;
    }

    /* emit code for subtree i whose result goes to `tag`; precondition can_hit(i) */
    /* children of the union region rooted at i, in DFS order, with the LCA depth of adjacent children */
    void collect_region(int i, int depth, std::vector<int> &kids, std::vector<int> &lca)
    {
        const RtNode &n = N(i);
        bool l = can_hit(n.left), r = can_hit(n.right);
        int sides[2] = { l ? n.left : -1, r ? n.right : -1 };
        bool first = true;
        for (int k = 0; k < 2; k++) {
            int c = sides[k];
            if (c < 0) continue;
            if (!first) lca.push_back(depth);       /* gap between the left part's last child and the right part's first */
            first = false;
            if (N(c).type == RT_NODE_UNION || N(c).type == RT_NODE_TOPUNION) collect_region(c, depth + 1, kids, lca);
            else kids.push_back(c);
        }
    }
    static double half_area(const Box &b)
    {
        if (b.empty) return 0.0;
        double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
        return x * y + y * z + z * x;
    }
    /* split idx[a,b) (b - a >= 2) in two: surface-area heuristic over all positions on all three axes (the regions
       here hold hundreds of children, not millions), or the median of the longest axis when sah is off */
    int split_range(std::vector<int> &idx, int a, int b, const std::vector<Box> &boxes, bool sah)
    {
        Box all;
        for (int k = a; k < b; k++) all = box_union(all, boxes[idx[k]]);
        int axis = 0;
        for (int d = 1; d < 3; d++) if (all.hi[d] - all.lo[d] > all.hi[axis] - all.lo[axis]) axis = d;
        auto sort_axis = [&](int ax) {
            std::sort(idx.begin() + a, idx.begin() + b, [&](int x, int y) {
                double cx = boxes[x].lo[ax] + boxes[x].hi[ax], cy = boxes[y].lo[ax] + boxes[y].hi[ax];
                return cx < cy || (cx == cy && x < y); });
        };
        int n = b - a, best_axis = axis, best_pos = (a + b) / 2;
        if (sah && n > 2) {
            double best = 1e300;
            std::vector<double> right(n);
            for (int ax = 0; ax < 3; ax++) {
                sort_axis(ax);
                Box acc;
                for (int k = n - 1; k > 0; --k) { acc = box_union(acc, boxes[idx[a + k]]); right[k] = half_area(acc); }
                acc = Box();
                for (int k = 1; k < n; k++) {
                    acc = box_union(acc, boxes[idx[a + k - 1]]);
                    double cost = half_area(acc) * k + right[k] * (n - k);
                    if (cost < best) { best = cost; best_axis = ax; best_pos = a + k; }
                }
            }
        }
        sort_axis(best_axis);
        return best_pos;
    }
    /* 4-wide BVH over idx[a0,b0) (>= 1 entries): two levels of binary splits per node; nodes are numbered breadth
       first so that the top of the tree is what lands in LDS.  Returns the root and the tree depth. */
    int build_bvh_once(std::vector<int> &idx, int a0, int b0, const std::vector<Box> &boxes, bool sah, int &depth_out)
    {
        struct Job { int node, a, b, depth; };
        std::vector<Job> queue;
        int root = (int)o.bvh.size();
        o.bvh.push_back(RtBvhNode());
        queue.push_back(Job{ root, a0, b0, 1 });
        depth_out = 1;
        for (size_t q = 0; q < queue.size(); q++) {
            Job j = queue[q];
            if (j.depth > depth_out) depth_out = j.depth;
            int cuts[5], nc = 0;
            if (j.b - j.a <= 4) { for (int k = j.a; k <= j.b; k++) cuts[nc++] = k; }
            else {
                int m = split_range(idx, j.a, j.b, boxes, sah);
                cuts[nc++] = j.a;
                if (m - j.a >= 2) cuts[nc++] = split_range(idx, j.a, m, boxes, sah);
                cuts[nc++] = m;
                if (j.b - m >= 2) cuts[nc++] = split_range(idx, m, j.b, boxes, sah);
                cuts[nc++] = j.b;
            }
            RtBvhNode node;
            memset(&node, 0, sizeof node);
            for (int c = 0; c < 4; c++) {
                node.child[c] = RT_BVH_EMPTY;
                for (int d = 0; d < 3; d++) { node.lo[d][c] = 0.0; node.hi[d][c] = 0.0; }
                if (c + 1 >= nc) continue;
                int a = cuts[c], b = cuts[c + 1];
                Box all;
                for (int k = a; k < b; k++) all = box_union(all, boxes[idx[k]]);
                for (int d = 0; d < 3; d++) {
                    double e = 1e-6 * (fabs(all.hi[d] - all.lo[d]) + fabs(all.lo[d]) + fabs(all.hi[d])) + 1e-9;
                    node.lo[d][c] = all.lo[d] - e; node.hi[d][c] = all.hi[d] + e;
                }
                if (b - a == 1) node.child[c] = ~idx[a];
                else {
                    node.child[c] = (int)o.bvh.size();
                    o.bvh.push_back(RtBvhNode());
                    queue.push_back(Job{ node.child[c], a, b, j.depth + 1 });
                }
            }
            o.bvh[j.node] = node;
        }
        return root;
    }
    int build_bvh(std::vector<int> &idx, int a0, int b0, const std::vector<Box> &boxes)
    {
        size_t mark = o.bvh.size();
        int depth = 0;
        int root = build_bvh_once(idx, a0, b0, boxes, true, depth);
        if (3 * depth + 1 > RT_BVH_STACK) {         /* too unbalanced for the traversal stack: fall back to median splits */
            o.bvh.resize(mark);
            root = build_bvh_once(idx, a0, b0, boxes, false, depth);
        }
        return root;
    }
    bool emit_region(int i, int tag)
    {
        std::vector<int> kids, lca;
        collect_region(i, 0, kids, lca);
        int K = (int)kids.size();
        /* (measured and dropped: the union at the ROOT as a region from two children on, for the generator's cull and the fast path of
           a mesh on a floor -- MESH10M 4K / 64 went from 797 to 1018 ms: the region's candidate walk and fold cost every shadow ray
           more than the two box tests they replace) */
        if (!with_bounds || K < RT_REGION_MIN || region_depth >= RT_MAX_REGION_DEPTH) return false;
        std::vector<Box> boxes(K);
        std::vector<int> finite_idx;
        RtRegion rg;
        memset(&rg, 0, sizeof rg);
        rg.n_children = K; rg.uid = next_uid++;
        rg.always0 = (int)o.region_always.size();
        bool any = false;
        for (int k = 0; k < K; k++) {
            boxes[k] = to_world(local_bound(kids[k]));
            if (boxes[k].empty) continue;                   /* can never be hit */
            any = true;
            if (boxes[k].finite()) finite_idx.push_back(k); else o.region_always.push_back(k);
        }
        if (!any) return true;
        rg.n_always = (int)o.region_always.size() - rg.always0;
        rg.bvh_root = finite_idx.empty() ? -1 : build_bvh(finite_idx, 0, (int)finite_idx.size(), boxes);
        /* sparse table of range minima over lca[0..K-2] */
        rg.lca0 = (int)o.region_lca.size();
        int levels = 1;
        while ((1 << levels) < K) levels++;
        /* level l, entry k = min(lca[k .. k + 2^l - 1]); entries past the end hold INT_MAX */
        std::vector<int> prev(lca);
        prev.resize(K, 0x7FFFFFFF);
        for (int l = 0; l < levels; l++) {
            o.region_lca.insert(o.region_lca.end(), prev.begin(), prev.end());
            std::vector<int> next(K, 0x7FFFFFFF);
            for (int k = 0; k < K; k++) next[k] = (k + (1 << l) < K) ? std::min(prev[k], prev[k + (1 << l)]) : prev[k];
            prev.swap(next);
        }
        rg.child_pc0 = (int)o.region_child_pc.size();
        o.region_child_pc.resize(o.region_child_pc.size() + K, -1);
        {
            RtBox zero;
            memset(&zero, 0, sizeof zero);
            o.region_child_boxes.resize(o.region_child_pc.size(), zero);
            for (int k = 0; k < K; k++) {
                if (boxes[k].empty) continue;
                RtBox &rb = o.region_child_boxes[rg.child_pc0 + k];
                if (!boxes[k].finite()) {           /* an "always" child (a sky dome): a box every ray meets, for the code that tests a sample tile's list */
                    for (int d = 0; d < 3; d++) { rb.lo[d] = -1.0e37; rb.hi[d] = 1.0e37; }
                    continue;
                }
                for (int d = 0; d < 3; d++) {           /* the padding of a BVH leaf (build_bvh_once) */
                    double e = 1e-6 * (fabs(boxes[k].hi[d] - boxes[k].lo[d]) + fabs(boxes[k].lo[d]) + fabs(boxes[k].hi[d])) + 1e-9;
                    rb.lo[d] = boxes[k].lo[d] - e; rb.hi[d] = boxes[k].hi[d] + e;
                }
            }
        }
        int ri = (int)o.regions.size();
        o.regions.push_back(rg);
        push(OP_REGION, ri, 0, tag);
        region_depth++;
        for (int k = 0; k < K; k++) {
            if (boxes[k].empty) continue;
            o.region_child_pc[rg.child_pc0 + k] = (int)o.prog.size();
            size_t before = o.prog.size();
            emit(kids[k], 2 * rg.uid, true);
            if (o.prog.size() == before + 1 && o.prog.back().op == OP_CHAIN) o.prog.back().b |= 1;    /* chain + return in one step */
            else push(OP_RET, ri, 0, 0);
        }
        region_depth--;
        o.regions[ri].end_pc = (int)o.prog.size();
        o.culled_subtrees += K;
        return true;
    }

    /* [affine | material | surface]* primitive  ->  one OP_CHAIN.  Table entries are created in the same order as
       the general code below would create them (pre-order), so indices, static affine tracking (Q12) and texture
       call-out levels are unchanged. */
    bool try_chain(int i, int tag, bool no_root_bound)
    {
        std::vector<int> path;
        int c = i, na = 0, nw = 0;
        for (;;) {
            if (c < 0) return false;
            int t = N(c).type;
            if (t == RT_NODE_AFFINE) na++;
            else if (t == RT_NODE_MATERIAL || t == RT_NODE_SURFACE) nw++;
            else break;
            path.push_back(c);
            c = N(c).left;
        }
        int lt = N(c).type, op;
        if (lt == RT_NODE_QUADRIC || lt == RT_NODE_CONEHYPERCYL) op = OP_QUADRIC;
        else if (lt == RT_NODE_POLYHEDRON) op = OP_POLY;
        else if (lt == RT_NODE_SLAB) op = OP_SLAB;
        else if (lt == RT_NODE_TORUS) op = OP_TORUS;
        else if (lt == RT_NODE_MESH) op = OP_MESH;
        else return false;
        if (na > 1 || nw > RT_CHAIN_WRAPS) return false;
        if ((int)affine_stack.size() + na > RT_MAX_AFFINE_DEPTH) return false;      /* let the general code report it */
        RtChain ch;
        memset(&ch, 0, sizeof ch);
        ch.box = -1;
        if (with_bounds) {
            Box b = to_world(local_bound(i));
            if (b.empty && !no_root_bound) return true;         /* statically empty: nothing to emit */
            if (!b.empty && b.finite()) { ch.box = add_box(b, true); ch.test_box = no_root_bound ? 0 : RT_CHAIN_TEST_BOX; if (ch.test_box) o.culled_subtrees++; }
        }
        ch.leaf_affine = -1;
        for (int k = 0; k < RT_CHAIN_WRAPS; k++) ch.wrap_tex[k] = -1;
        int wraps[RT_CHAIN_WRAPS], wtex[RT_CHAIN_WRAPS], wdense[RT_CHAIN_WRAPS], tex_added = 0, own_affine = -1;
        for (size_t k = 0; k < path.size(); k++) {
            int j = path[k];
            const RtNode &n = N(j);
            const double *p = D(j);
            if (n.type == RT_NODE_AFFINE) {
                RtAffine a;
                memcpy(a.minv, p + 9, sizeof a.minv);
                memcpy(a.tr, p + 18, sizeof a.tr);
                a.tree_parent = affine_stack.empty() ? -1 : affine_stack.back();
                a.link = (n.link >= 0 && n.link < (int)node_affine.size()) ? node_affine[n.link] : -1;
                int ai = (int)o.affines.size();
                o.affines.push_back(a);
                node_affine[j] = ai;
                Fwd f; memcpy(f.m, p, sizeof f.m); memcpy(f.t, p + 18, sizeof f.t);
                affine_stack.push_back(ai); fwd_stack.push_back(f);
                last_affine = ai;
                own_affine = ai;
                memcpy(ch.minv, a.minv, sizeof ch.minv); memcpy(ch.tr, a.tr, sizeof ch.tr);
                ch.n_affine = 1;
            } else if (n.type == RT_NODE_MATERIAL) {
                if (n.i0 < RT_MAT_NOSHADE || n.i0 > RT_MAT_TITANIUM) return fail(RT_ERR_INVALID, "material luster out of range"), true;
                if (n.i2 < 0 || n.i2 >= RT_PROP__COUNT) return fail(RT_ERR_UNSUPPORTED, "unknown propagator"), true;
                RtMaterialDev m;
                memset(&m, 0, sizeof m);
                m.density = p[0]; m.refr = p[1];
                for (int q = 0; q < RT_NSPEC; q++) m.color[q] = (float)p[2 + q];
                m.luster = n.i0; m.prop = n.i2; m.id = n.id;
                if (n.i1) { tex_added++; if (++tex_depth > RT_MAX_TEX) return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_TEX textured ancestors"), true; }
                m.tex = tex_node(n.i1, n.link);
                wtex[ch.n_wrap] = m.tex; wdense[ch.n_wrap] = m.density > 0.0 ? 1 : 0;
                wraps[ch.n_wrap++] = (int)o.mats.size();
                o.mats.push_back(m);
            } else {
                RtSurfaceDev sf;
                memset(&sf, 0, sizeof sf);
                sf.roughness = n.i0; sf.id = n.id;
                if (n.i1) { tex_added++; if (++tex_depth > RT_MAX_TEX) return fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_TEX textured ancestors"), true; }
                sf.tex = tex_node(n.i1, n.link);
                wtex[ch.n_wrap] = sf.tex; wdense[ch.n_wrap] = 0;
                wraps[ch.n_wrap++] = ~(int)o.surfs.size();
                o.surfs.push_back(sf);
            }
        }
        ch.prim_op = op;
        ch.prim = leaf(op, c);
        {
            const RtPrim &pr = o.prims[ch.prim];
            ch.p[0] = pr.p[0]; ch.p[1] = pr.p[1]; ch.p[2] = pr.p[2];
            ch.i0 = pr.i0; ch.plane0 = pr.plane0; ch.leaf_affine = pr.leaf_affine;
        }
        /* the hits' late normal transform walks leaf_affine's parent links (Shade.cpp:271-279): when that is this chain's own
           affine and it has no parent, the record's minv is the whole transform */
        if (own_affine >= 0 && ch.leaf_affine == own_affine && o.affines[own_affine].link < 0) ch.test_box |= RT_CHAIN_OWN_NORMAL;
        for (int k = 0; k < ch.n_wrap; k++) {                   /* innermost first */
            int src = ch.n_wrap - 1 - k;
            ch.wrap[k] = wraps[src]; ch.wrap_tex[k] = wtex[src];
            if (wdense[src]) ch.wrap_dense |= 1 << k;
        }
        for (int k = 0; k < ch.n_affine; k++) { affine_stack.pop_back(); fwd_stack.pop_back(); }
        tex_depth -= tex_added;
        o.chains.push_back(ch);
        push(OP_CHAIN, (int)o.chains.size() - 1, (op == OP_TORUS || op == OP_MESH) ? 2 : 0, tag);     /* b bit 1: a long primitive (see the interpreter's parking) */
        return true;
    }

    void emit(int i, int tag, bool no_root_bound = false)
    {
        if (rc) return;
        if (try_chain(i, tag, no_root_bound)) return;
        const RtNode &n = N(i);
        int bound_pc = -1;
        bool is_leaf = (n.type >= RT_NODE_QUADRIC && n.type <= RT_NODE_TORUS) || n.type == RT_NODE_CONEHYPERCYL || n.type == RT_NODE_MESH;
        /* a box in front of: every leaf, every affine, every binary node (not in front of pure wrappers: same box as their child) */
        bool boxed = with_bounds && !no_root_bound && (is_leaf || n.type == RT_NODE_AFFINE || (n.type >= RT_NODE_UNION && n.type <= RT_NODE_DIFFERENCE));
        if (boxed) {
            Box b = to_world(local_bound(i));
            if (b.empty) return;                       /* no points at all: statically empty */
            if (b.finite()) { bound_pc = push(OP_BOUND, add_box(b, true), -1, 0); o.culled_subtrees++; }
        }
        switch (n.type) {
        case RT_NODE_QUADRIC: case RT_NODE_CONEHYPERCYL: push(OP_QUADRIC, leaf(OP_QUADRIC, i), 0, tag); break;
        case RT_NODE_POLYHEDRON: push(OP_POLY, leaf(OP_POLY, i), 0, tag); break;
        case RT_NODE_SLAB: push(OP_SLAB, leaf(OP_SLAB, i), 0, tag); break;
        case RT_NODE_TORUS: push(OP_TORUS, leaf(OP_TORUS, i), 0, tag); break;
        case RT_NODE_MESH: push(OP_MESH, leaf(OP_MESH, i), 0, tag); break;
        case RT_NODE_AFFINE: {
            const double *p = D(i);
            RtAffine a;
            memcpy(a.minv, p + 9, sizeof a.minv);
            memcpy(a.tr, p + 18, sizeof a.tr);
            a.tree_parent = affine_stack.empty() ? -1 : affine_stack.back();
            a.link = (n.link >= 0 && n.link < (int)node_affine.size()) ? node_affine[n.link] : -1;
            int k = (int)o.affines.size();
            o.affines.push_back(a);
            node_affine[i] = k;
            if ((int)affine_stack.size() + 1 > RT_MAX_AFFINE_DEPTH) { fail(RT_ERR_UNSUPPORTED, "affine nesting deeper than RT_MAX_AFFINE_DEPTH"); return; }
            Fwd f; memcpy(f.m, p, sizeof f.m); memcpy(f.t, p + 18, sizeof f.t);
            affine_stack.push_back(k); fwd_stack.push_back(f);
            last_affine = k;
            push(OP_PUSH_AFFINE, k, 0, 0);
            emit(n.left, tag);
            push(OP_POP_AFFINE, 0, 0, 0);
            affine_stack.pop_back(); fwd_stack.pop_back();
            break;
        }
        case RT_NODE_UNION: case RT_NODE_TOPUNION: {
            bool l = can_hit(n.left), r = can_hit(n.right);
            if (l && r && emit_region(i, tag)) break;
            if (l && r) {
                int uid = next_uid++;
                emit(n.left, 2 * uid);
                emit(n.right, 2 * uid + 1);
                push(OP_UNION, 0, uid, tag);
            } else emit(l ? n.left : n.right, tag);
            break;
        }
        case RT_NODE_INTERSECTION: {
            int uid = next_uid++;
            emit(n.left, 2 * uid);
            int sk = push(OP_SKIP_EMPTY, uid, -1, 0);
            emit(n.right, 2 * uid + 1);
            push(OP_INTER, 0, uid, tag);
            o.prog[sk].b = (int)o.prog.size();
            break;
        }
        case RT_NODE_DIFFERENCE: {
            if (!can_hit(n.right)) { emit(n.left, tag); break; }
            int uid = next_uid++;
            emit(n.left, 2 * uid);
            int sk = push(OP_SKIP_EMPTY, uid, -1, 0);
            emit(n.right, 2 * uid + 1);
            push(OP_DIFF, 0, uid, tag);
            o.prog[sk].b = (int)o.prog.size();
            break;
        }
        case RT_NODE_ANGULARCLIP: {
            const double *p = D(i);
            int uid = next_uid++;
            emit(n.left, 2 * uid);
            RtClip c;
            memset(&c, 0, sizeof c);
            memcpy(c.cv1, p, sizeof c.cv1); memcpy(c.cv2, p + 3, sizeof c.cv2);
            c.intersect = n.i0; c.leaf_affine = last_affine; c.id = n.id;
            o.clips.push_back(c);
            push(OP_ANGCLIP, (int)o.clips.size() - 1, uid, tag);
            break;
        }
        case RT_NODE_BOUNDINGBOX: {
            const double *p = D(i);
            Box b = Box::cube(p[0], p[1], p[2], p[3], p[4], p[5]);
            int pc = push(OP_REFBOX, add_box(b, false), -1, 0);
            emit(n.left, tag);
            o.prog[pc].b = (int)o.prog.size();
            break;
        }
        case RT_NODE_MATERIAL: {
            const double *p = D(i);
            if (n.i0 < RT_MAT_NOSHADE || n.i0 > RT_MAT_TITANIUM) { fail(RT_ERR_INVALID, "material luster out of range"); return; }
            if (n.i2 < 0 || n.i2 >= RT_PROP__COUNT) { fail(RT_ERR_UNSUPPORTED, "unknown propagator"); return; }
            RtMaterialDev m;
            memset(&m, 0, sizeof m);
            m.density = p[0]; m.refr = p[1];
            for (int k = 0; k < RT_NSPEC; k++) m.color[k] = (float)p[2 + k];
            m.luster = n.i0; m.prop = n.i2; m.id = n.id;
            if (n.i1 && ++tex_depth > RT_MAX_TEX) { fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_TEX textured ancestors"); return; }
            m.tex = tex_node(n.i1, n.link);
            int mi = (int)o.mats.size();
            o.mats.push_back(m);
            int uid = next_uid++;
            emit(n.left, 2 * uid);
            push(OP_MATERIAL, mi, uid, tag);
            if (n.i1) tex_depth--;
            break;
        }
        case RT_NODE_SURFACE: {
            RtSurfaceDev s;
            memset(&s, 0, sizeof s);
            s.roughness = n.i0; s.id = n.id;
            if (n.i1 && ++tex_depth > RT_MAX_TEX) { fail(RT_ERR_UNSUPPORTED, "more than RT_MAX_TEX textured ancestors"); return; }
            s.tex = tex_node(n.i1, n.link);
            int si = (int)o.surfs.size();
            o.surfs.push_back(s);
            int uid = next_uid++;
            emit(n.left, 2 * uid);
            push(OP_SURFACE, si, uid, tag);
            if (n.i1) tex_depth--;
            break;
        }
        default:
            fail(RT_ERR_INVALID, "unknown node type " + std::to_string(n.type));
            return;
        }
        if (bound_pc >= 0) o.prog[bound_pc].b = (int)o.prog.size();
    }
};

} // namespace

/* Wrappers pushed down through unions (option push_wrappers), on a private copy of the node array:
     Surface(A u B)  = Surface(A) u Surface(B)      -- Surface::Intersect (Solids.cpp:1498-1519) labels every hit with t != 0 of the list
                                                        it is handed; a merge (Solids.cpp:1014-1058) selects hits and rewrites materials
                                                        and never reads the surface, so labelling before or after it is the same;
     Material(A u B) = Material(A) u Material(B)    -- when A u B holds no Material of its own and the density is positive:
                                                        Material::Intersect (Solids.cpp:1466-1487) renames every dense hit, and below
                                                        it every dense hit is neutronium, so the merge compares equal densities on
                                                        either side of the rewrite, keeps the same hits and hands back the same name.
   What it buys: a scene like Surface(Union of nine objects) + floor, or Surface(Material(cage of 42 struts)) + floor + dome, becomes
   ONE union region at the root -- the generator's cull, the per-tile candidate lists and the fast path apply -- instead of a wrapper
   around a nested region.  Hit lists are unchanged (the oracle sees the description as it came). */
static int g_push_wrappers = 0;
void rt_flatten_set_push_wrappers(int on) { g_push_wrappers = on ? 1 : 0; }
namespace {
struct WrapperPush {
    std::vector<RtNode> &nd;
    const double *data;
    std::vector<signed char> has_mat;       /* memo: subtree holds a Material (-1 unknown) */
    WrapperPush(std::vector<RtNode> &n, const double *dd) : nd(n), data(dd), has_mat(n.size(), -1) {}
    bool holds_material(int i)
    {
        if (i < 0) return false;
        if ((size_t)i < has_mat.size() && has_mat[i] >= 0) return has_mat[i] != 0;
        const bool r = nd[i].type == RT_NODE_MATERIAL || holds_material(nd[i].left) || holds_material(nd[i].right);
        if ((size_t)i < has_mat.size()) has_mat[i] = r ? 1 : 0;
        return r;
    }
    void wrapper(int i)                     /* node i is a wrapper whose subtree has been visited: push it down while it sits on a union */
    {
        if (nd[i].type != RT_NODE_SURFACE && nd[i].type != RT_NODE_MATERIAL) return;
        const int c = nd[i].left;
        if (c < 0 || (nd[c].type != RT_NODE_UNION && nd[c].type != RT_NODE_TOPUNION) || nd[c].left < 0 || nd[c].right < 0) return;
        if (nd[i].type == RT_NODE_MATERIAL && (!(data[nd[i].data] > 0.0) || holds_material(c))) return;
        RtNode w1 = nd[i], w2 = nd[i], u = nd[c];
        w1.left = u.left; w2.left = u.right;
        const int i1 = (int)nd.size();
        nd.push_back(w1); nd.push_back(w2);
        u.left = i1; u.right = i1 + 1;
        nd[i] = u;                          /* the wrapper's slot becomes the union (its parent keeps pointing here) */
        wrapper(i1); wrapper(i1 + 1);
    }
    void visit(int i, int depth)
    {
        if (i < 0 || depth > 4096) return;
        const int l = nd[i].left, r = nd[i].right;
        visit(l, depth + 1);
        const int t = nd[i].type;
        if (t == RT_NODE_UNION || t == RT_NODE_TOPUNION || t == RT_NODE_INTERSECTION || t == RT_NODE_DIFFERENCE) visit(r, depth + 1);
        wrapper(i);
    }
};
}

int rt_flatten_scene(const RtSceneDesc *d_in, RtFlatScene &out, std::string &err, bool with_bounds)
{
    const RtSceneDesc *d = d_in;
    if (!d || d->n_nodes < 0 || (d->n_nodes && !d->nodes)) { err = "null scene description"; return RT_ERR_INVALID; }
    for (int i = 0; i < d->n_nodes; i++) {
        const RtNode &n = d->nodes[i];
        if (n.left >= d->n_nodes || n.right >= d->n_nodes || n.link >= d->n_nodes || n.data < 0 || n.data + n.ndata > d->n_data) { err = "node index out of range"; return RT_ERR_INVALID; }
        if (n.left >= 0 && n.left <= i) { err = "scene is not a tree in pre-order"; return RT_ERR_INVALID; }
        if (n.right >= 0 && n.right <= i) { err = "scene is not a tree in pre-order"; return RT_ERR_INVALID; }
    }
    if (d->n_lights > RT_MAX_LIGHTS) { err = "more than RT_MAX_LIGHTS point lights"; return RT_ERR_UNSUPPORTED; }
    std::vector<RtNode> pushed;
    RtSceneDesc d2;
    if (g_push_wrappers && with_bounds && d->n_nodes > 0 && d->root >= 0) {
        pushed.assign(d->nodes, d->nodes + d->n_nodes);
        pushed.reserve((size_t)d->n_nodes * 2 + 16);
        WrapperPush wp(pushed, d->data);
        wp.visit(d->root, 0);
        if ((int)pushed.size() != d->n_nodes) {
            d2 = *d;
            d2.nodes = &pushed[0]; d2.n_nodes = (int)pushed.size();
            d = &d2;
        }
    }
    Flattener f(d, out, err, with_bounds);
    f.node_affine.assign((size_t)(d->n_nodes > 0 ? d->n_nodes : 1), -1);
    /* built-in materials / surfaces (Solids.cpp:1463-1464, :169) */
    RtMaterialDev vac, neu;
    memset(&vac, 0, sizeof vac); memset(&neu, 0, sizeof neu);
    vac.density = 0.0; vac.refr = 1.0 + 0.0; vac.luster = RT_MAT_TRANSPARENT; vac.tex = -1; vac.id = -1;
    neu.density = 1000000.0; neu.refr = 1.0 + 1000000.0; neu.luster = RT_MAT_PLASTIC; neu.tex = -1; neu.id = -2;
    for (int k = 0; k < RT_NSPEC; k++) { vac.color[k] = 0.0f; neu.color[k] = 1.0f; }
    out.mats.push_back(vac); out.mats.push_back(neu);
    RtSurfaceDev none, matte;
    memset(&none, 0, sizeof none); memset(&matte, 0, sizeof matte);
    none.roughness = RT_SURF_MATTE; none.tex = -1; none.id = -9;
    matte.roughness = RT_SURF_MATTE; matte.tex = -1; matte.id = -3;
    out.surfs.push_back(none); out.surfs.push_back(matte);
    if (d->root >= 0 && f.can_hit(d->root)) f.emit(d->root, 0);
    f.push(OP_END, 0, 0, 0);
    if (f.rc) return f.rc;
    for (int i = 0; i < d->n_lights; i++) {
        RtLightDev l;
        memset(&l, 0, sizeof l);
        memcpy(l.org, d->lights[i].org, sizeof l.org); memcpy(l.vA, d->lights[i].vA, sizeof l.vA); memcpy(l.vB, d->lights[i].vB, sizeof l.vB);
        memcpy(l.flux, d->lights[i].flux, sizeof l.flux);
        l.type = d->lights[i].type; l.no_shadow = d->lights[i].no_shadow;
        out.lights.push_back(l);
    }
    return 0;
}
