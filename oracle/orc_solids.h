/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * orc_solids.h: CPU restatement of the reference's CSG intersection (Solids.cpp).  The object graph
 * is rebuilt from an RtSceneDesc as a pointer tree with one virtual hits() per node, walked
 * recursively per ray, exactly as the reference does (RayTracer2020.h:68-75).  Hit lists are the
 * reference's full sorted boundary lists (RayTracer2020.h:108-132), here std::vector<Hit>.
 */
#pragma once
#include <vector>
#include <stdio.h>
#include "orc_math.h"
#include "../include/rt_scene_desc.h"

#define RT_MAXROOTS_ORC 16     /* MAXHITS, Solids.h:20 */

namespace orc {

struct Ray {                    /* RayTracer2020.h:26-36 */
    V3 org, dir;
};
static inline V3 at(const Ray &r, double t) { return r.org + r.dir * t; }

struct Node;
struct TexRef {                 /* TextureList, Solids.h:295-303 */
    V3    p;                    /* ray(t) in the coordinates of the node that queued it */
    int   tex;                  /* RtTextureId */
    Node *pa;                   /* affine level at which it fires */
    const RtImageMap *images;   /* the scene's planet image maps (RtSceneDesc.images), for the PlanetMapping call-outs */
};
struct Hit {                    /* RayTracer2020.h:108-132 */
    double  t;
    V3      n;                  /* cvNormal */
    V3      np;                 /* cvPerturbed */
    int     perturbed;
    Node   *surf;               /* ps: 0 = none (ray starts inside), kMatte = the static matte surface */
    Node   *mat;                /* pm: kVacuum / kNeutronium / a Material node */
    Node   *affine;             /* paAffineTransforms */
    Node   *leaf;               /* instrumentation: primitive that produced the root */
    std::vector<TexRef> tex;
};
/* Q13: the solid colour textures write pm->color / pm->nLuster of the SHARED Material (TextureMaps.cpp:530-683).  Stated
   policy, as for the spectral metals (Q3): the write goes to a private copy of the material that this one hit refers to
   from then on, so a hit is coloured by the texture value at its own position and nothing leaks to other rays. */
typedef std::vector<Hit> HitList;

struct Extent { V3 lo, hi; };

struct Ctx {                    /* the reference's file-static / global state, made explicit */
    Node  *last_affine;         /* s_paLastHitAffine, Solids.cpp:170 -- deliberately persists across rays */
    int    depth;               /* static nRayDepth, Shade.cpp:416 */
    double aux[6];              /* g_uCamera,g_vCamera,g_uLight,g_vLight,g_uMicroFacet,g_vMicroFacet */
    long   n_primary, n_shadow, n_secondary;
    mutable std::vector<Node *> scratch;    /* per-hit material copies made by colour textures (Q13); emptied per primary ray */
    Node  *clone_material(const Node *m) const;
    void   drop_scratch() const;
    ~Ctx() { drop_scratch(); }
    Ctx() : last_affine(0), depth(0), n_primary(0), n_shadow(0), n_secondary(0) { for (int i = 0; i < 6; i++) aux[i] = 0.0; }
};

struct Node {
    int     type, id, index;
    Node   *left, *right;
    int     i0, i1, i2;
    Node   *link;               /* affine: paParent; material/surface: paAffineTransforms */
    /* primitive parameters */
    double  P, Q, R, Rneck, minor;
    std::vector<V3> planes;
    /* triangle mesh (policy Q14): vertices, 3 vertex indices per triangle; its extent is kept in ex */
    std::vector<V3>  mverts;
    std::vector<int> mtris;
    /* affine */
    M3      m, mi;
    V3      tr;
    /* angular clip */
    V3      cv1, cv2;
    /* bounding box */
    Extent  ex;
    /* material */
    double  density, refr;
    Spec    color;
    const RtImageMap *images;   /* material / surface: the scene's planet image maps, 0 = none */
    Node() : type(0), id(-1), index(-1), left(0), right(0), i0(0), i1(0), i2(0), link(0),
             P(0), Q(0), R(0), Rneck(0), minor(0), density(0), refr(1), images(0) {}
};

/* g_mVacuum / g_mNeutronium / s_sMatte: Solids.cpp:1463-1464, 169 */
struct Builtins {
    Node vacuum, neutronium, matte;
    Builtins()
    {
        vacuum.type = RT_NODE_MATERIAL; vacuum.id = -1; vacuum.density = 0.0; vacuum.refr = 1.0 + 0.0;
        vacuum.color = spec_const(0.0f); vacuum.i0 = RT_MAT_TRANSPARENT;
        neutronium.type = RT_NODE_MATERIAL; neutronium.id = -2; neutronium.density = 1000000.0; neutronium.refr = 1.0 + 1000000.0;
        neutronium.color = spec_const(1.0f); neutronium.i0 = RT_MAT_PLASTIC;
        matte.type = RT_NODE_SURFACE; matte.id = -3; matte.i0 = RT_SURF_MATTE;
    }
};
extern Builtins g_b;

/* ---- root finders: Solids.cpp:277-561 ---- */
int solve_linear(double *roots, double a, double b);
int solve_quadratic(double *roots, double a, double b, double c);
double cube_root(double x);
int solve_cubic(double *roots, double a, double b, double c, double d);
int solve_quartic(double *roots, double a, double b, double c, double d, double e);

int  extent_test(const Extent &e, const Ray &ray);               /* Extent::TestIntersection, Solids.cpp:1533-1572 */
Extent node_extent(const Node *n);                               /* virtual GetExtent() */
void node_hits(Node *n, const Ray &ray, Ctx &cx, HitList &out);  /* virtual Intersect() */
Node *optimize(Node *n, Node *enclosing_affine);                 /* virtual Optimize() */

} // namespace orc
