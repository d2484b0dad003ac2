/*
 * rt_device.h -- flattened scene layout shared by the host flattener (rt_flatten.cpp) and the HIP
 * kernels (rt_kernels.hip).
 *
 * The reference walks a pointer tree of virtual Solid::Intersect calls per ray
 * (RayTracer2020.h:68-75).  Here the optimised tree becomes
 *   - a linear PROGRAM (post-order, 16-byte instructions) that a per-ray stack machine interprets;
 *     every list-producing instruction names the instruction that will consume its result, so empty
 *     lists are never materialised and the pending stack holds non-empty lists only;
 *   - conservative world-space boxes in front of every finite subtree (skip = BVH-style culling
 *     that cannot change a hit list: a ray that misses the box yields the empty list either way);
 *   - SoA parameter tables (affine transforms, primitives, planes, materials, surfaces, texture
 *     call-outs, lights).
 * Everything is tiny next to HBM (S0: 257 spheres = ~60 KB) and lives in L2/LDS; the HBM-resident
 * data of a render are the per-sample and per-frame arrays allocated in rt_api.hip.
 */
#pragma once
#include <stdint.h>
#include "../../include/rt_scene_desc.h"

#define RT_MAX_AFFINE_DEPTH 12      /* nested (merged) affine levels on a root-to-leaf path */
#define RT_MAX_TEX          4       /* textured Material/Surface ancestors per hit */
#define RT_LIST_STACK       24      /* pending non-empty hit lists per ray */
#define RT_ARENA            512     /* working hits per ray (private memory, or the lane's part of the pool) */
#define RT_ARENA_SOFT       256     /* ... of which the first 256 are used while the live lists fit there: dead records are packed out when a
                                       new list would end above this mark (arena_compact), so that a ray's records stay in as little memory as
                                       possible; only a ray with more live hits than this goes higher (the last Laboratory frames do) */
#define RT_MAX_LIGHTS       8
#define RT_MAX_DEPTH        11      /* Shade.cpp:420 */
#define RT_MAX_CAND         64      /* children of one union region whose boxes a ray may touch */
#define RT_MAX_REGION_DEPTH 4       /* nested union regions */
#ifndef RT_TILE_CAND
#define RT_TILE_CAND        16      /* children of the root region listed per sampling tile (k_tile_mask); 8 left 1 % of S0's primary rays to the interpreter (tiles near the horizon that see more than eight spheres): 23.4 vs 21.8 ms per 4K / 64 frame */
#endif
#if RT_TILE_CAND > 31
#error "a queue entry of the fast path holds the touched entries of a tile's list as bits 0..30"
#endif
#define RT_REGION_MIN       4       /* unions of fewer children stay plain binary code */

enum RtOp {
    OP_END = 0,
    OP_BOUND,       /* a: box index, b: jump target if the world ray misses the box */
    OP_REFBOX,      /* a: box index (reference BoundingBox extent, local frame), b: jump target; Solids.h:495 */
    OP_PUSH_AFFINE, /* a: affine index */
    OP_POP_AFFINE,
    OP_QUADRIC,     /* a: primitive index */
    OP_POLY,        /* a: primitive index */
    OP_SLAB,        /* a: primitive index */
    OP_TORUS,       /* a: primitive index */
    OP_MATERIAL,    /* a: material index (>= 2) */
    OP_SURFACE,     /* a: surface index (>= 2) */
    OP_UNION,
    OP_INTER,
    OP_DIFF,
    OP_ANGCLIP,     /* a: clip index */
    OP_SKIP_EMPTY,  /* b: jump target when the left operand of the binary op at `cons` is empty (Solids.cpp:1152-1157, :1190-1195) */
    OP_RETAG,       /* pass the pending list (if any) on to a new consumer: Union/Difference with a null child */
    OP_REGION,      /* a: union region index, tag: consumer of the region's result */
    OP_RET,         /* a: union region index: end of one region child's code */
    OP_CHAIN,       /* a: chain index: [box] affine* primitive (material|surface)* in one step; b bit 0: followed by the region's OP_RET, bit 1: the primitive is a torus or a mesh */
    OP_MESH         /* a: primitive index (RtPrim.plane0 = mesh index): closed triangle mesh, policy Q14 */
};

/* Every instruction that leaves a list names its consumer: tag = 2 * (consumer pc) + side (0 left / only, 1 right). */
typedef struct RtInstr {
    int32_t op;
    int32_t a;
    int32_t b;
    int32_t tag;
} RtInstr;

typedef struct RtBox { double lo[3], hi[3]; } RtBox;

typedef struct RtAffine {       /* AffineSolid after Optimize() (Solids.cpp:957-983) */
    double minv[9];             /* mInverse, row major */
    double tr[3];               /* vTranslate */
    int32_t link;               /* paParent (normal-transform chain, Shade.cpp:271-279), -1 none */
    int32_t tree_parent;        /* enclosing affine in the tree (ray-transform chain), -1 none */
} RtAffine;

typedef struct RtPrim {
    double  p[4];               /* quadric: P,Q,R ; torus: minor radius */
    int32_t i0;                 /* quadric: nClip ; polyhedron: nSides */
    int32_t plane0;             /* polyhedron: first plane (index into planes, 3 doubles each) */
    int32_t leaf_affine;        /* affine that hits of this primitive record (s_paLastHitAffine, Solids.cpp:213), -1 none */
    int32_t id;                 /* RtNode.id */
} RtPrim;

typedef struct RtMaterialDev {  /* index 0 = g_mVacuum, 1 = g_mNeutronium (Solids.cpp:1463-1464) */
    double  density, refr;
    float   color[RT_NSPEC];
    int32_t luster, prop;
    int32_t tex;                /* texture call-out index (into RtTexNode), -1 none */
    int32_t id;
} RtMaterialDev;

typedef struct RtSurfaceDev {   /* index 0 = none (ps == 0), 1 = the static matte surface (Solids.cpp:169) */
    int32_t roughness;
    int32_t tex;
    int32_t id;
    int32_t pad_;
} RtSurfaceDev;

typedef struct RtTexNode {      /* a queued TextureMap call-out (Solids.cpp:1476-1483, :1509-1516) */
    int32_t tex;                /* RtTextureId */
    int32_t pa;                 /* paAffineTransforms of the owning node: level at which it fires, -1 = none */
    int32_t tree_affine;        /* innermost affine enclosing the owning node in the tree: frame of pt->p = ray(t) */
    int32_t pad_;
} RtTexNode;

typedef struct RtClip {         /* AngularClip, Solids.h:409-419 */
    double  cv1[3], cv2[3];
    int32_t intersect;
    int32_t leaf_affine;
    int32_t id;
    int32_t pad_;
} RtClip;

/* A union region: a maximal connected set of Union/TopUnion nodes with its K non-union children in DFS
 * order.  The reference evaluates every child and merges along the binary tree (Solids.cpp:1060-1075);
 * a child whose list is empty is the identity there, so only children whose conservative box the ray
 * touches need evaluating -- found with a spatial BVH -- PROVIDED the non-empty lists are merged in the
 * tree's own order.  That order is recovered from lca[]: lca_depth(i, i+1) of adjacent children, with a
 * sparse table for range minima: of three consecutive non-empty lists a < b < c, (a,b) merge first iff
 * depth(lca(a,b)) > depth(lca(b,c)). */
typedef struct RtRegion {
    int32_t n_children;
    int32_t child_pc0;      /* first entry in region_child_pc */
    int32_t bvh_root;       /* index into bvh, -1 = none */
    int32_t always0, n_always;  /* children with unbounded boxes: always candidates (indices into region_always) */
    int32_t lca0;           /* sparse table base: level l starts at lca0 + l * n_children */
    int32_t end_pc;
    int32_t uid;            /* children produce lists tagged 2*uid */
} RtRegion;

/* 4-wide node: the four child boxes are tested together (independent arithmetic hides latency at one or two
   waves per SIMD), the tree is half as deep as a binary one, and the first RT_LDS_BVH_NODES nodes of a scene
   are staged in LDS by every workgroup. */
#define RT_BVH_EMPTY     ((int32_t)0x80000000)
#define RT_LDS_BVH_NODES 192        /* 192 * 208 B = 39 KB of LDS */
#define RT_LDS_CHAINS    320        /* 320 * 208 B = 65 KB of LDS (persistent tracing kernels only) */
#define RT_BVH_STACK     28         /* pending nodes per ray (LDS, 2 B each) */
typedef struct RtBvhNode {
    double  lo[3][4], hi[3][4];
    int32_t child[4];       /* >= 0: node index; < 0: ~(region child index) (leaf); RT_BVH_EMPTY: unused slot */
} RtBvhNode;

/* A chain: single-child wrappers around one primitive, the commonest shape of a scene object
   (Translate(Scale(Material(Surface(Sphere)))) -- after Optimize() one merged affine at most).  Evaluated by one
   OP_CHAIN instead of [BOUND] PUSH_AFFINE primitive MATERIAL/SURFACE.. POP_AFFINE: same operations in the same
   order.  The record is self-contained (208 bytes: the affine, the primitive's parameters, what the wrappers do to
   a hit), so that one ray needs one record, and the first records of a scene are staged in LDS. */
#define RT_CHAIN_WRAPS   6
#define RT_CHAIN_TEST_BOX   1
#define RT_CHAIN_OWN_NORMAL 2
typedef struct RtChain {
    double  minv[9], tr[3];     /* the affine (n_affine == 1): AffineSolid after Optimize(), Solids.cpp:957-983 */
    double  p[3];               /* quadric P,Q,R ; torus: minor radius in p[0] */
    int32_t prim_op, prim;      /* OP_QUADRIC.. ; index into prims (ids for the probes) */
    int32_t i0, plane0;         /* quadric nClip / polyhedron nSides ; first plane */
    int32_t leaf_affine;        /* affine the hits record (s_paLastHitAffine, Solids.cpp:213), -1 none */
    int32_t n_affine;           /* 0 or 1 */
    int32_t box;                /* the chain's conservative world box, -1 = unbounded */
    int32_t test_box;           /* bit 0: test the box (0 when the caller, a region's BVH leaf, has already tested it); bit 1 (RT_CHAIN_OWN_NORMAL): the
                                   late normal transform of this chain's hits (Shade.cpp:271-279) is this record's own minv alone -- leaf_affine is the
                                   chain's affine and has no parent link -- so a kernel that holds the record needs no affine table */
    int32_t n_wrap;
    int32_t wrap_dense;         /* bit w: wrapper w is a material of density > 0 */
    int32_t wrap[RT_CHAIN_WRAPS];       /* innermost first; >= 0: material index, < 0: ~surface index */
    int32_t wrap_tex[RT_CHAIN_WRAPS];   /* texture call-out the wrapper queues, -1 none */
} RtChain;

/* Triangle meshes (policy Q14).  One 4-wide BVH per mesh over its triangles, in the mesh's own frame: a node is 64 bytes
   holding four child boxes that are tested together -- half the dependent loads of a binary tree, and the four slab tests
   are independent arithmetic.  The boxes are 8-bit offsets on a per-node grid (org + q * scale per axis, lower corners
   rounded down and upper ones up, so a decoded box contains the child's float box, which contains its triangles): a walk
   is a chain of divergent node reads, and what bounds it is how many 16-byte pieces the texture path has to fetch per lane
   and how many nodes the caches hold -- measured on the 10 M-triangle scene: 128-byte nodes with float boxes took 8 vector
   loads per visit and missed L2 on 40 % of them.  Triangles are stored in leaf order as (v0, e1, e2) with the edges
   subtracted once on the host -- ObjectFormat.cpp:85-86 subtracts them per test, same doubles either way. */
#define RT_MESH_MAXHITS 16          /* MAXHITS, Solids.h:20: crossings kept per ray and mesh (the nearest ones) */
#define RT_MESH_STACK   48          /* pending BVH entries per ray (LDS, 4 B each): 3 per level + 1, 2^27 triangles = 13 levels */
#define RT_MESH_LEAF    4           /* triangles per leaf at most */
#define RT_MESH_EMPTY   ((int32_t)0x80000000)      /* unused child slot (its box is inverted: never hit) */
typedef struct __attribute__((aligned(64))) RtMeshNode {
    float   org[3], scale[3];           /* the node's grid: coordinate = org + q * scale, q in 0..255 */
    uint8_t qlo[3][4], qhi[3][4];       /* [axis][slot] */
    int32_t child[4];       /* >= 0: node index relative to the mesh's first node; RT_MESH_EMPTY: unused; else a leaf: ~child = first triangle | (count - 1) << 27 */
} RtMeshNode;
typedef struct RtMeshTri {
    double  v0[3], e1[3], e2[3];
    int32_t id, pad_;       /* index in the file's order: breaks ties between equal t */
} RtMeshTri;
typedef struct RtMeshInfo {
    int64_t node0, tri0;    /* first node / triangle of this mesh in the scene's arrays */
    int32_t n_nodes, n_tris;
} RtMeshInfo;

typedef struct RtLightDev {
    double  org[3], vA[3], vB[3];
    float   flux[RT_NSPEC];
    int32_t type, no_shadow;
} RtLightDev;

/* an image of RtSceneDesc.images in device memory (PlanetMapping's maps) */
typedef struct RtImageDev {
    const float *px;            /* [ch][h][w], null = slot not loaded */
    int32_t w, h, ch, wrap_x, wrap_y, pad_;
} RtImageDev;

/* constant part of a scene, one copy in device memory */
typedef struct RtDevScene {
    const RtInstr       *prog;
    const RtBox         *boxes;
    const RtAffine      *affines;
    const RtPrim        *prims;
    const double        *planes;
    const RtMaterialDev *mats;
    const RtSurfaceDev  *surfs;
    const RtTexNode     *texnodes;
    const RtClip        *clips;
    const RtLightDev    *lights;
    const RtRegion      *regions;
    const RtBvhNode     *bvh;
    const RtChain       *chains;
    const int32_t       *region_child_pc;
    const int32_t       *region_always;
    const int32_t       *region_lca;
    const double        *blue;          /* g_rgvBlueNoise: 1600 x 2 */
    const int32_t       *qi2, *qi3;     /* g_BlueNoiseQuasiIndex2/3 */
    const double        *disk128;       /* DiskSamples128: 128 x 2 (LowDiscrepancy.cpp:129) */
    const int32_t       *q128;          /* rgnQuasiIndex2, 3, 5: 3 x 128 (LowDiscrepancy.cpp:262-333) */
    const float         *metal_nk;      /* [11][2][10]: refraction, extinction per MAT_* */
    int32_t n_prog, n_lights, has_ambient, sampler, venera, n_bvh;
    int32_t n_bvh_lds, n_chains;        /* BVH nodes staged in LDS; number of chain records */
    int32_t n_boxes, n_affines, n_mats, n_surfs, n_regions, n_child_pc, n_lca;      /* table sizes, for staging them in LDS */
    int32_t aux_light, aux_tex;         /* which per-sample auxiliary dimensions outlive sample generation: g_u/vLight (an area light
                                           exists), g_u/vMicroFacet (a texture call-out exists); g_u/vCamera never does */
    int32_t root_cull, root_cull_arg;   /* 0: none; 1: the whole program is union region root_cull_arg; 2: ... is one chain with box root_cull_arg */
    float   fractal_omega;      /* powf(2.1f, -1.0f) of GradientFractalNoise (Texture.cpp:251), evaluated by the host's libm */
    int32_t pad1_;
    float   ambient[RT_NSPEC];
    float   spec2rgb[RT_NSPEC][3];
    float   chan[3][RT_NSPEC];
    RtCamera cam;
    double  aux_init[6];
    float   tex_palette[5][RT_NSPEC];   /* g_sGreen, g_sEqualWhite, g_sOrange, g_sRed, g_sBlue: the solid colour textures' spectra */
    float   fractal_omega19;            /* powf(1.9f, -1.0f) from the host's libm (FractalNoise of the solid textures) */
    int32_t n_meshes;
    const RtMeshInfo *meshes;
    const RtMeshNode *mesh_nodes;
    const RtMeshTri  *mesh_tris;
    const RtBox      *region_child_boxes;   /* parallel to region_child_pc */
    const int32_t    *root_alt_pc;          /* root_cull == 1: per child of the root region, pc of "that chain for the root consumer; OP_END" behind the program, -1 = the child is not a bare chain */
    int32_t           n_root_alt, root_child_pc0;
    const int32_t    *root_fast;            /* root_cull == 1: per child of the root region, the chain k_fast may run for a primary ray that can touch this child alone, -1 = not eligible; null = no fast path */
    /* the mesh pre-walk (k_mesh_cross): the program holds exactly ONE mesh instruction, reached through the affines
       mesh_pre_aff[0 .. mesh_pre_naff) (outermost first) and, when the mesh is a chain's primitive, that chain's own affine;
       mesh_pre_naff < 0 = no pre-walk (no mesh, or more than one) */
    int32_t           mesh_pre_naff, mesh_pre_chain, mesh_pre_mesh;
    int32_t           sched;                /* the scene runs the FULL kernels whose interpreter takes one class of opcodes per round (intersect_scene, SCHED): long programs, whose rays sit at different instructions */
    int32_t           mesh_pre_aff[RT_MAX_AFFINE_DEPTH];
    const float      *cull32;               /* parallel to region_child_boxes: lo[3], hi[3] in single precision, grown for float arithmetic (k_generate_cull); null = off */
    const RtImageDev *images;               /* [RT_IMG__COUNT], null = the scene has no planet image-map texture */
    const float      *planet_spec;          /* [6][RT_NSPEC]: g_sWhite, g_sRed, g_sGreen, g_sBlue, g_sD6500 (RGBtoSpectrum, RadiometryColor.cpp:947-963), s_sOcean (TextureMaps.cpp:1123) */
} RtDevScene;
