/*
 * rt_scene_desc.h -- plain-C description of a scene, shared by
 *   - the host front end (ray-tracer_amd/frontend/, which mirrors the reference's C++ scene API and
 *     serialises the object graph it builds),
 *   - the MI355X render library (ray-tracer_amd/csrc/, C-ABI in rt_amd.h), and
 *   - the CPU oracle (oracle/, test infrastructure only).
 *
 * The description is the reference's object graph written down as flat arrays: one RtNode per
 * `Solid` (RayTracer2020.h:68-75, Solids.h), child pointers as indices, doubles in one pool.
 * No pointers to C++ objects, no torch types.
 */
#ifndef RT_SCENE_DESC_H
#define RT_SCENE_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_NSPEC        10          /* RadiometryColor.h:10-11: 380..740 nm step 40 */
#define RT_BLUE_NSQRT   40          /* Patterns.h:16 */
#define RT_BLUE_N       1600        /* Patterns.h:17 */
#define RT_EPSILON      2.0e-7      /* RayTracer2020.h:17 */
#define RT_INFINITY     1.0e+37     /* RayTracer2020.h:18 */

/* Solid subclasses (Solids.h:5-17) */
enum RtNodeType {
    RT_NODE_NULL         = 0,   /* NullSolid                          Solids.h:60 */
    RT_NODE_QUADRIC      = 1,   /* data: P,Q,R        i0: nClip       Solids.h:65-75 */
    RT_NODE_POLYHEDRON   = 2,   /* data: 3*nSides     i0: nSides      Solids.h:130-138 */
    RT_NODE_SLAB         = 3,   /*                                    Solids.h:140-144 */
    RT_NODE_TORUS        = 4,   /* data: minor radius                 Solids.h:177-183 */
    RT_NODE_AFFINE       = 5,   /* data: M[9],Minv[9],T[3]; left=child; link=paParent     Solids.h:309-320 */
    RT_NODE_UNION        = 6,   /* left,right                         Solids.h:371-379 */
    RT_NODE_TOPUNION     = 7,   /* left,right                         Solids.h:381-387 */
    RT_NODE_INTERSECTION = 8,   /* left,right                         Solids.h:389-397 */
    RT_NODE_DIFFERENCE   = 9,   /* left,right                         Solids.h:399-407 */
    RT_NODE_ANGULARCLIP  = 10,  /* left=ps; data: cv1[3],cv2[3]; i0: nIntersect           Solids.h:409-419 */
    RT_NODE_BOUNDINGBOX  = 11,  /* left=ps; data: pMin[3],pMax[3] (ex computed at construction)  Solids.h:490-498 */
    RT_NODE_MATERIAL     = 12,  /* left=psChild; data: fDensity,fRefractiveIndex,color[10];
                                   i0: nLuster, i1: texture id, i2: propagator id; link=paAffineTransforms  Solids.h:269-291 */
    RT_NODE_SURFACE      = 13,  /* left=psChild; i0: nRoughness, i1: texture id; link=paAffineTransforms   Solids.h:201-212 */
    RT_NODE_CONEHYPERCYL = 14,  /* Quadric with own GetExtent; data: P,Q,R,Rneck; i0: nClip      Solids.h:116-121 */
    RT_NODE_MESH         = 15   /* closed triangle mesh as a Solid.  data: 3*i0 vertex coordinates, then 3*i1 vertex indices
                                   (0-based, stored as doubles); i0: nVertices, i1: nTriangles.  The reference holds meshes
                                   (ObjectFormat.cpp:12-28, LoadObjFile :31-75) and a Moeller-Trumbore test (:79-102) but never
                                   makes them a Solid: policy Q14 in DESIGN.md says how this node behaves. */
};

/* SURF_* (Solids.h:187-192) and MAT_* (Solids.h:214-224) keep the reference's integer values. */
enum { RT_SURF_MATTE = 1, RT_SURF_BURNISHED = 2, RT_SURF_SMOOTH = 3, RT_SURF_SHINY = 4, RT_SURF_POLISHED = 5, RT_SURF_WET = 6 };
enum { RT_MAT_NOSHADE = 0, RT_MAT_PLASTIC = 1, RT_MAT_METAL = 2, RT_MAT_TRANSPARENT = 3, RT_MAT_COPPER = 4, RT_MAT_GOLD = 5,
       RT_MAT_IRON = 6, RT_MAT_ALUMINUM = 7, RT_MAT_OSMIUM = 8, RT_MAT_SILVER = 9, RT_MAT_TITANIUM = 10 };

/* TextureMap function pointers (Solids.h:199, TextureMaps.h) become ids; 0 = none. */
enum RtTextureId {
    RT_TEX_NONE = 0, RT_TEX_BUMPY = 1, RT_TEX_MICROFACETS = 2, RT_TEX_MICROFACETS2 = 3, RT_TEX_WRINKLED = 4,
    RT_TEX_FACETED20 = 5, RT_TEX_DENTED = 6, RT_TEX_WAVY = 7, RT_TEX_FIBERS = 8, RT_TEX_BLISTERED = 9,
    RT_TEX_HAMMERED = 10, RT_TEX_TECTONIC = 11, RT_TEX_QUASIMICROFACET = 12, RT_TEX_VORONOIBUMP = 13,
    RT_TEX_VORONOIANTIBUMP = 14, RT_TEX_VORONOICRINKLED = 15,
    /* solid colour textures (TextureMaps.cpp:518-700) and SeaScape (:497-513, which has no effect on the hit) */
    RT_TEX_VORONOI = 16, RT_TEX_AGATE = 17, RT_TEX_MARBLE = 18, RT_TEX_CLOUDS = 19, RT_TEX_WOOD = 20, RT_TEX_SPOTS = 21,
    RT_TEX_VEINS = 22, RT_TEX_SEASCAPE = 23,
    /* the planet image maps (PlanetMapping, TextureMaps.cpp:955-1084, through the wrappers :1098-1243): colour map, bump map and
       water mask come from RtSceneDesc.images, by the slots of RtImageSlot */
    RT_TEX_EARTH = 24, RT_TEX_EARTH2 = 25, RT_TEX_MARS = 26, RT_TEX_MOON = 27, RT_TEX_VENUS = 28, RT_TEX_MERCURY = 29, RT_TEX_SKY = 30,
    RT_TEX__COUNT
};
/* The images PlanetMapping reads, one slot per file-static Image of TextureMaps.cpp:945-947 (file names: :1093-1230). */
enum RtImageSlot {
    RT_IMG_SKY = 0, RT_IMG_EARTH = 1, RT_IMG_EARTHBUMP = 2, RT_IMG_EARTHWATER = 3, RT_IMG_MARS = 4, RT_IMG_MARSBUMP = 5,
    RT_IMG_MOON = 6, RT_IMG_MOONBUMP = 7, RT_IMG_VENUS = 8, RT_IMG_VENUSBUMP = 9, RT_IMG_MERCURY = 10, RT_IMG_MERCURYBUMP = 11,
    RT_IMG__COUNT = 12
};
enum { RT_WRAP_CLAMP = 0, RT_WRAP_AROUND = 1, RT_WRAP_MIRROR = 2 };        /* Image.h:17-19 */
typedef struct RtImageMap {         /* an Image (Image.h:140-156) as ReadBMP leaves it: linear floats, plane after plane, row 0 = top */
    int32_t width, height, channels;        /* channels: 1 or 3 */
    int32_t wrap_x, wrap_y;                 /* m_bWrapX, m_bWrapY */
    int32_t pad_;
    const float *pixels;                    /* [channels][height][width]; null = this slot was not loaded */
} RtImageMap;
/* Propagator function pointers (Solids.h:267); 0 = none. */
enum RtPropagatorId { RT_PROP_NONE = 0, RT_PROP_COLOREDGLASS = 1, RT_PROP_FOGGY = 2, RT_PROP_ATMOSPHERE = 3, RT_PROP__COUNT };

typedef struct RtNode {
    int32_t type;       /* RtNodeType */
    int32_t id;         /* creation serial given by the front end; stable across Optimize() */
    int32_t left;       /* child / left child index, -1 = null pointer */
    int32_t right;      /* right child index, -1 = null pointer */
    int32_t i0, i1, i2; /* see RtNodeType */
    int32_t link;       /* Optimize() result (Solids.cpp:957-983, 1489-1528): node index of the enclosing affine, -1 = none */
    int32_t data;       /* offset into RtSceneDesc.data (in doubles) */
    int32_t ndata;
} RtNode;

enum RtLightType { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1 };
typedef struct RtLight {        /* PointLight RayTracer2020.h:153-163, AreaLight Lights.h:20-30 */
    int32_t type;
    int32_t no_shadow;          /* nNoShadow */
    double  org[3];             /* pOrg */
    double  vA[3], vB[3];       /* AreaLight edge vectors; zero for point lights */
    float   flux[RT_NSPEC];
} RtLight;

enum RtCameraType { RT_CAM_PERSPECTIVE = 0, RT_CAM_STEREOGRAPHIC = 1, RT_CAM_ORTHOGRAPHIC = 2, RT_CAM_APERTURE = 3 };
typedef struct RtCamera {       /* Cameras.h:8-67 */
    int32_t type;
    int32_t pad_;
    double  org[3];
    double  focal;
    double  vX[3], vY[3];       /* ApertureCamera lens axes (already scaled by fApertureRadius) */
    double  focus_distance;
} RtCamera;

enum RtSamplerType { RT_SAMPLER_TEST = 0, RT_SAMPLER_FASTBLUE = 1, RT_SAMPLER_JITTER = 2, RT_SAMPLER_BLUENOISE = 3, RT_SAMPLER_ADAPTIVEBLUE = 4, RT_SAMPLER_DEBUG = 5 };
typedef struct RtSampler {      /* Sampling.h; Sampling.cpp:324-348, 474-548 */
    int32_t type;
    int32_t venera;             /* nVenera: vRGB instead of sRGB */
    const int32_t *qindex2;     /* g_BlueNoiseQuasiIndex2 as left by QIndex1() -- RT_BLUE_N ints (FASTBLUE only) */
    const int32_t *qindex3;     /* g_BlueNoiseQuasiIndex3 -- RT_BLUE_N ints (FASTBLUE only) */
    double  aux_init[6];        /* g_uCamera,g_vCamera,g_uLight,g_vLight,g_uMicroFacet,g_vMicroFacet as they stand
                                   when a sampler that never writes them (TestSamplingTile, JitterSamplingTile) runs */
    uint32_t prng_x, prng_c;    /* ML_nMarsagliaX, ML_nMarsagliaC (MonteCarlo.cpp:9-10) as they stand when Render is called:
                                   JitterSamplingTile draws its sample positions from that stream (Sampling.cpp:407-408) */
} RtSampler;

typedef struct RtSceneDesc {
    int32_t        n_nodes;
    int32_t        root;            /* psModel, -1 = none */
    const RtNode  *nodes;
    int32_t        n_data;
    int32_t        n_lights;
    const double  *data;
    const RtLight *lights;          /* plPointLights list order */
    int32_t        has_ambient;     /* plAmbient != 0 */
    float          ambient_flux[RT_NSPEC];
    RtCamera       camera;
    RtSampler      sampler;
    int32_t        n_images;        /* 0, or RT_IMG__COUNT when a planet texture is used */
    int32_t        pad_;
    const RtImageMap *images;       /* indexed by RtImageSlot */
} RtSceneDesc;

typedef struct RtRenderDesc {
    int32_t width, height;          /* Image size (im.m_nWidth, im.m_nHeight) */
    double  spp;                    /* fSamplesPerPixel */
    /* Multi-GPU sharding (no counterpart in the reference): this call produces output rows
       [row0,row1) only; the planes buffer still has full-image strides.  row1 <= 0 means all rows. */
    int32_t row0, row1;
} RtRenderDesc;

typedef struct RtStats {
    int64_t samples;                /* return value of SamplingTile::Render */
    int64_t rays_primary;           /* psModel->Intersect calls from VisualRay at depth 0 (Shade.cpp:422) */
    int64_t rays_shadow;            /* psModel->Intersect calls from LightRay (Shade.cpp:319) */
    int64_t rays_secondary;         /* psModel->Intersect calls from VisualRay at depth > 0 */
    int64_t hit_overflow;           /* rays whose hit list exceeded the per-ray arena (GPU only) */
    double  seconds;                /* wall time of the render call */
    double  seconds_trace;          /* of which: sample generation + intersect + shading */
    double  seconds_reconstruct;    /* of which: splat + resample + normalise */
    int64_t refined_blocks;         /* AdaptiveBlueSamplingTile: 20x20 blocks that got a second pass (their rays are counted above) */
    int64_t rays_fast;              /* of rays_primary: traced by k_fast (one chain, hits in registers); GPU only, measurement */
    int64_t rays_visual0;           /* of rays_primary: traced by k_visual (the interpreter); the rest ended in the generator's cull */
} RtStats;

#ifdef __cplusplus
}
#endif
#endif
