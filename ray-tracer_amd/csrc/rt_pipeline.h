/* rt_pipeline.h -- HBM-resident state of the wavefront pipeline (rt_kernels.hip) and of reconstruction. */
#pragma once
#include <stdint.h>
#include <stddef.h>

/* frame flags */
#define FR_BOUNDARY   1        /* the walk stopped at a boundary: R is multiplied by tr on resolve (Shade.cpp:503) */
#define FR_C1         2        /* has a reflection term (child frame or black) */
#define FR_C2         4        /* has a refraction term */
#define FR_FIN        8        /* has a trailing propagator sterisent term (Shade.cpp:512-516) */
#define FR_FINAL      16       /* MAT_NOSHADE early return: R is the result */
#define FR_SLOT2      32       /* this frame is its parent's refraction child */
#define FR_PERTURBED  64
#define FR_DONE       128      /* resolved and delivered by k_shade (no reflection / refraction terms): k_resolve skips it */
/* what k_visual did NOT store because k_shade / k_resolve can tell from the flag (bytes not written are the cheapest) */
#define FR_TR_ONE     (1 << 16) /* transmittance is 1 on all bands (no propagator on the way): f_tr not written */
#define FR_R_ZERO     (1 << 17) /* radiance gathered on the way is 0: f_R not written */
#define FR_R_AMBIENT  (1 << 18) /* ... is exactly the ambient term colour * E of the boundary (Lights.cpp:33-37): f_R not written */
#define FR_COLOR_MAT  (1 << 19) /* the boundary's colour is its material's (IDS_LUSTER holds luster | material << 8): f_color not written */
#define FR_DEPTH(f)        (((f) >> 8) & 0xFF)
#define FR_MAKE_DEPTH(d)   (((d) & 0xFF) << 8)
#define FR_KEEP_MASK       (0xFF00 | FR_SLOT2)

enum { GEO_P = 0, GEO_N = 3, GEO_E = 6, GEO_HN = 9, GEO_FNE = 12, GEO_F = 13, GEO_FNT = 14, GEO_ETA = 15, GEO_W2 = 16, GEO_COUNT = 17 };
enum { IDS_ROUGH = 0, IDS_LUSTER = 1, IDS_SHBASE = 2, IDS_SHMASK = 3, IDS_COUNT = 4 };
/* [CTR_SHADOW .. CTR_WORK_SS] are cleared together per level; CTR_WORK_*: next unclaimed item of the persistent kernels */
/* every counter sits on a cache line of its own (32 ints): they are hit by same-address atomics from every wave, and
   atomics on one line serialise in its L2 channel whichever word they touch */
#define CTR_LINE 32
/* CTR_SHADOW and CTR_SLOW_V are the two halves of one 64-bit word: k_visual bumps both with a single atomic (wave_alloc2) */
/* CTR_FAST: entries of the fast queue (k_generate_cull -> k_fast), CTR_WORK_F: k_fast's work counter; CTR_SNAP: values a
   one-thread kernel (k_level_snap) copies between the stages of a level so that the next kernels of the stream read their
   bounds from the device instead of waiting for the host: [0] the level's last frame + 1, [1] its shadow rays (clamped to
   the queue).  CTR_OVERFLOW is cleared once per render call, everything else once per pass (k_pass_begin). */
enum { CTR_FRAMES = 0 * CTR_LINE, CTR_OVERFLOW = 1 * CTR_LINE, CTR_SHADOW = 2 * CTR_LINE, CTR_SLOW_V = 2 * CTR_LINE + 1, CTR_SLOW_S = 4 * CTR_LINE,
       CTR_WORK_V = 5 * CTR_LINE, CTR_WORK_VS = 6 * CTR_LINE, CTR_WORK_S = 7 * CTR_LINE, CTR_WORK_SS = 8 * CTR_LINE, CTR_PROF = 9 * CTR_LINE,
       CTR_FAST = 10 * CTR_LINE, CTR_WORK_F = 11 * CTR_LINE, CTR_SNAP = 12 * CTR_LINE, CTR_WORK_SH = 13 * CTR_LINE,
       CTR_WORK_MV = 14 * CTR_LINE, CTR_WORK_MS = 15 * CTR_LINE,      /* k_mesh_cross over the visual / shadow slow list: cleared per level by the host */
       CTR_WORK_F2 = 16 * CTR_LINE,         /* k_fast_shade's work counter */
       CTR_COUNT = 17 * CTR_LINE };
#define CTR_LEVEL_CLEAR (CTR_WORK_SS + 1 - CTR_SHADOW)     /* ints from CTR_SHADOW up to and including CTR_WORK_SS */
#ifndef RT_CLAIM
#define RT_CLAIM 128        /* items a persistent wave claims per atomic: two 64-item passes.  Same box, 4K / 64 spp: 512 / 256 / 128 / 64 items =
                               S0 58.2 / 55.8 / 53.5 / 55.7 ms, S2 - / 478 / 471 / 520, S1 350 / 342 / 342 / 345: large claims leave waves idle at the end
                               of every launch, small ones queue up on the counter */
#endif

typedef struct RtPipe {
    int     fcap, scap, ncap;   /* frame, shadow-ray and root capacities */
    int     pad_;
    /* frames, SoA [field][fcap] */
    double *f_ray;              /* 6 */
    int    *f_parent, *f_root, *f_flags;
    float  *f_R, *f_tr, *f_color, *f_c1, *f_c2, *f_w1, *f_fin;     /* 10 each */
    double *f_geo;              /* GEO_COUNT */
    int    *f_ids;              /* IDS_COUNT */
    double *aux;                /* 6 x ncap: the distributed-sampling dimensions per root sample; only the pairs the scene needs
                                   after generation are written and read (RtDevScene.aux_light -> [2],[3]; aux_tex -> [4],[5]) */
    /* shadow rays, SoA [field][scap] */
    double *s_ray;              /* 6 */
    int    *s_light, *s_root;
    float  *s_E;                /* 10 */
    int    *slow_v, *slow_s;    /* frames / shadow rays the LITE kernels handed to the FULL ones: fcap, scap */
    int    *counters;           /* CTR_COUNT */
    /* the fast queue: primary rays that can touch ONE child of the root union region, a bare chain of a two-root primitive
       without rare features (RtDevScene.root_fast).  An entry is the sample (index within the pass) and the chain; k_fast
       recomputes the ray from the sample index, keeps the (at most three) hits in registers and, where no shadow ray or
       child ray is needed, shades and delivers the sample without ever making a frame.  2 x ncap ints: [k][chain]. */
    int    *fast_q;
    /* the fast path in two kernels (option fast_split): k_fast_hit finds the boundary a fast ray's walk stops at and leaves a
       44-byte record -- sample, distance, the hit's normal after the late transform, the materials on either side and the
       surface -- for k_fast_shade, which recomputes the ray from the sample index and shades.  SoA over ncap entries, entry e
       belongs to entry e of the fast queue: fr_k [1] (-1: that ray has no boundary), fr_t [1], fr_n [3], fr_ids [2] (material
       before | material entered << 16; surface).  Null = one kernel. */
    int    *fr_k; double *fr_t; double *fr_n; int *fr_ids;
    /* the mesh pre-walk (scenes with RtDevScene.mesh_pre_naff >= 0): for entry k of the slow list that the FULL kernel is
       about to work through, the triangle crossings of its ray with the scene's mesh -- mx_n[k] of them (-1: the walk ran out
       of stack, walk again in line), unsorted, crossing i at [i * mxcap + k]: distance and position in mesh_tris.  Written by
       k_mesh_cross between the LITE and the FULL launch, for the visual rays and then again for the shadow rays of a level. */
    int    *mx_n; double *mx_t; int *mx_q;
    int     mxcap;
    int     mx_all;             /* 1: k_mesh_cross ran over EVERY ray of the level before the LITE kernels (which then read the crossings
                                   themselves): a visual ray's record is entry f - mx_f0, a shadow ray's its own index; 0: entry = position
                                   in the slow list */
    int     mx_f0, pad3_;
    /* the sampling tiles' candidate lists (RtSampleGrid::tile_mask / tile_cand) for the PRIMARY rays of this chunk: root
       sample k of the chunk lies in tile (tile_first + k) / 1600.  Null = none. */
    const unsigned char  *tile_n;
    const unsigned short *tile_cand;
    int64_t               tile_first;
    /* hit records of the scheduled FULL kernels (ArenaT<true>, rt_core.hip.h): RT_ARENA contiguous 48-byte records per lane of the
       largest such launch (n_cu x RT_FULL_PER_CU x 256 lanes); null = those kernels are not used */
    void                 *arena_pool;
    const unsigned short *rgb_slot;     /* RtSampleGrid::rgb_slot of the chunk being traced (for the kernels that only get the pipe) */
    /* ray order for those kernels (k_sort_keys, rt_kernels.hip): keys in and out (2 x sort_cap), identity values (sort_cap), the
       origins' box (8 ints) and the radix sort's scratch; null = levels are traced in queue order */
    unsigned             *sort_keys;
    int                  *sort_vals, *sort_box;
    void                 *sort_temp;
    int                   sort_cap, pad4_;
} RtPipe;

typedef struct RtSampleGrid {   /* geometry of one render call's sample set */
    int     sampler;
    int     width, height;      /* output image */
    int     nW, nH;             /* supersampled image (FASTBLUE), Sampling.cpp:484-485 */
    int     tiles_x, tiles_y;   /* 40x40 blue-noise tiles covering it, Sampling.cpp:498-499 */
    int     tile_y0, tiles_y_local;     /* tile rows traced by this call (row sharding) */
    int     test_row0, test_rows;       /* TEST sampler: output rows traced by this call */
    int     jit_row0, jit_rows;         /* JITTER sampler: supersample rows traced by this call */
    int     jit_segs;                   /* JITTER: 64-pixel segments per supersample row */
    const uint32_t *jit_states;         /* JITTER: MWC state (x, c) at the start of every segment of the traced rows */
    int     pad_;
    /* BLUENOISE (Sampling.cpp:427-466): tiles of bn_sqrt output pixels at the accumulated positions bn_tx[tiles_x], bn_ty[tiles_y] */
    double  bn_sqrt;
    const double *bn_tx, *bn_ty;
    /* ADAPTIVEBLUE (Sampling.cpp:562-694): 20x20 supersample-pixel blocks, ad_bx per row; block rows [ad_bj0, ad_bj1] are
       analysed by this call; ad_list[n] = refined block ids in the reference's order (set for the refinement pass only),
       ad_ord[block] = position in that list or -1 */
    int     ad_bx, ad_by, ad_bj0, ad_bj1;
    const int *ad_list, *ad_ord;
    /* FASTBLUE: one byte per traced tile (tx * tiles_y_local + local ty), 0 = no ray through the tile can touch a box of
       the scene (k_tile_mask): k_generate_cull finishes its 1600 samples without generating them.  Null = not used. */
    const unsigned char *tile_mask;
    /* with the mask: tile_mask[t] = n in 1..RT_TILE_CAND means tile_cand[t * RT_TILE_CAND + 0..n-1] are ALL the children
       of the root union region (ascending) whose box a ray through the tile can touch, so such a ray is tested against
       those boxes instead of walking the BVH; 255 = too many or not applicable: walk.  Null = no lists. */
    const unsigned short *tile_cand;
    /* FASTBLUE reconstruction: one byte per traced tile, 0 = no supersample pixel of the tile can matter: every output
       pixel whose filter support reaches it sees black samples only, so its value is 0 whatever the weights are
       (k_tile_need: the mask dilated by twice the filter's reach).  k_splat_gather writes zeros there.  Null = all needed. */
    const unsigned char *tile_need;
    /* FASTBLUE: the order in which the 1600 samples of a tile are TRACED: sorted by the 8x8-sample cell of the tile they
       fall into (25 cells of about 64 samples), so that the 64 rays of a wave come from one or two neighbouring cells
       instead of from all over the tile.  Which sample a work item is changes, nothing else does: results are stored
       by sample index.  Null = trace in index order (chunks that do not start and end on tile boundaries). */
    const unsigned short *fb_order;
    /* FASTBLUE render calls whose reconstruction is k_splat_rows_need: where sample i of a tile keeps its colour in the tile's 1600
       slots of the sample buffer -- the samples sorted by the supersample row they fall into, then by x, so that what a row of the
       reconstruction gathers from a tile lies in two short runs instead of all over the tile's 19 KB.  Null = slot i (every other
       caller: the sample buffer is then in sample order, which is what rt_trace_samples hands out).  Needs fb_order (tile-aligned chunks). */
    const unsigned short *rgb_slot;
    int     need_lazy;                  /* the passes behind the splat gather take the pixels of tiles with tile_need == 0 as zeros without reading
                                           them (RtNeedMap::lazy): the gather does not write them */
    int     mask_skips_rgb;             /* the caller's reconstruction honours the mask: masked tiles' (black) sample colours are neither written nor read */
    double  size, adjust_x, adjust_y;   /* fSize, vAdjust, Sampling.cpp:490-496 */
    double  inv_size;                   /* 1 / size, for the approximate (culling-only) sample position */
} RtSampleGrid;

typedef struct RtSplatEntry { double bx, by; int i, pad_; } RtSplatEntry;      /* a candidate sample: g_rgvBlueNoise[i] * BLUE_NSQRT (both coordinates) and i */
typedef struct RtSplatTable {   /* contributors of each pixel of a (40+2)^2 tile neighbourhood, CSR */
    const int *offsets;         /* 42*42 + 1 */
    const int *entries;         /* candidate sample indices, ascending = the reference's splat order */
    /* for k_splat_rows_need, which stages the one or two cell rows a supersample row needs in LDS: the same entries with the sample's
       position in the tile spelled out (no second look-up), the positions of every tile row ordered by the number of splats they
       receive (most first) [40][40], and the largest number of entries a cell row holds */
    const RtSplatEntry *entries_pos;
    const unsigned char *order;
    int max_row_entries, pad_;
} RtSplatTable;

typedef struct RtTapTable {     /* ComputeTaps / Resample, Sampling.cpp:86-184 */
    const double *taps;
    const int    *first, *count, *offset;
    int           uniform;      /* every destination has the same number of taps with bit-identical values (integer ratios) */
    int           max_count;
    double        wsum_uniform; /* uniform: the sum of the taps in tap order (what every destination's own running sum comes to, Sampling.cpp:171) */
} RtTapTable;
