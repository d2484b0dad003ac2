/*
 * rt_amd.h -- C ABI of the MI355X render library (librt_amd.so, built from ray-tracer_amd/csrc/).
 *
 * This is the drop-in boundary for the reference's render hot path: everything behind
 *     int RayTracer::Render(Image &im, double fSamplesPerPixel)        RayTracer2020.cpp:33-38
 * (SamplingTile::Render -> CastFunction -> Camera::Emit -> RayTracer::VisualRay -> Solid::Intersect ...).
 * Plain pointers and sizes only; the caller owns every buffer it passes.  One rt_scene handle
 * belongs to one GPU (the device current when it was uploaded); handles are thread-compatible,
 * not thread-safe.  All functions return 0 on success, a negative RT_ERR_* code otherwise, and
 * rt_last_error() describes the last failure on the calling thread.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include "rt_scene_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,        /* malformed scene / arguments */
    RT_ERR_UNSUPPORTED = -2,    /* node / texture / sampler without a device path yet */
    RT_ERR_DEVICE = -3,         /* HIP error */
    RT_ERR_HIT_OVERFLOW = -4,   /* a ray's hit list exceeded the per-ray arena; the image is not valid */
    RT_ERR_NOMEM = -5
};

typedef struct rt_scene rt_scene;

const char *rt_last_error(void);

/* Flatten + upload a scene to the current HIP device.  Replaces the implicit "scene is a pointer
   graph reachable from RayTracer" of the reference (RayTracer2020.h:188-202); the description must
   be the graph as it stands after psModel->Optimize() (TestScene.cpp:1614).  The planet image maps a scene's textures read
   (RtSceneDesc.images: the file-static Images of TextureMaps.cpp:945-947 as Image::ReadBMP leaves them) are copied to the
   device here; a planet texture whose images are missing is RT_ERR_INVALID. */
int  rt_scene_upload(const RtSceneDesc *desc, rt_scene **out);
void rt_scene_free(rt_scene *s);

/* RayTracer::Render (RayTracer2020.cpp:33-38; SamplingTile::Render, RayTracer2020.h:182).
   planes: HOST float[3][height][width], layout i + W*(j + H*k) (Image.h:235), linear RGB.
   stats->samples is the reference's return value. */
int  rt_render(rt_scene *s, const RtRenderDesc *rd, float *planes, RtStats *stats);

/* Same, writing into DEVICE memory (hipMalloc'ed float[3][H][W]) on `hip_stream` (a hipStream_t, 0 =
   default stream) without the final device->host copy: what bench.py times (inputs resident in HBM)
   and what a multi-GPU caller gathers with RCCL. */
int  rt_render_device(rt_scene *s, const RtRenderDesc *rd, float *planes_dev, RtStats *stats, void *hip_stream);

/* rt_render_device for callers that cannot allocate device memory: it has NO counterpart in the reference and exists for
   code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments, same image (bit
   for bit), but every buffer the render needs -- wavefront queues, per-sample colours, reconstruction images, tables -- is
   carved out of one file-scope device array of rt_static_scratch_size() bytes (128 MiB; each GPU the library runs on holds
   it until the process ends) instead of being allocated: no hipMalloc / hipFree and no allocating library routine is called.
   A render whose buffers (about 28 bytes per sample + 2 KB per sample of the smallest pass) do not fit is rejected with
   RT_ERR_NOMEM and a message that says how much it needs -- render fewer rows per call or use rt_render_device.  Calls on
   one GPU share that GPU's array: calls that overlap there race.  Option "time_kernels" is ignored. */
int  rt_render_device_static(rt_scene *s, const RtRenderDesc *rd, float *planes_dev, RtStats *stats, void *hip_stream);
int64_t rt_static_scratch_size(void);

/* Output stage -- Image::WriteBMP (Image.cpp:416-474): ClampToGamut (Image.cpp:293-327) + GammaEncode (Image.h:206-220)
   against the table of SetGamma (Image.cpp:498-517; gamma = -1.0 is GAMMA_SRGB), 24-bit bottom-up BGR lines.
   rt_bmp_encode: planes HOST float[3][H][W] -> the complete file image (54-byte header of WriteHeaderBMP,
   Image.cpp:330-414, then the lines), file_bytes of rt_bmp_size(width, height) bytes.
   rt_bmp_encode_device: planes and pixel lines (size - 54 bytes) both in DEVICE memory, no header, on `hip_stream`. */
int64_t rt_bmp_size(int width, int height);
int  rt_bmp_encode(const float *planes, int width, int height, double gamma, unsigned char *file_bytes, int64_t cap);
int  rt_bmp_encode_device(const float *planes_dev, int width, int height, double gamma, unsigned char *pixels_dev, void *hip_stream);

/* Output stage -- Video (Video.cpp:8-80, YUV4MPEG2 "C444", 30 fps progressive).  rt_y4m_header: the stream header
   NewVideo writes (Video.cpp:33), returns its length (out == 0: length only); width and height must be multiples of 4
   (Video.cpp:14).  rt_y4m_encode_frame: Video::WriteFrame (Video.cpp:49-80) for HOST float[3][H][W] planes:
   "FRAME\n" + the Y, Cb, Cr planes, rt_y4m_frame_size(width, height) bytes; gamma = m_fGamma (Video.h:20, 2.0).
   rt_y4m_encode_device: planes and the 3*W*H plane bytes both in DEVICE memory, no "FRAME\n", on `hip_stream`. */
int  rt_y4m_header(int width, int height, char *out, int cap);
int64_t rt_y4m_frame_size(int width, int height);
int  rt_y4m_encode_frame(const float *planes, int width, int height, float gamma, unsigned char *frame_bytes, int64_t cap);
int  rt_y4m_encode_device(const float *planes_dev, int width, int height, float gamma, unsigned char *ycbcr_dev, void *hip_stream);

/* Host-only planning of a row-sharded render (no counterpart in the reference, which renders whole images):
   for output rows [rd->row0, rd->row1) of a FASTBLUE render, the supersample rows [out8[0], out8[1]) the vertical
   filter reads (Sampling.cpp:144-154) and the blue-noise tile rows [out8[2], out8[2] + out8[3]) whose samples can
   reach them through the bilinear splat; then nW, nH (Sampling.cpp:484-485) and the tile counts.  A rank that
   traces exactly those tiles needs no exchange with its neighbours.  Needs no GPU. */
int  rt_plan_rows(int sampler, const RtRenderDesc *rd, int *out8);

/* ---- parity probes: the same device code paths, exposed per stage (host buffers) ------------- */

/* Sample generation of the scene's sampling tile (Sampling.cpp:335-346 / 498-517): samples
   [first, first+n) in the reference's loop order.  out8: vx, vy (normalised image position) and
   g_uCamera, g_vCamera, g_uLight, g_vLight, g_uMicroFacet, g_vMicroFacet.  out_xy: position in the
   supersampled image (FASTBLUE) or pixel centre (TEST).  Returns total sample count in *total. */
int  rt_probe_samples(rt_scene *s, const RtRenderDesc *rd, int64_t first, int64_t n, double *out8, double *out_xy, int64_t *total);

/* Camera::Emit (Cameras.h:8-67) for n positions v2[2n] with aux dims aux6[6n] (may be null). */
int  rt_probe_camera(rt_scene *s, int n, const double *v2, const double *aux6, double *org3, double *dir3);

/* psModel->Intersect(ray) (RayTracer2020.h:72) for n rays: the full hit list, up to max_hits records
   per ray: t, normal (object space; after TransformNormals, Shade.cpp:264-284, when transform != 0),
   ids (RtNode.id) of the primitive, the material (-1 vacuum, -2 neutronium) and the surface
   (-3 default matte, -9 none). */
int  rt_probe_intersect(rt_scene *s, int n, const double *org3, const double *dir3, const double *aux6, int max_hits, int transform,
                        int *out_count, double *out_t, double *out_n3, int *out_leaf, int *out_mat, int *out_surf);

/* RayTracer::VisualRay (Shade.cpp:401-520) for n rays: 10-band radiance per ray. */
int  rt_probe_radiance(rt_scene *s, int n, const double *org3, const double *dir3, const double *aux6, float *out_spec10);

/* Per-sample RGB of a render (what is splatted), in loop order: float[3*samples]. */
int  rt_probe_sample_rgb(rt_scene *s, const RtRenderDesc *rd, float *out_rgb, int64_t max_samples, RtStats *stats);

/* Leaf functions the reference spot-checks on their own, same device code as the render path, no scene needed.
   rt_probe_noise: BandNoise(p, 3) (Texture.cpp:98-123), FractalNoise(p, 3, 5, 1.0f, 1.9f, 0) (:200-241) and Voronoi(p) (:334-374) at n
   points p3[3n] -- what TestBandNoise (Test.cpp:1635-1653) and TestVoronoi (Test.cpp:3798-3836) print.
   rt_probe_fresnel: FresnelLaw_Real (Shade.cpp:90-113) for indices (n1, n2) at n cosines -> out_F, out_fNT; and, when out_spec10 is
   given, Fresnel_Spectral (Shade.cpp:117-153) of the spectral metal `metal_luster` (MAT_COPPER .. MAT_TITANIUM) seen from a medium
   of index n1 -- the sweep of TestFresnel (Test.cpp:1467-1515); another luster: FresnelLaw_Complex with the real indices. */
int  rt_probe_noise(int n, const float *p3, float *out_band, float *out_fractal5, float *out_voronoi);
int  rt_probe_fresnel(int n, const double *cos_ne, double n1, double n2, int metal_luster, double *out_F, double *out_fNT, float *out_spec10);

/* BuildQuasiIndex (Patterns.cpp:1723-1736), RandomUnsigned (MonteCarlo.h:17-27): host-side integer helpers. */
void rt_build_quasi_index(int *out, int prime, int n);
void rt_random_unsigned(uint32_t state_xc[2], int n, uint32_t *out);
/* the state after ndraws calls of RandomUnsigned, without making them (what a render with JitterSamplingTile leaves behind) */
void rt_random_skip(uint32_t state_xc[2], uint64_t ndraws);

/* Measurement support for bench.py (no counterpart in the reference): sums, over those of the n rays that reach
   k_visual (rays that miss every box of the scene end in the generation kernel), of the ideal-culling event
   counts of SURVEY.md section 8d (leaf tests, affine transforms, box tests, merged hits, shaded boundaries, light
   terms); then the average duration (ms, HIP events on the render stream) and size (rays) of the k_visual
   launches made since the previous call while option "time_kernels" was 1; then how many of the n rays reached
   k_visual.  out9: 9 doubles. */
int  rt_probe_work(rt_scene *s, int n, const double *org3, const double *dir3, double *out9);

/* Measurement support for bench.py (no counterpart in the reference): out2[0] = total duration in ms (HIP events on the
   render stream) and out2[1] = number of the k_fast launches made since the previous call while option "time_kernels" was 1.
   k_fast traces the primary rays that can only touch bare chains of the root union region (RtStats.rays_fast of them). */
int  rt_probe_fast_time(rt_scene *s, double *out2);
/* Same for the k_mesh_cross launches that walk every ray of a level through the scene's mesh (option mesh_pre = 2):
   out3[0] = total ms, out3[1] = launches, out3[2] = rays walked. */
int  rt_probe_mesh_time(rt_scene *s, double *out3);

/* Tunables (no counterpart in the reference).  name: "chunk_samples" (samples per wavefront pass, default 67,104,000 = 41940 sampling tiles, at most 2^26; the queues of a pass take about 2 KB of HBM per sample -- 134 GB at the default -- and a pass is made smaller when the device has less free),
   "bvh" (0/1: conservative culling boxes, union-region BVH and fused chains; takes effect at rt_scene_upload),
   "root_cull" (0/1: finish rays that miss every box of the scene in the generation kernel; at rt_scene_upload),
   "fuse_splat" (0/1/2, default 2: FastBlue reconstruction gathers the splats and filters rows in one kernel, the supersampled
   image staying in LDS; 2: that kernel honours the need map (tiles that cannot matter are neither gathered nor filtered), deals
   pixels of equal tile position to a wave's lanes and reads the splat table's rows from LDS; 1: the first fused kernel, without
   the need map; 0: two kernels through memory; same result),
   "persistent", "lds_chains" (0/1: launch geometry / chain records in LDS, for A/B runs),
   "lds_tables" (0/1/2, default 1; 2 = the program and region tables alone when the rest does not fit: stage the scene's small tables -- program, boxes, affines, materials, surfaces, region tables -- in LDS
   next to the chain records when they fit; takes effect at rt_scene_upload),
   "tile_mask" (0/1: a pre-pass marks the sampling tiles no ray of which can touch a box of the scene; their samples end
   without being generated),
   "alt_chain" (0/1, default 1: a primary ray whose sampling tile lists a single child of the root region, a bare
   chain, runs that chain alone instead of the region; at rt_scene_upload; results do not depend on it),
   "tile_need" (0/1/2/3, default 3: FastBlue reconstruction skips the supersample pixels of tiles farther than twice the
   filter's reach from every tile that can hold a coloured sample -- every output pixel that could read them is 0 whatever
   they hold -- in the splat gather (1) and in the two resampling passes (2); 3: the gather does not even write them and the
   row pass takes them as the zeros they would hold; results do not depend on it),
   "tile_cand" (0/1/2, default 2: the same pass lists, per tile, the children of the root union region a ray through the
   tile can touch -- up to 16; 1: the generator tests those boxes instead of walking the BVH per sample, 2: so does the
   tracing kernel for primary rays; results do not depend on it),
   "full_waves" (1 / 2, default 2: waves per SIMD the kernels for the rare features -- tori, meshes, textures,
   propagators, spectral metals -- are built for; two hide the latency of mesh walks, one has no register spills; at
   rt_scene_upload),
   "cell_order" (0/1, default 1: the samples of a FastBlue tile are traced cell by cell, which needs chunk_samples to be a
   multiple of 1600; results do not depend on it),
   "full_tables" (0/1, default 1: those kernels also read chain records and scene tables from LDS when two of their
   workgroups fit a CU; at rt_scene_upload),
   "time_kernels" (0/1: HIP events around the k_visual, k_fast and k_mesh_cross launches, read back by rt_probe_work / rt_probe_fast_time / rt_probe_mesh_time),
   "fast_path" (0/1, default 1: primary rays that can only touch bare chains of the root union region -- two-root primitives
   without texture call-outs, propagators or spectral metals -- are traced by k_fast: no interpreter, no hit arena, hits in
   registers, and no frame where neither a shadow ray nor a child ray is needed; at rt_scene_upload; results do not depend on it),
   "fast_split" (0/1, default 1: the fast path as two kernels -- k_fast_hit: ray, chain, hits, where the walk stops -> a 44-byte
   boundary record per queue entry; k_fast_shade: light loop and RGB from the record -- each at a higher occupancy than the one
   kernel k_fast (0); results do not depend on it),
   "cull_f32" (0/1, default 1: the generator tests the boxes of a sample tile's list in single precision against boxes grown by
   1e-4 of the scene's scale; it can only keep more samples than the double test, which then miss in the kernel that takes them;
   at rt_scene_upload; results do not depend on it),
   "level_sync" (0/1, default 0: 1 makes every pass wait for the host after every level, also where the device-side
   bounds make that unnecessary; for A/B runs),
   "level_copy" (0/1, default 1: the counters the host reads at the end of a level -- frames, shadow rays, what the common
   kernels handed over -- come back in one copy into page-locked memory; 0: one copy per counter into pageable memory; for A/B runs),
   "matte_shape" (0/1, default 1: a scene none of whose surfaces is polished never makes a reflected or refracted ray, so its
   queues hold one frame per root sample and one shadow ray per light instead of four and two per light, and its passes are
   three times chunk_samples unless its levels may be launched directly (direct_full), where smaller passes decide better; at rt_scene_upload; results do not depend on it),
   "mesh_pre" (0/1/2, default 2: scenes with exactly one mesh instruction walk their rays through the mesh's BVH in a kernel of
   its own, k_mesh_cross -- 1: the rays the common kernels hand over, before the kernels for the rare features run; 2: every ray
   of a level, before the common kernels, which then read the crossings themselves; 0: the walk inside the kernels for the rare
   features; at rt_scene_upload; results do not depend on it),
   "direct_full" (0/1, default 1: a recursion level of a chunk whose rays the common kernels mostly handed over the last time the
   same chunk was rendered goes to the kernels for the rare features at once; results do not depend on it),
   "sched" (0/1/2, default 2 = for programs of 256 instructions and more: behind the first recursion level those kernels run, in
   every round of the interpreter, the lanes of a wave that sit at the smallest program counter -- every jump of a program goes
   forward, so the lanes ahead wait for the others to catch up and the lanes of a round execute one and the same instruction;
   the kernels for the rare features reserve 29 KB of private memory per lane; at rt_scene_upload; results do not depend on it),
   "rgb_sorted" (0/1, default 1: render calls whose reconstruction is the fused gather + row filter keep the colours of a sampling
   tile's 1600 samples in the order (supersample row, x) instead of sample order, so that what a row of the reconstruction gathers
   from a tile lies together in memory; rt_trace_samples always hands out sample order; results do not depend on it),
   "expect_empty" (0/1, default 1: a scene whose first level was empty in the call before -- every primary ray ended in the
   generator or in the fast path -- has that level shaded by a few looping workgroups and traced by a small launch instead of
   launches sized for the upper bound, which is all the host knows; results do not depend on it, nor on the expectation being right),
   "sort_rays" (0/1, default 1: the levels behind the first that go to the kernels for the rare features directly are traced in the order of a key --
   direction octant and dominant axis, then the Morton cell of the origin -- sorted on the device; results do not depend on it),
   "direct_pct_deep" (default 40: the share of a level's rays the common kernels must have handed over for a sorted level to go
   to the kernels for the rare features directly; 60 for the other levels),
   "pool_cache" (0/1, default 1: the queue pool of a scene that is freed is kept -- at most two per process -- and taken over by the
   next scene on the same device that needs one, instead of being given back and allocated again: an animation uploads a scene per
   frame; setting it to 0 gives the kept pools back at once),
   "direct_launches" (read-only: launches made under direct_full since the library was loaded). */
int  rt_set_option(const char *name, int64_t value);
/* the current value of a tunable, -1 (and rt_last_error) for an unknown name */
int64_t rt_get_option(const char *name);

#ifdef __cplusplus
}
#endif
#endif
