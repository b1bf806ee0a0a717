/* rt_hip.h -- C-ABI of the MI355X (gfx950) path-tracing shim, librt_hip.so.
 *
 * This is the drop-in boundary for the reference's hot path: everything
 * render() (gue-ni/raytracer.c raytracer.c:176-223) does per pixel --
 * get_camera_ray :375-384, trace_path :482-554, intersect :393-464,
 * intersect_sphere :77-118, intersect_triangle :120-174,
 * calculate_surface_normal :42-45, the sampling RNG :227-253, reflect /
 * checkered_texture / refract :349-391, sample accumulation and the gamma-5
 * tonemap :212-220 -- runs on the device behind the entry points below.
 * Plain C: pointers and sizes only, no C++ or torch types, no exceptions.
 * The reference has no FFI layer of its own (its API is raytracer.h); the
 * host library libraytracer_amd.so implements raytracer.h on top of this
 * header, and INTEGRATION.md shows the binding a maintainer of the reference
 * would add to call it directly.
 *
 * Conventions
 *   - every function returns 0 on success or a negative RT_HIP_E* code;
 *     rt_hip_last_error() gives the message (thread-local).
 *   - "d_" pointers are device memory on the scene's device; "h_" are host.
 *   - a stream argument is a hipStream_t passed as void* (NULL = the null
 *     stream); calls taking a stream are asynchronous on it.
 *   - there is NO CPU fallback anywhere: without a usable GPU every entry
 *     point that needs one fails with RT_HIP_ENODEV.
 *
 * Image decomposition: the image is cut into 8x8-pixel tiles, numbered
 * row-major (tiles_x = ceil(width/8)).  One call renders the tiles
 *     t = tile_first + k * tile_stride,  k = 0 .. tile_count-1
 * into a COMPACT tile-major buffer: tile k occupies floats
 * [k*192, (k+1)*192) as 64 pixels (row-major inside the tile) x RGB.  With
 * tile_first = rank, tile_stride = world size this is the interleaved
 * multi-GPU partition; rt_hip_untile() scatters a (gathered) compact buffer
 * back to a row-major image.  Pixels of edge tiles that fall outside the
 * image are written as zeros and skipped by rt_hip_untile().
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_HIP_TILE 8        /* tile edge, pixels */
#define RT_HIP_TILE_PIXELS 64

enum
{
  RT_HIP_OK = 0,
  RT_HIP_ENODEV = -1,   /* no usable HIP device */
  RT_HIP_EINVAL = -2,   /* bad argument */
  RT_HIP_ENOMEM = -3,   /* host or device allocation failed */
  RT_HIP_ERUNTIME = -4, /* a HIP / RCCL runtime call failed */
  RT_HIP_ELIMIT = -5,   /* scene exceeds what the kernel supports */
  RT_HIP_ECANCELLED = -6, /* rt_hip_render_image stopped early on the cancel flag; output is partial */
};

/* Layout-identical to the reference's Object (raytracer.h:104-111): 88 bytes. */
typedef struct
{
  uint32_t flags; /* M_DEFAULT 2 | M_REFLECTION 4 | M_REFRACTION 8, | M_CHECKERED 16 */
  double radius;
  double center[3];
  double color[3];
  double emission[3];
} RtHipSphere;

/* Layout-identical to the reference's Vertex (raytracer.h:61): 40 bytes. */
typedef struct
{
  double pos[3];
  double tex[2];
} RtHipVertex;

/* A triangle mesh with its material (the MeshObject extension of
 * include/raytracer.h; reference raytracer.h:77-81 + :95-102).
 * vertices: host pointer, 3*num_triangles entries, unindexed. */
typedef struct
{
  uint32_t flags;
  double color[3];
  double emission[3];
  size_t num_triangles;
  const RtHipVertex *vertices;
} RtHipMesh;

/* Layout-identical to the reference's Camera (raytracer.h:121-124): 96 bytes. */
typedef struct
{
  double position[3];
  double horizontal[3];
  double vertical[3];
  double lower_left_corner[3];
} RtHipCamera;

typedef struct
{
  int32_t width, height; /* pixels */
  int32_t samples;       /* per pixel (Options.samples) */
  int32_t max_depth;     /* the reference's compile-time MAX_DEPTH (raytracer.h:25) */
  uint64_t seed;         /* stream key, see rt_rng.h */
  uint32_t tile_first, tile_stride, tile_count;
  uint32_t integrator;   /* RT_HIP_TRACE_PATH (0, the default) or RT_HIP_CAST_RAY */
} RtHipParams;

/* The two sides of the `#if 1` in the reference's render() (raytracer.c:207-211): the path
 * tracer trace_path (:482-554), which is what the reference ships, and cast_ray (:556-641),
 * the Whitted-style integrator it keeps compiled next to it (one fixed point light, Phong
 * shading, shadow rays, mirror / "refraction" children; no random draws beyond the camera
 * jitter).  With RT_HIP_CAST_RAY, RT_HIP_STAT_RAYS counts cast_ray calls (:558) and
 * RT_HIP_STAT_CASTS counts scene scans, i.e. the primary and the shadow scan of every hit. */
enum
{
  RT_HIP_TRACE_PATH = 0,
  RT_HIP_CAST_RAY = 1
};

/* counters accumulated (+=) by a render call */
enum
{
  RT_HIP_STAT_RAYS = 0,  /* reference ray_count: trace_path calls (raytracer.c:484) */
  RT_HIP_STAT_CASTS = 1, /* rays that ran the scene scan = "ray-bounces" */
  RT_HIP_STAT_TESTS = 2, /* reference intersection_test_count (raytracer.c:79,122) */
  RT_HIP_STAT_SAMPLES = 3,
  RT_HIP_NSTATS = 4
};

typedef struct RtHipScene RtHipScene; /* opaque, device-resident */

/* ---- device / errors ---------------------------------------------------------- */

int rt_hip_device_count(void);
const char *rt_hip_last_error(void);
/* name and compute-unit count of a device (name_cap bytes incl. NUL) */
int rt_hip_device_info(int device, char *name, size_t name_cap, int *compute_units);

/* ---- scene: replaces the Object[] argument of render() ------------------------- */

/* Uploads spheres and meshes to `device` in the kernel's layout.  Object ids
 * follow the reference's scan order: spheres 0..n_spheres-1, then meshes. */
int rt_hip_scene_create(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes,
                        size_t n_meshes, int device, RtHipScene **out_scene);
void rt_hip_scene_destroy(RtHipScene *scene);
int rt_hip_scene_device(const RtHipScene *scene);
size_t rt_hip_scene_primitives(const RtHipScene *scene); /* spheres + triangles */
/* diagnostic: how many triangles the scene marked as HULL FACETS -- every triangle of the scene lies on the inner
 * side of their plane, the stored normal (calculate_surface_normal, raytracer.c:42-45) pointing outward (*n_plus)
 * or inward (*n_minus).  A bounce that leaves such a facet on its outer side cannot meet a triangle, so the
 * hierarchy kernels do not walk it (pt_device.h).  0 / 0 for more than 65,536 triangles (the marking is quadratic). */
int rt_hip_scene_hull_facets(const RtHipScene *scene, uint32_t *n_plus, uint32_t *n_minus);
/* name of the render kernel a launch of this scene takes (the family is picked by scene content:
 * triangles, table size, M_CHECKERED / M_REFRACTION materials) -- for profiles and bench lines */
const char *rt_hip_kernel_name(const RtHipScene *scene, uint32_t integrator);

/* ... and of the kernel the calling thread's last rt_hip_render_tiles / _chunked call actually launched: the launch's own
 * facts (samples x depth against the windowed sums of the M_REFRACTION kernels, the width of the pending-ray pool that could
 * be had) can name another row of the pick table than the scene alone does. */
const char *rt_hip_last_launch_kernel(void);

/* The kernel family by index (0 .. rt_hip_kernel_count() - 1): name, and how many render launches of it this PROCESS has
 * made (*launches, may be NULL) -- what a test run actually exercised.  NULL beyond the family. */
int rt_hip_kernel_count(void);
const char *rt_hip_kernel_launches(int index, uint64_t *launches);

/* The pick table itself, without a device: which kernel a launch of a scene of this class takes.  A class is what the family is
 * split by (pt_kernel.hip, pt_pick_table): counts decide staging and the mesh form, the flags the material code. */
typedef struct
{
  uint32_t integrator;              /* RT_HIP_TRACE_PATH | RT_HIP_CAST_RAY */
  uint32_t n_spheres, n_meshes, n_triangles;
  uint32_t any_checker, any_refract, any_mirror_glass; /* materials: M_CHECKERED, M_REFRACTION, M_REFLECTION | M_REFRACTION on one object */
  uint32_t wide_range;              /* a centre or radius beyond 1e17 */
  uint32_t mesh_round;              /* the triangles' bounding sphere is no larger a target than their box */
  int32_t samples_per_chunk, max_depth;
  uint32_t have_park_ws;            /* the parked-walk workspace could be allocated */
  uint32_t wide_pend_ok;            /* the pending-ray pool could be had at 4 x 512 stacks per slot */
} RtHipSceneClass;
const char *rt_hip_kernel_for_class(const RtHipSceneClass *scene_class);

/* Device-side failures of the render launches on `device` since the last call: sticky bits, read and cleared here.  A
 * workgroup that cannot get a slot of a per-device pool renders nothing; its tile reads NaN (bytes 255) -- which is also what
 * a legitimate NaN sample gives (raytracer.c:218-220), so the pixel values cannot tell the caller: this can.  Call it after
 * synchronising the streams launched on.  Returns RT_HIP_OK with *flags = 0, or RT_HIP_ERUNTIME with the reason in
 * rt_hip_last_error().  rt_hip_render_image() checks it itself and returns the error. */
enum
{
  RT_HIP_FAIL_PEND_SLOT = 1, /* no free slot in the pending-ray pool (two-child materials) */
  RT_HIP_FAIL_PARK_SLOT = 2  /* no free slot in the parked-walk workspace (mesh hierarchies) */
};
int rt_hip_launch_status(int device, uint32_t *flags);

/* ---- the hot path: replaces the loop nest of render() (raytracer.c:184-222) ---- */

/* d_tiles_rgb : tile_count*192 floats  (linear per-pixel sample mean)
 * d_tiles_rgb8: tile_count*192 bytes   (gamma-5 tonemap, raytracer.c:218-220); may be NULL
 * d_stats     : RT_HIP_NSTATS uint64 accumulators; may be NULL */
int rt_hip_render_tiles(const RtHipScene *scene, const RtHipCamera *camera, const RtHipParams *params,
                        float *d_tiles_rgb, uint8_t *d_tiles_rgb8, uint64_t *d_stats, void *stream);

/* The same render with every tile's samples split over `sample_chunks` workgroups (finer
 * work units: matters when a GPU holds few tiles, e.g. 1/8 of a 1080p frame).  Partial sums
 * are exact integers, so the image is bit-identical for every sample_chunks.  d_workspace:
 * rt_hip_chunk_workspace_bytes(tile_count) bytes of device memory (may be NULL when
 * sample_chunks == 1); it is cleared, filled and resolved on `stream`.
 * rt_hip_suggest_chunks() returns a good value for the scene's device. */
size_t rt_hip_chunk_workspace_bytes(uint32_t tile_count);   /* enough for any scene */
/* ... for this scene: scenes without M_REFRACTION need a sixth of it (plain fixed-point sums; the others keep windowed sums) */
size_t rt_hip_scene_chunk_workspace_bytes(const RtHipScene *scene, uint32_t tile_count);
uint32_t rt_hip_suggest_chunks(const RtHipScene *scene, uint32_t tile_count, int32_t samples);
/* ... knowing the launch's max_depth: scenes with M_REFRACTION need samples_per_chunk x 2^(max_depth + 1) <= 2^30 for the pooled
 * and parked-walk kernels (their windowed pixel sums) -- the suggestion is at least that many chunks.  A launch that gets fewer
 * (or no workspace) and does not fit runs on the static kernel of the family, three times slower on a glass mesh: the image is
 * the same.  rt_hip_render_tiles_chunked raises a too-small chunk count itself whenever a workspace was handed over. */
uint32_t rt_hip_suggest_chunks_depth(const RtHipScene *scene, uint32_t tile_count, int32_t samples, int32_t max_depth);
int rt_hip_render_tiles_chunked(const RtHipScene *scene, const RtHipCamera *camera, const RtHipParams *params,
                                uint32_t sample_chunks, void *d_workspace, float *d_tiles_rgb,
                                uint8_t *d_tiles_rgb8, uint64_t *d_stats, void *stream);

/* Scatter a compact tile buffer into row-major images (either output may be
 * NULL together with its input). */
int rt_hip_untile(const float *d_tiles_rgb, const uint8_t *d_tiles_rgb8, int32_t width, int32_t height,
                  uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count, float *d_image_rgb,
                  uint8_t *d_image_rgb8, void *stream);

/* ---- self-test hook --------------------------------------------------------------- */

/* Evaluates one of the kernel's exact-arithmetic building blocks on host arrays (n values
 * each) so that tests can compare it bit for bit with IEEE results computed on the host:
 * op 0 sqrt without range scaling (valid for 0 or >= 2^-767), 1 quotient by a small integer
 * through its reciprocal, 2 library sqrt, 3 IEEE division, 4 the fused r*2^-30 - 1 mapping,
 * 5 reciprocal without range scaling (valid for 2^-500 <= a <= 2^500). */
int rt_hip_selftest_math(int op, const double *h_a, const double *h_b, double *h_out, size_t n, int device);

/* Runs the render kernels' own exact primitive tests (intersect_sphere raytracer.c:77-118,
 * intersect_triangle :120-174 as the device states them) and their conservative phase-1
 * filter on host arrays, one GPU lane per case, so that known-answer vectors reach the device
 * code itself.  kind 0: h_prims = n x 4 (cx cy cz radius); kind 1: h_prims = n x 9 (v0 v1 v2
 * positions); h_rays = n x 6 (origin, direction).  Case i = ray i against primitive i:
 * h_hit[i] 0/1 and h_tuv[3i..] = t (DBL_MAX on a miss) and, for triangles, the barycentric
 * u, v (a texture coordinate is st0 (1-u-v) + st1 u + st2 v, raytracer.c:154-167).
 * h_keep[3i + f], f = 0..2: 64-bit masks of the filter's three forms (f = 0: spheres, sign tests
 * from LDS; triangles, the per-lane fp32 Moeller-Trumbore pre-test that follows the bounding-sphere
 * filter in small scenes; f = 1: compares from LDS; f = 2: compares by scalar loads) for ray i
 * against the 64 primitives of its block [64 (i/64), +64): bit j set = primitive 64 (i/64) + j
 * is kept.  The filter is built for ray origins within near_R (rays beyond keep everything),
 * as rt_hip_render_tiles builds it for a camera.  |centre|, |radius| <= 1e17. */
int rt_hip_selftest_intersect(int kind, const double *h_rays, const double *h_prims, size_t n, double near_R,
                              uint8_t *h_hit, double *h_tuv, uint64_t *h_keep, int device);

/* Fault injection: from now on the named optional device allocations of the shim behave as if hipMalloc had failed, so that
 * the fallback rows of the pick table (no parked-walk workspace: the lane-waiting kernels; no pending-ray pool of 4 x 512 stacks
 * per slot: the static kernel of the family) are reachable on a device with 288 GB.  0 switches it off.  A scene that has
 * already met its workspace keeps it. */
enum
{
  RT_HIP_FAIL_ALLOC_PARK_WS = 1,
  RT_HIP_FAIL_ALLOC_WIDE_PEND = 2
};
void rt_hip_selftest_fail_alloc(uint32_t mask);

/* How many slots per XCD the two per-device pools get on `device` (the parked-walk workspace, the pending-ray pool): CUs per
 * XCD x the most workgroups of any slot-taking kernel a CU can hold, + 25 %, rounded up to a multiple of 32. */
int rt_hip_selftest_pool_slots(int device, uint32_t *park_slots_per_xcd, uint32_t *pend_slots_per_xcd);

/* Launches n_workgroups one-wave workgroups; h_counts[x] = how many of them read HW_REG_XCC_ID == x
 * (bits 3:0).  The parked-walk kernels partition their workspace by that id (every owner a slot ever
 * has must sit behind the same L2): on an MI355X the counts must be spread over ids 0..7. */
int rt_hip_selftest_xcc(uint32_t n_workgroups, uint32_t h_counts[16], int device);

/* ---- convenience for C hosts: whole image, host buffers, synchronous ----------- */

/* Cooperative cancellation of rt_hip_render_image(): while a flag is registered, long frames
 * are rendered in slabs and *flag is polled between them (set it from a signal handler).  On
 * cancellation the finished tiles are still gathered and copied out, the rest of the image is
 * zero, and the call returns RT_HIP_ECANCELLED.  NULL unregisters. */
void rt_hip_set_cancel_flag(const volatile int *flag);

/* Renders width x height on n_devices GPUs of this process (tiles interleaved
 * over devices: logical device g renders tiles g, g + n_devices, ...; the compact tile buffers
 * are gathered onto logical device 0, scattered to the row-major image there and copied to the
 * host).  Calls are serialised (one frame at a time).  A segment travels by grouped ncclSend /
 * ncclRecv over cached communicators when its device differs from the root's, and as a
 * device-to-device copy when it does not (see rt_hip_set_device_map).
 * n_devices > 1 is EXPERIMENTAL in one respect only: its indexing, slabs, cancellation, caching
 * and counters run in the GPU tests at 2, 3 and 8 LOGICAL devices mapped onto one GPU, and its
 * RCCL calls run there with one rank (RT_HIP_FORCE_COMM=1) -- but as of this writing no machine
 * with two physical GPUs has run it; `bench.py --gpus N` exercises it in a child process whenever
 * N > 1 GPUs are present and compares its frame with the one-device frame.
 * h_image_rgb (w*h*3 floats) and
 * h_image_rgb8 (w*h*3 bytes) may each be NULL.  h_stats: RT_HIP_NSTATS values,
 * overwritten.  kernel_seconds: device time of the render kernels (max over
 * devices), may be NULL.  params->tile_* are ignored. */
int rt_hip_render_image(const RtHipSphere *spheres, size_t n_spheres, const RtHipMesh *meshes,
                        size_t n_meshes, const RtHipCamera *camera, const RtHipParams *params,
                        int n_devices, float *h_image_rgb, uint8_t *h_image_rgb8, uint64_t *h_stats,
                        double *kernel_seconds);

/* rt_hip_render_image() keeps what it built -- per-device scenes, streams, tile buffers, the RCCL
 * communicators -- and reuses it while the device count, the image size and the scene's bytes
 * stay the same (a caller rendering frame after frame re-creates nothing).  This releases it, and
 * with it the per-device pool of pending-ray stacks that scenes with two-child materials
 * (M_REFRACTION under trace_path, M_REFLECTION | M_REFRACTION under cast_ray) render with;
 * rt_hip_cache_builds() counts how often a context had to be (re)built (for tests). */
void rt_hip_release_cache(void);
uint64_t rt_hip_cache_builds(void);

/* What the shim holds on `device` besides scenes and frames, bytes (0: not allocated): the parked-walk workspace (scenes with a
 * mesh hierarchy; freed with the last such scene) and the pending-ray pool (two-child materials; grows to the deepest and widest
 * launch, is rebuilt to fit after 16 launches in a row needed at most a quarter of a pool above 1 GB, freed by
 * rt_hip_release_cache()).  INTEGRATION.md has the sizes. */
int rt_hip_pool_bytes(int device, size_t *park_ws_bytes, size_t *pend_pool_bytes);

/* Where the host time of the last rt_hip_render_image() went, seconds: [0] context (scene compare; on a rebuild: upload,
 * buffers, workspaces, communicators), [1] launches + kernels + gather + scatter until every stream is idle, [2] the frame,
 * bytes and counters over PCIe. */
void rt_hip_last_image_phases(double seconds[3]);

/* Logical -> physical device map of rt_hip_render_image(): with a map of n entries, n_devices may be up to n and logical
 * device g runs on HIP device map[g]; entries may repeat.  Logical devices that share a physical one keep separate scenes,
 * streams and tile buffers (their kernels run concurrently on it); the partition, the slabs, the gather's slot arithmetic,
 * the per-segment scatter and the counter sums are the code that runs with n distinct GPUs, so a one-GPU machine executes
 * the whole n_devices > 1 path (map = {0, 0, 0}) and the image is bit-identical to n_devices = 1.  n = 0 removes the map
 * (logical = physical).  The environment variable RT_HIP_DEVICE_MAP="0,0,0", read at the first frame, sets the same map for
 * hosts that cannot call this (the reference's main.c behind libraytracer_amd.so).  Changing the map rebuilds the cached
 * context at the next frame.  Replaces nothing in the reference (its one parallel construct is the `omp parallel for` of
 * raytracer.c:184-185). */
int rt_hip_set_device_map(const int *map, int n);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_H */
