/* pt_device.h -- device-side data layout shared by the kernels and the shim.
 *
 * HBM layout of a scene (RtHipScene), built once by rt_hip_scene_create() as one blob:
 *
 *   entry_src  [n_spheres + n_triangles] x 6 f64 : cx, cy, cz, R2, |c|, Rb   (scan order)
 *       sphere  : centre, radius*radius (the product raytracer.c:87 forms per test; forming
 *                 it once is the same double), |centre|, 0
 *       triangle: centre / squared radius / radius of a bounding sphere (filter only)
 *       The first four doubles of the sphere records are what the exact test and the
 *       normal use; they are staged in LDS per workgroup (scenes beyond the 24 KiB staging
 *       budget -- and sphere-only scenes beyond ~85 spheres by preference, pt_prefer_streaming --
 *       read the compact copy geom4 from memory instead).
 *   filt       [ceil(entries/2)] x 6 f32x2 (+ [n_triangles] x 16 f32, see pt_filt_bytes) : cx cy cz r2_hi neg_tol kq, two primitives
 *       per f32x2 -- the packed-fp32 phase-1 filter table (pt_build_filter).  Its thresholds depend
 *       on the camera distance (near_R), so the shim keeps one table set per (scene, near_R),
 *       built once and immutable afterwards; staged in LDS by small scenes, otherwise read with
 *       wave-uniform indices through the constant address space, i.e. by scalar loads
 *       (pt_filter.h, ConstPair: as plain global loads they had been compiled to vector loads).
 *   material   [n_spheres + n_meshes] x 8 f64    : prob, albedo_rr xyz, emission xyz, flags
 *       prob      = MAX(color) (raytracer.c:497)
 *       albedo_rr = color * (1/prob), the albedo after a survived Russian roulette (:500);
 *                   same operands, same two operations as the reference performs per
 *                   bounce, so the same doubles.
 *   color_raw  [n_spheres + n_meshes] x 3 f64    : the colours as given; cast_ray
 *                   (raytracer.c:574) shades with them, trace_path never reads them.
 *   tri_geom   [n_triangles] x 9 f64 : v0, edge1 = v1-v0, edge2 = v2-v0
 *       (raytracer.c:132-133 forms the edges per test; precomputed = same).
 *   tri_normal [n_triangles] x 3 f64 : calculate_surface_normal(v0,v1,v2)
 *       (raytracer.c:42-45), read only for the winning triangle.
 *   tri_tex    [n_triangles] x 6 f64 : st0, st1, st2 (read only for a winning triangle of an
 *       M_CHECKERED mesh).
 *   tri_object [n_triangles] u32     : material slot of the owning mesh; bits 31 / 30 (PT_HULL_PLUS / PT_HULL_MINUS,
 *       set on the device by pt_build_hull_flags): every triangle of the scene lies on the inner side of this
 *       triangle's plane, the stored normal pointing out / in -- a ray that leaves such a HULL FACET on its outer
 *       side cannot meet a triangle again.
 *   bvh_src    [n_bvh_nodes] x 16 f64 : a binary bounding-volume hierarchy over the triangles
 *       (built for every scene that has any; the small-scene kernels scan triangles through
 *       the flat filter and ignore it).  A node holds the boxes of its TWO children
 *       (child 0: min xyz, max xyz; child 1: the same) and their references packed as two u32
 *       in double 12: an inner child is its node index, a leaf child is
 *       PT_BVH_LEAF_FLAG | first << PT_BVH_COUNT_BITS | count and covers bvh_tri[first .. first+count).  Node 0
 *       is the root; the tree is balanced (median splits), depth <= PT_BVH_STACK.
 *   bvh_nodes  [n_bvh_nodes] x 16 u32/f32 : the fp32 copy for one near_R (pt_build_bvh): the six
 *       planes as (child 0, child 1) pairs -- min x, max x, min y, max y, min z, max z -- so one
 *       packed-fp32 instruction serves both children; boxes widened by a bound on the fp32
 *       ray/box arithmetic; then the two references.
 *   bvh_tri    [n_triangles] u32     : triangle indices in leaf order.
 */
#ifndef PT_DEVICE_H
#define PT_DEVICE_H

#include <stdint.h>

#define PT_TILE 8
#define PT_TILE_PIXELS 64
#define PT_SLICES 4         /* sample slices per pixel inside one wavefront */
#define PT_BLOCK 256        /* 4 wavefronts: 64 pixels x 4 slices */
#define PT_MAT_STRIDE 8     /* doubles per material record */
#define PT_FILT_LDS_MAX 256  /* primitives up to which the filter table is also staged in LDS */
#ifndef PT_BVH_LEAF
#define PT_BVH_LEAF 15        /* max triangles per BVH leaf (< 2^PT_BVH_COUNT_BITS); config 5 at 4K x 256 spp: 2: 435 ms,
                               * 3: 428, 4: 428, 7: 420 (round 1's kernel).  With the pipelined fp32 pre-test at the leaves
                               * (round 2: a leaf visit costs one exposed round trip whatever its size) at 64 / 256 spp:
                               * 2: 88.3 / 326.7 ms, 4: 86.5 / 321.0, 7 (5 per leaf in config 5): 71.9 / 268.2,
                               * 15 with PT_BVH_COUNT_BITS 4 (10 per leaf): 71.4 / 266.8, 31 with 5 bits (20 per leaf): 73.2 / 273.9.
                               * End of round 3 (node visits and pre-tests trimmed, the walk waits for its node fetches more
                               * than it computes): 4: 70.2 / 261.1 ms, 7: 58.5 / 219.3, 15: 58.4 / 217.7, 31: 60.3 / 225.3 -> 15 */
#endif
#define PT_BVH_NODE_WORDS 16  /* 64-byte device node: 6 f32x2 planes (child 0, child 1) + 2 refs + pad */
#define PT_BVH_SRC_DOUBLES 16 /* fp64 source node: 2 x (min xyz, max xyz), refs, pad */
#ifndef PT_BVH_COUNT_BITS
#define PT_BVH_COUNT_BITS 4 /* bits of a leaf reference that hold its triangle count (PT_BVH_LEAF < 2^bits) */
#endif
#define PT_BVH_LEAF_FLAG 0x80000000u
#define PT_BVH_STACK 24       /* per-lane traversal stack (LDS): tree depth limit */
#define PT_GEOM_STRIDE 4     /* LDS doubles per sphere: cx cy cz r2 */
#define PT_FILT_STRIDE 6     /* HBM f32x2 per primitive PAIR: cx cy cz r2_hi neg_tol kq (phase-1 filter; kq = |c|^2 - r2_hi of the sign-test form) */
#define PT_TRI32_STRIDE 16    /* floats per triangle of the fp32 pre-test table: v0 e1 e2 (9), Ea KU KV KT (4), pad (3) */
#define PT_ENTRY_SRC_STRIDE 6 /* HBM doubles per primitive: cx cy cz R2 |c| Rb (bounding data, fp64) */

/* parked-walk kernels (pt_render_tiles_tri_queued*): bytes of ring per wave in the workspace, slots
 * (workgroups) per XCD the pool provides: 32 CUs x at most 5 resident workgroups, with slack */
#define PT_PARK_WIN_BYTES 9216u /* the refraction form's windowed pixel sums of the wave's tile: 64 pixels x 3 channels x 6 words (the LAST bytes of a wave's region) */
#ifndef PT_PARK_WAVE_BYTES
#define PT_PARK_WAVE_BYTES (65536u + 512u + PT_PARK_WIN_BYTES) /* 512 entries of 128 bytes (pt_body_queued.h, PT_PARK_Q: why 512), then the tile's 64 per-pixel RNG keys, then the windows */
#endif
/* slots per XCD of the two pools (this one and the pending-ray pool below): derived per device from the occupancy of the kernels
 * that take slots, with 25 % slack (pt_pool_slots_per_xcd).  The development build (librt_hip_dev.so) takes RT_HIP_POOL_SLOTS=n
 * instead: with n = 1 acquisition FAILS for all but eight workgroups, which is how the failure path is tested. */
#define PT_PARK_XCDS 8u
/* device-side failures a launch can report through PtLaunch.status (rt_hip.h: RT_HIP_FAIL_*) */
#define PT_FAIL_PEND_SLOT 1u
#define PT_FAIL_PARK_SLOT 2u

#define PT_HULL_PLUS 0x80000000u
#define PT_HULL_MINUS 0x40000000u
#define PT_HULL_MAX_TRIS 65536u /* pt_build_hull_flags is quadratic: larger triangle sets go without the flags */
#define PT_ACC_WS_WORDS (PT_TILE_PIXELS * 3 + 3) /* chunked renders: u64 per tile in the workspace: 192 sums + 3 NaN masks */
#define PT_WIN_N 6 /* words of a windowed pixel-channel sum (pt_scene_ctx.h, win_add): the M_REFRACTION forms of the pooled and parked-walk kernels */
#define PT_ACC_WS_WORDS_WIN (PT_TILE_PIXELS * 3 * PT_WIN_N + 3) /* ... whose chunked renders keep 192 x PT_WIN_N words + 3 NaN masks per tile */

#define PT_REFRACT_MAX_DEPTH 32 /* the pending-ray stacks of the two-child kernels hold max_depth + 2 entries (a pool in global memory
                                 * sized by the launch: 20 KB per entry and resident workgroup) */
#define PT_PEND_FIELDS_HOST 10u /* doubles per pending ray: o, d, T, depth (pt_scene_ctx.h: PT_PEND_FIELDS) */
#define PT_PEND_COLUMNS 512u /* stacks per pool slot: the static body uses one per lane (256), the pooled refraction kernel 128 per wave */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline bool pt_refr_pool_fits(int32_t samples, int32_t max_depth)
{ /* pt_scene_ctx.h, win_add: a window word holds 2^31 pieces.  A path adds one piece per word when it ends, and a sample has at
   * most 2^max_depth paths (every one of its refractive hits starts one more: a full binary tree's leaves) -- the rule keeps
   * samples x 2^(max_depth + 1) <= 2^30, a factor four inside the capacity.  (`samples`: of one sample chunk; until round 5 the
   * rule was 2^(max_depth + 2) on a launch's whole sample count.) */
  return max_depth + 1 <= 30 && ((int64_t)samples << (max_depth + 1)) <= ((int64_t)1 << 30);
}
/* the fewest sample chunks with which a launch of `samples` per pixel fits the rule; 0: none does (max_depth > 29) */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t pt_refr_pool_chunks_needed(int32_t samples, int32_t max_depth)
{
  if (max_depth + 1 > 30 || samples < 1)
    return 0u;
  const int64_t per_chunk = ((int64_t)1 << 30) >> (max_depth + 1); /* >= 1 */
  return (uint32_t)(((int64_t)samples + per_chunk - 1) / per_chunk);
}

#define PT_FLAG_DIFFUSE 2u
#define PT_FLAG_MIRROR 4u
#define PT_FLAG_REFRACT 8u
#define PT_FLAG_CHECKER 16u

struct PtSceneView
{
  const double *entry_src; /* n_spheres + n_triangles bounding records, scan order */
  const double *geom4;     /* n_spheres x PT_GEOM_STRIDE: cx cy cz r2, what the exact test reads when the
                            * scene is too large to stage in LDS */
  float *filt;             /* packed-fp32 filter table for the launch's near_R (pt_build_filter; the shim keeps one
                            * immutable table set per (scene, near_R)) */
  const double *material;
  const double *color_raw;
  const double *tri_geom;
  const double *tri_normal;
  const double *tri_tex;
  const uint32_t *tri_object;
  const double *bvh_src;
  float *bvh_nodes;
  const uint32_t *bvh_tri;
  const double *tri_geom_leaf; /* n_triangles x 9: tri_geom in LEAF order (entry k belongs to triangle bvh_tri[k]), so a leaf's
                                * triangles are contiguous and their loads do not wait for the index load */
  uint32_t n_spheres, n_meshes, n_triangles, any_checker;
  /* the triangles' bounding sphere shows a ray no more than their bounding box does on average (pi R^2 against
   * (ab + bc + ca) / 2): the parked-walk kernel's probe then tests the sphere alone (pt_render_tiles_tri_queued_sph) */
  uint32_t mesh_round;
  uint32_t any_refract, n_bvh_nodes; /* n_bvh_nodes == 0: the scene has no triangles */
  uint32_t wide_range;               /* a centre or radius beyond 1e17: fp32 sums could overflow */
  uint32_t any_mirror_glass;         /* a material with M_REFLECTION and M_REFRACTION: cast_ray traces two children per hit */
  uint32_t bvh_depth;                /* inner nodes on the longest root-to-leaf path: the traversal stack a lane needs */
};
/* Small scenes keep the filter table in LDS and (sphere-only ones) use the sign-test form of
 * the filter, whose NaN-free argument needs every |c|, r <= 1e17; everything else streams the
 * table through scalar loads and keeps the NaN-safe compares. */
#if defined(__HIPCC__)
__host__ __device__
#endif
#ifndef PT_GEOM_LDS_BYTES
#define PT_GEOM_LDS_BYTES (24u * 1024u) /* the LDS staging budget of sphere geometry + materials */
#endif
static inline bool pt_geom_in_lds(const PtSceneView &sc)
{ /* sphere geometry + materials within the staging budget */
  return (PT_GEOM_STRIDE * (uint64_t)sc.n_spheres + PT_MAT_STRIDE * ((uint64_t)sc.n_spheres + sc.n_meshes)) * 8u <= PT_GEOM_LDS_BYTES;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
/* Sphere-only scenes of more than ~85 spheres are better off NOT staged although they would fit: a workgroup's life is
 * short (64 pixels x spp), staging copies 120 bytes per sphere into LDS for each of the frame's tens of thousands of
 * workgroups, and the staged bytes cost resident workgroups (256 spheres: 3 per CU instead of 5).  The pooled body that reads
 * geometry and materials from memory and the filter table through scalar loads (pt_render_tiles_pool_mem_s) measured, per
 * sphere test at 1920x1080 x 64 spp in rooms packed as main.c:65-138 would (profiles/r04_staging_sweep.txt): 10 spheres
 * 1.88 ps against 1.84 staged, 38: 0.660 / 0.634, 64: 0.455 / 0.463, 128: 0.327 / 0.375, 192: 0.284 / 0.327, 256: 0.267 / 0.389.
 * (The pooled sphere kernels only: scenes with triangles, M_REFRACTION or cast_ray keep the staging budget.) */
#ifndef PT_STREAM_ABOVE_BYTES
#define PT_STREAM_ABOVE_BYTES (8u * 1024u)
#endif
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline bool pt_stream_sized(const PtSceneView &sc)
{ /* a sphere scene large enough to be better off streamed */
  return sc.n_triangles == 0u && !sc.wide_range &&
         (PT_GEOM_STRIDE * (uint64_t)sc.n_spheres + PT_MAT_STRIDE * ((uint64_t)sc.n_spheres + sc.n_meshes)) * 8u > PT_STREAM_ABOVE_BYTES;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline bool pt_prefer_streaming(const PtSceneView &sc) { return pt_stream_sized(sc) && !sc.any_refract; } /* (refractive ones: pt_render_tiles_refr_pool_mem, pt_pick_kernel) */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline bool pt_filter_in_lds(const PtSceneView &sc)
{
  return (uint64_t)sc.n_spheres + sc.n_triangles <= PT_FILT_LDS_MAX && !sc.wide_range && pt_geom_in_lds(sc);
}

/* Sign-test form of the phase-1 filter (small sphere scenes; pt_build_filter writes the table, the kernels' BigPrune
 * and the host's big_prune_for reason about it): how far a sphere's squared radius is widened in its table entry
 * kq = |c|^2 - r2_hi', with e = 2^-24, A = |c| + near_R and g = | |c|^2 - r^2 |.  ONE definition for the device code that
 * builds the table and the host code that bounds hit-distance estimates by it (round-3 advisor finding: the two used to
 * be maintained by hand in separate files):
 *   pt_sign_widen_r2    what is added to r^2 before the subtraction from |c|^2 (the fp32 arithmetic's error bound,
 *                       derivation at pt_build_filter);
 *   pt_sign_widen_kq    what is taken off kq afterwards, for the chain's own roundings of kq;
 *   pt_sign_widen_total their sum: r2_hi' - r^2 as the table has it, before kq's final rounding DOWN to fp32 (one more
 *                       2 e |kq|: callers that need an upper bound on the widening multiply by 1.001). */
#define PT_E32 5.9604644775390625e-08 /* 2^-24 */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline double pt_sign_widen_r2(double A, double g) { return (40.0 * PT_E32 * A * A + 8.0 * PT_E32 * g) * (1.0 + 8.0 * PT_E32); }
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline double pt_sign_widen_kq(double g) { return 4.0 * PT_E32 * g; }
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline double pt_sign_widen_total(double A, double g) { return pt_sign_widen_r2(A, g) + pt_sign_widen_kq(g); }

/* The filter buffer of a small scene (pt_filter_in_lds) holds two tables: the pair table of
 * phase 1 (PT_FILT_STRIDE f32x2 per primitive pair, + the look-ahead pair), then, 16-byte aligned,
 * the fp32 triangle table of the per-lane pre-test (PT_TRI32_STRIDE floats per triangle). */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t pt_filt_pair_slots(uint32_t n_entries)
{ /* f32x2 slots of the pair table, rounded up to a 16-byte boundary */
  return (PT_FILT_STRIDE * ((n_entries + 1u) / 2u + 1u) + 1u) & ~1u;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline size_t pt_filt_bytes(uint32_t n_spheres, uint32_t n_triangles)
{
  return (size_t)pt_filt_pair_slots(n_spheres + n_triangles) * 8u + (size_t)n_triangles * PT_TRI32_STRIDE * 4u;
}

struct PtCamera
{
  double pos[3], horizontal[3], vertical[3], llc[3];
};

struct PtLaunch
{
  PtSceneView scene;
  PtCamera cam;
  int32_t width, height, samples, max_depth;
  uint64_t seed;
  double near_R;  /* the phase-1 filter's assumption |o| <= near_R (rays beyond it skip the filter) */
  /* wave-uniform values formed on the host so they arrive in SGPRs (formed on the device
   * they would occupy -- and spill -- vector registers): near_R^2, width-1, height-1 as the
   * reference forms them, (double)options->width - 1.0 (raytracer.c:203-204) */
  double near_R2, w_minus_1, h_minus_1;
  double filt_shift; /* 12 e (max |c| + near_R): how far behind the origin the sign-test filter starts its ray */
  double hull_margin; /* a ray leaves a hull facet for good if (outward normal) . d exceeds this (rt_hip_shim.hip) */
  uint32_t diag_flags; /* PT_DIAG builds only: bit 0 = walk the rays the probe's bounding sphere rejects, to check them */
  float mesh_bound[5]; /* bvh_probe: the triangles' bounding sphere for this near_R: cx cy cz r2_hi neg_tol (pt_intersect.h, MeshBound) */
  /* two constants passed in so that they live in SGPRs (as literals the compiler parks each in a
   * VGPR pair for the whole loop, and spilled them): BACKGROUND's component 10/255
   * (raytracer.h:46) and DBL_MAX, the initial min_t of intersect() (raytracer.c:396) */
  double background, t_start;
  double inv_w_minus_1, inv_h_minus_1; /* RN(1/(W-1)), RN(1/(H-1)) for div_small_int */
  double acc_scale, acc_inv_scale; /* power-of-two fixed-point scale of the pixel sums */
  uint32_t tile_first, tile_stride, tile_count, tiles_x;
  uint32_t sample_chunks; /* workgroups per tile: each renders 1/sample_chunks of the samples */
  uint32_t acc_windows;   /* sample_chunks > 1: acc_ws holds WINDOWED sums (tile_count x 192 x PT_WIN_N words, then the NaN masks): the M_REFRACTION forms */
  uint32_t integrator;    /* 0 trace_path (raytracer.c:482-554), 1 cast_ray (:556-641) */
  /* sign-test kernels: the scene's leading pairs of wall-sized spheres (radius >= 1000) are pruned among themselves before
   * the exact tests (pt_filter.h, BigPrune): how many pairs (0: off), the distance margin, the least distance and per sphere the least q32 of a wall that may prune */
  uint32_t big_pairs;
  float big_delta, big_tmin;
  float big_qmin[8];
  /* parked-walk kernels: workspace of PT_PARK_XCDS x park_slots_per_xcd slots x 4 waves x PT_PARK_WAVE_BYTES and
   * one in-use flag per slot (zero between launches).  nullptr (the allocation failed): pt_launch_render takes the
   * lane-waiting _tri_big kernels instead, which need no workspace -- the parked-walk kernels never run without one */
  char *park_ws;
  uint32_t *park_flags;
  uint32_t park_slots_per_xcd;
  /* kernels with two-child materials (M_REFRACTION under trace_path; M_REFLECTION | M_REFRACTION under cast_ray): the pool of
   * pending-ray stacks (pt_scene_ctx.h, PendStack): PT_PARK_XCDS x pend_slots_per_xcd slots of pend_slot_doubles doubles
   * (= pend_entries x 10 fields x PT_BLOCK lanes) and one in-use flag per slot */
  double *pend_ws;
  uint32_t *pend_flags;
  uint32_t pend_slots_per_xcd, pend_entries;
  uint64_t pend_slot_doubles; /* = pend_entries x 10 fields x PT_PEND_COLUMNS */
  /* the device's status word (rt_hip_launch_status): a workgroup that cannot get a pool slot ORs PT_FAIL_* into it, renders
   * nothing and leaves its tile NaN -- the caller gets an error code, not a plausible image */
  uint32_t *status;
  unsigned long long *acc_ws;        /* sample_chunks > 1: tile_count x 192 fixed-point sums, then tile_count x 3 NaN masks */
  float *tiles_rgb;
  uint8_t *tiles_rgb8;
  unsigned long long *stats;
};

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
/* host-side launchers, defined next to the kernels in pt_kernel.hip */
size_t pt_render_lds_bytes(const PtSceneView &scene);
/* what a launch adds to the scene's content when the kernel is picked (pt_kernel.hip: pt_pick_table) */
struct PtPickFacts
{
  uint32_t integrator;
  int32_t samples, max_depth; /* samples: per sample chunk */
  bool have_park_ws;          /* the parked-walk kernels' ring workspace exists on the device */
  bool wide_pend_ok;          /* the pending-ray pool can be had at 4 x 512 stacks per slot */
};
int pt_pick_kernel(const PtSceneView &scene, const PtPickFacts &facts);          /* -> index into the family */
hipError_t pt_launch_render(const PtLaunch &launch, hipStream_t stream, int which);
const char *pt_kernel_name_of(int which);
int pt_kernel_count(void);
bool pt_kernel_uses_pend_pool(int which);
bool pt_kernel_is_queued(int which);
bool pt_kernel_takes_chunks(int which);
bool pt_kernel_is_windowed(int which); /* pixel sums are windowed (win_add): the chunk workspace has PT_ACC_WS_WORDS_WIN words per tile */
uint32_t pt_kernel_pend_columns_of(int which);
unsigned long long pt_kernel_launches(int which);
uint32_t pt_pool_slots_per_xcd(bool park_pool); /* on the current device */
#ifdef PT_DEV_KERNELS
int pt_dev_variant();
#endif
hipError_t pt_launch_build_hull_flags(const double *tri_geom, const double *tri_normal, uint32_t n_tri, double tau,
                                      uint32_t *tri_object, hipStream_t stream);
hipError_t pt_launch_build_tables(const PtSceneView &scene, double near_R, float *filt, float *bvh_nodes, hipStream_t stream);
hipError_t pt_launch_selftest_xcc(unsigned int *counts, uint32_t n_workgroups, hipStream_t stream);
hipError_t pt_launch_selftest(int op, const double *a, const double *b, double *out, size_t n, hipStream_t stream);
hipError_t pt_launch_selftest_intersect(int kind, const double *rays, const double *prims, const double *entry_src,
                                        float *filt, float *tri32, uint32_t n, double near_R, double filt_shift,
                                        uint8_t *hit, double *tuv, unsigned long long *keep, hipStream_t stream);
hipError_t pt_launch_untile(const float *tiles_rgb, const uint8_t *tiles_rgb8, int width, int height,
                            uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count, float *image_rgb,
                            uint8_t *image_rgb8, hipStream_t stream);
#endif

#endif /* PT_DEVICE_H */
