/* pt_scene_ctx.h -- what a workgroup stages in LDS (SceneCtx, stage_scene), one sample's path state, pending-ray stacks, windowed pixel sums,
 * the camera (get_camera_ray, raytracer.c:375-384) and the per-sample RNG seeding (start_sample).
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_SCENE_CTX_H
#define PT_SCENE_CTX_H

/* ---- scene as staged in LDS ------------------------------------------------------------ */

struct SceneCtx
{
  const double *geom;     /* LDS: n_sph x PT_GEOM_STRIDE: cx cy cz r2 (fp64, exact tests and normals) */
  const double *mat;      /* LDS: (n_sph + n_meshes) x PT_MAT_STRIDE */
  const double *color_raw; /* HBM: (n_sph + n_meshes) x 3, the colours as given (cast_ray shades with them) */
  const double *tri;      /* HBM: n_tri x 9 (v0, e1, e2), gathered per lane in phase 2 */
  const double *tri_normal;
  const double *tri_tex;
  const uint32_t *tri_object;
  const f32x2 *filt;      /* HBM: ceil(n_entries/2) x PT_FILT_STRIDE packed-fp32 filter pairs */
  const f32x2 *filt_lds;  /* LDS copy of it when the scene is small (PT_FILT_LDS_MAX), else nullptr */
  const float4 *tri32;    /* LDS: the fp32 triangle table of the pre-test (small scenes with triangles), else nullptr */
  const float *bvh_nodes; /* HBM: triangle hierarchy of large meshes (n_bvh_nodes may be 0) */
  const uint32_t *bvh_tri;
  const double *tri_leaf; /* HBM: tri geometry in leaf order (pt_device.h) */
  uint32_t n_bvh_nodes;
  MeshBound mesh_bound;   /* bvh_probe's bounding sphere of all triangles (launch arguments: SGPRs) */
  double hull_margin;     /* a ray leaves a hull facet for good if outward . d exceeds this (launch argument) */
  double near_R2;         /* the filter is valid for ray origins with |o|^2 <= near_R2 */
  double filt_shift;      /* tol_max of the sign-test filter form (scan_filtered) */
  double bg, t_start;     /* BACKGROUND's component and DBL_MAX, from the launch arguments (SGPR pairs) */
  uint32_t n_sph, n_tri;
  int max_depth;
  bool stale_uv;          /* M_CHECKERED materials AND triangles: hit.u / hit.v follow the TriLast rule */
  const double *atan_tab; /* LDS: atan2_tab's coefficients (kernels with M_CHECKERED code), else nullptr */
  BigPrune big;           /* pruning of the leading wall-sized spheres among themselves (sign-test kernels), or off */
};

/* GEOM_LDS: sphere geometry and materials are staged in LDS (the pointers are LDS pointers at
 * compile time); otherwise the scene is beyond the staging budget (pt_geom_in_lds) and the
 * kernel reads them from memory.  Kernels pick the instantiation once, at entry. */
template <bool GEOM_LDS, bool FILT_LDS, bool SPH_FILT = false>
__device__ __forceinline__ SceneCtx stage_scene(const PtLaunch &L, double *lds)
{
  /* FILT_LDS without GEOM_LDS (pt_render_tiles_pool_mem_s: sphere scenes beyond the staging budget): the small scenes' FORM of
   * the filter -- sign tests, descending pairs, per-tile culling, wall pruning -- with the pair table read from memory
   * (wave-uniform addresses: scalar loads) instead of from an LDS copy */
  constexpr bool FILT_FROM_MEMORY = FILT_LDS && !GEOM_LDS;
  static_assert(!SPH_FILT || !FILT_LDS, "SPH_FILT: the sphere pairs only, for the hierarchy kernels");
  const PtSceneView &sc = L.scene;
  const uint32_t n_sph = sc.n_spheres, n_mat = sc.n_spheres + sc.n_meshes;
  constexpr bool staged = GEOM_LDS;
  double *geom = lds;
  double *mat = geom + PT_GEOM_STRIDE * (size_t)n_sph;
  if (staged)
  {
    for (uint32_t i = threadIdx.x; i < n_sph; i += PT_BLOCK)
    {
      const double *src = sc.entry_src + PT_ENTRY_SRC_STRIDE * (size_t)i; /* cx cy cz r2 |c| R */
      double *g = geom + PT_GEOM_STRIDE * i;
      g[0] = src[0];
      g[1] = src[1];
      g[2] = src[2];
      g[3] = src[3];
    }
    for (uint32_t k = threadIdx.x; k < PT_MAT_STRIDE * n_mat; k += PT_BLOCK)
      mat[k] = sc.material[k];
  }
  /* Small scenes keep the filter table in LDS (measured 4 % faster than scalar loads on the
   * 38-sphere room: ds_read is prefetched across pairs, s_load is not); large ones stream
   * it through the constant cache. */
  const uint32_t n_entries = n_sph + sc.n_triangles;
  f32x2 *filt_lds = nullptr;
  if (SPH_FILT && !GEOM_LDS) /* (pt_render_tiles_tri_queued_mem*: the sphere pairs from memory too, by scalar loads) */
    filt_lds = reinterpret_cast<f32x2 *>(sc.filt);
  if (SPH_FILT && GEOM_LDS)
  { /* the pairs that cover the spheres (the last may carry the first triangle's bound: masked in the scan) */
    filt_lds = reinterpret_cast<f32x2 *>(mat + PT_MAT_STRIDE * (size_t)n_mat);
    const uint32_t n_slots = pt_filt_pair_slots(n_sph);
    const f32x2 *src = reinterpret_cast<const f32x2 *>(sc.filt);
    for (uint32_t k = threadIdx.x; k < n_slots; k += PT_BLOCK)
      filt_lds[k] = src[k];
  }
  if (FILT_FROM_MEMORY)
    filt_lds = reinterpret_cast<f32x2 *>(sc.filt);
  if (FILT_LDS && !FILT_FROM_MEMORY)
  {
    filt_lds = reinterpret_cast<f32x2 *>(mat + PT_MAT_STRIDE * (size_t)n_mat);
    /* the pair table (+ the look-ahead pair) and, behind it, the fp32 triangle table */
    const uint32_t n_slots = pt_filt_pair_slots(n_entries) + sc.n_triangles * (PT_TRI32_STRIDE / 2);
    const f32x2 *src = reinterpret_cast<const f32x2 *>(sc.filt);
    for (uint32_t k = threadIdx.x; k < n_slots; k += PT_BLOCK)
      filt_lds[k] = src[k];
  }
  SceneCtx ctx;
  if (GEOM_LDS)
  {
    ctx.geom = geom;
    ctx.mat = mat;
  }
  else
  {
    ctx.geom = sc.geom4;
    ctx.mat = sc.material;
  }
  ctx.color_raw = sc.color_raw;
  ctx.tri = sc.tri_geom;
  ctx.tri_normal = sc.tri_normal;
  ctx.tri_tex = sc.tri_tex;
  ctx.tri_object = sc.tri_object;
  ctx.filt = reinterpret_cast<const f32x2 *>(sc.filt);
  ctx.filt_lds = filt_lds;
  /* hierarchy scenes: the same table in LEAF order, in HBM behind the pair table (leaf_pretest).  A small scene's
   * table is in scan order, for its own kernels (pt_launch_build_tables): the general kernels, which walk the
   * hierarchy of such a scene too, go without the pre-test there */
  ctx.tri32 = FILT_LDS ? reinterpret_cast<const float4 *>(filt_lds + pt_filt_pair_slots(n_entries))
                       : ((sc.n_bvh_nodes != 0u && !pt_filter_in_lds(sc))
                              ? reinterpret_cast<const float4 *>(reinterpret_cast<const f32x2 *>(sc.filt) +
                                                                                 pt_filt_pair_slots(n_entries))
                                               : nullptr);
  ctx.bvh_nodes = sc.bvh_nodes;
  ctx.bvh_tri = sc.bvh_tri;
  ctx.tri_leaf = sc.tri_geom_leaf;
  ctx.n_bvh_nodes = sc.n_bvh_nodes;
  ctx.mesh_bound = {L.mesh_bound[0], L.mesh_bound[1], L.mesh_bound[2], L.mesh_bound[3], L.mesh_bound[4]};
  ctx.hull_margin = L.hull_margin;
  ctx.near_R2 = L.near_R2;
  ctx.filt_shift = L.filt_shift;
  ctx.bg = L.background;
  ctx.t_start = L.t_start;
  ctx.n_sph = n_sph;
  ctx.n_tri = sc.n_triangles;
  ctx.max_depth = L.max_depth;
  ctx.stale_uv = sc.any_checker != 0 && sc.n_triangles != 0;
  ctx.big = BigPrune{nullptr, 0u};
  ctx.atan_tab = nullptr;
  return ctx;
}

/* ---- one sample's path state ------------------------------------------------------------ */

struct Path
{
  V3 o, d;      /* current ray */
  V3 T;         /* throughput */
  V3 Ls;        /* radiance gathered so far */
  uint64_t rng;
  int depth;
};

/* Deferred second child of an M_REFRACTION hit (raytracer.c:523-529 traces two children per
 * hit, the "refracted" one completely first): depth-first order = a LIFO of pending rays.
 * At most one entry is pushed per depth level, so max_depth + 2 slots suffice.
 * Round 4: the LIFO is no longer a private array (34 x 80 B = 2.7 KB of scratch memory per lane, indexed dynamically:
 * the one thing that kept the static-body kernels from ever being free of scratch) but lives in a workspace slot in
 * global memory that the workgroup takes from a per-device pool at entry (pt_pool_acquire, as the parked-walk kernels
 * take their rings) -- entry-major, field-major, lane-minor: [entry][o xyz, d xyz, T xyz, depth][PT_BLOCK lanes], so a
 * wave's push or pop of one field is one coalesced 512-byte access.  A lane only ever reads what it wrote itself.  The
 * slot is sized by the launch's max_depth (PtLaunch.pend_entries = max_depth + 2). */
#define PT_PEND_FIELDS 10u
static_assert(PT_PEND_FIELDS == PT_PEND_FIELDS_HOST, "pending-ray record");
struct PendStack
{
  double *base;    /* this lane's (static body) or this path's (pooled body) first double (nullptr in kernels without a stack) */
  int capacity;    /* entries */
  /* doubles from one field / one entry to the next.  Static body: [entry][field][PT_BLOCK lanes] (a wave's push of a field is
   * one coalesced access).  Pooled body (pt_render_tiles_refr_pool): a path's stack moves with the path between lanes, so it
   * is addressed by the path's id, [id][entry][field]: 80 contiguous bytes per pending ray */
  uint32_t field_stride, entry_stride;
  __device__ __forceinline__ void push(int e, const V3 &o, const V3 &d, const V3 &T, int depth) const
  {
    double *q = base + (size_t)e * entry_stride;
    const uint32_t f = field_stride;
    q[0 * f] = o.x; q[1 * f] = o.y; q[2 * f] = o.z;
    q[3 * f] = d.x; q[4 * f] = d.y; q[5 * f] = d.z;
    q[6 * f] = T.x; q[7 * f] = T.y; q[8 * f] = T.z;
    q[9 * f] = __longlong_as_double((long long)depth);
  }
  __device__ __forceinline__ void pop(int e, V3 &o, V3 &d, V3 &T, int &depth) const
  {
    const double *q = base + (size_t)e * entry_stride;
    const uint32_t f = field_stride;
    o = {q[0 * f], q[1 * f], q[2 * f]};
    d = {q[3 * f], q[4 * f], q[5 * f]};
    T = {q[6 * f], q[7 * f], q[8 * f]};
    depth = (int)__double_as_longlong(q[9 * f]);
  }
};

/* ---- order-free pixel sums WITHOUT a bound on the terms (pt_render_tiles_refr_pool) -----------------------------------
 * The pooled kernels add radiance terms to 64-bit fixed-point sums, which needs a bound on a term (throughput <= 1).  Scenes
 * with M_REFRACTION have none: the reference's fresnel weight reaches 7.3 per hit from inside a sphere, 1 - fresnel -6.3
 * (raytracer.c:517-529).  Here a pixel channel is PT_WIN_N signed 64-bit words, word k collecting the bits
 * [PT_WIN_E0 + 32 k, PT_WIN_E0 + 32 k + 32) of every term: a double's 53-bit mantissa is cut -- exactly, by shifts -- into the
 * (at most three) 32-bit pieces that fall into consecutive words, and each piece is added with an integer LDS atomic.
 * Integer addition commutes and associates, so the sums do not depend on the order or grouping of terms (any lane / wave /
 * tile / GPU assignment gives the same words), there is no rounding at all above 2^PT_WIN_E0, and a word overflows only
 * after 2^31 pieces (pt_refr_pool_fits keeps a sample chunk's samples x 2^(max_depth + 1) below 2^30; chunks are merged in
 * carry-normalised form, win_normalize, whose words are below 2^32 each).  Range: 2^-64 (bits below are dropped: 5e-20
 * absolute per term) to 2^128, all a float32 pixel can hold; a term at or above that flags the pixel like a NaN.  (Six words: a
 * seventh would cost the kernel its fourth workgroup per CU.) */
#define PT_WIN_E0 (-64) /* (PT_WIN_N: pt_device.h) */
__device__ __forceinline__ bool win_add(unsigned long long *w, double x)
{ /* -> false: x is not finite or too large (the caller flags the pixel) */
  const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
  const int ex = (int)((bits >> 52) & 0x7FFu);
  if (ex == 0x7FF)
    return false;
  const unsigned long long mant = (bits & 0xFFFFFFFFFFFFFull) | (ex ? 0x10000000000000ull : 0ull);
  /* x = +-mant * 2^(e2), e2 = max(ex, 1) - 1075; its bit 0 sits `sh` bits above the accumulator's origin */
  const int sh = (ex ? ex : 1) - 1075 - PT_WIN_E0;
  if (mant == 0ull || sh <= -53)
    return true; /* zero, or entirely below 2^PT_WIN_E0 */
  if (sh + 53 > 32 * PT_WIN_N)
    return false;
  /* v = mant shifted so that v's bit 0 is bit 0 of word k0 (k0 = floor(sh / 32); sh < 0: the low bits are dropped) */
  const int k0 = sh >= 0 ? (sh >> 5) : 0;
  const int r = sh >= 0 ? (sh & 31) : 0;
  const unsigned long long m = sh >= 0 ? mant : (mant >> (-sh));
  const unsigned long long lo = m << r;                                /* bits 0..63 of v (m < 2^53, r < 32: bits up to 84) */
  const unsigned long long hi = r ? (m >> (64 - r)) : 0ull;            /* bits 64.. of v */
  const long long sgn = (long long)bits < 0 ? -1ll : 1ll;
  const unsigned long long p0 = lo & 0xFFFFFFFFull, p1 = lo >> 32, p2 = hi; /* p2 < 2^21 */
  if (p0) atomicAdd(&w[k0], (unsigned long long)(sgn * (long long)p0));
  if (p1) atomicAdd(&w[k0 + 1], (unsigned long long)(sgn * (long long)p1));
  if (p2) atomicAdd(&w[k0 + 2], (unsigned long long)(sgn * (long long)p2));
  return true;
}
/* Carry-normalised form: words 0 .. PT_WIN_N - 2 in [0, 2^32), the top word takes the rest.  Exact (word k has weight 2^(32 k):
 * what leaves a word upward is a multiple of 2^32) and CANONICAL: one integer total has one such form, whatever pieces it was
 * added up from -- so win_value of the normalised words is the same double for every grouping of the samples (sample chunks
 * merged in the workspace, tiles split over devices), and a merged word collects one piece below 2^32 per chunk. */
__device__ __forceinline__ void win_normalize(unsigned long long *w)
{
#pragma unroll
  for (int k = 0; k + 1 < PT_WIN_N; k++)
  {
    const long long c = (long long)w[k] >> 32; /* floor(w / 2^32) */
    w[k] -= (unsigned long long)c << 32;
    w[k + 1] += (unsigned long long)c;
  }
}
/* the sum: words (normalised by the caller: win_normalize) combined from the top (each conversion and product is exact up to 2^-53
 * relative of its own word: the result is within a few ulps of the exact sum, which is more than the reference's own
 * left-to-right fp64 summation guarantees) */
__device__ __forceinline__ double win_value(const unsigned long long *w)
{
  double v = 0.0;
#pragma unroll
  for (int k = PT_WIN_N - 1; k >= 0; k--)
    v += ldexp((double)(long long)w[k], PT_WIN_E0 + 32 * k);
  return v;
}

/* ids of the pending-ray stacks of the pooled refraction kernels: 128 per wave in render_tiles_pooled (a wave never holds more than
 * 64 paths in its lanes and 64 on its waiting list), 511 in render_tiles_queued (lanes + list + up to 382 parked rays in its
 * ring), handed out lazily -- at a path's first M_REFRACTION hit -- from a free mask of WORDS 64-bit words in LDS, by
 * compare-and-swap: lanes of one wave contend in lock step, one wins per round, and few ask in the same trip */
template <uint32_t WORDS = 2u>
__device__ __forceinline__ uint32_t pend_id_take(unsigned long long *free_mask)
{
  for (;;)
  {
    uint32_t w = 0u;
    unsigned long long m = free_mask[0];
#pragma unroll
    for (uint32_t k = 1u; k < WORDS; k++)
      if (m == 0ull)
      {
        m = free_mask[k];
        w = k;
      }
    if (m == 0ull)
      return 64u * WORDS; /* none free (cannot happen: as many ids as a wave can hold paths) */
    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
    if (atomicCAS(&free_mask[w], m, m & ~(1ull << bit)) == m)
      return bit + 64u * w;
  }
}
__device__ __forceinline__ void pend_id_give(unsigned long long *free_mask, uint32_t id)
{
  atomicOr(&free_mask[id >> 6], 1ull << (id & 63u));
}
/* the pooled refraction kernels' view of a path's stack: the id is taken at the FIRST push (most paths never meet an
 * M_REFRACTION surface and never ask), records are [id][entry][field], 80 contiguous bytes.  IDS = ids per wave (a power of
 * two); NONE = "no id yet" (0xFF in render_tiles_pooled, whose meta word has eight bits for it; IDS - 1 in render_tiles_queued,
 * whose last id is never handed out: its free mask starts without it, 511 ids for at most 510 paths) */
template <uint32_t IDS, uint32_t NONE>
struct PoolStackT
{
  double *wave_base;             /* the wave's stacks in the workgroup's pool slot */
  unsigned long long *free_mask; /* LDS: the wave's free ids */
  uint32_t *id;                  /* the path's id (a register of the calling lane), NONE: none yet */
  int capacity;                  /* entries per stack */
  /* (min: "none free" -- which more ids than paths rule out -- must not address another wave's stacks) */
  __device__ __forceinline__ double *rec(int e) const { return wave_base + ((size_t)min(*id, IDS - 1u) * (uint32_t)capacity + (uint32_t)e) * PT_PEND_FIELDS; }
  __device__ __forceinline__ void push(int e, const V3 &o, const V3 &d, const V3 &T, int depth) const
  {
    if (*id == NONE)
    {
      const uint32_t got = pend_id_take<IDS / 64u>(free_mask);
      *id = got >= IDS ? NONE : got;
    }
    double *q = rec(e);
    q[0] = o.x; q[1] = o.y; q[2] = o.z;
    q[3] = d.x; q[4] = d.y; q[5] = d.z;
    q[6] = T.x; q[7] = T.y; q[8] = T.z;
    q[9] = __longlong_as_double((long long)depth);
  }
  /* The path may be popped by ANOTHER lane of this wave, after a trip through the waiting list.  A wave's vector-memory
   * operations complete in issue order, so the record is in L2 before any later load of this wave is served; what a later
   * load must not do is hit a stale line in the CU's L1 (left by an earlier pop of the same slot): the pops bypass it
   * (agent-scope relaxed loads = `sc1`, like the parked-walk kernels' ring).  No wait at the push: a fence there
   * (s_waitcnt vmcnt(0) in a trip in which any lane hits glass, i.e. most trips) cost 2 % of the frame. */
  __device__ __forceinline__ void pop(int e, V3 &o, V3 &d, V3 &T, int &depth) const
  {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(rec(e));
    double v[PT_PEND_FIELDS];
#pragma unroll
    for (uint32_t f = 0; f < PT_PEND_FIELDS; f++)
      v[f] = __longlong_as_double((long long)__hip_atomic_load(q + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    o = {v[0], v[1], v[2]};
    d = {v[3], v[4], v[5]};
    T = {v[6], v[7], v[8]};
    depth = (int)__double_as_longlong(v[9]);
  }
};
typedef PoolStackT<128u, 0xFFu> PoolStack; /* render_tiles_pooled */

/* a slot of a pool of `per` slots per XCD with one in-use flag each (zero between launches).  None to be had after a bounded
 * search (the pools are sized so that this cannot happen, pt_pool_slots_per_xcd): 0xFFFFFFFF, and `fail_bit` is ORed into the
 * device's status word -- the caller's workgroup renders nothing, and the host reports the launch as failed (rt_hip_launch_status) */
__device__ __forceinline__ uint32_t pt_pool_acquire(uint32_t *flags_base, uint32_t per, uint32_t *status, uint32_t fail_bit)
{
  if (flags_base == nullptr || per == 0u)
  {
    if (status != nullptr)
      atomicOr(status, fail_bit);
    return 0xFFFFFFFFu;
  }
  /* s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, width 4): the XCD this wave runs on */
  const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
  uint32_t *flags = flags_base + xcc * per;
  uint32_t i = ((blockIdx.x * 2654435761u) >> 7) % per;
  for (uint32_t probes = 0; probes < 64u * per; probes++)
  {
    if (atomicCAS(&flags[i], 0u, 1u) == 0u)
      return xcc * per + i;
    i = (i + 1u == per) ? 0u : i + 1u;
    if ((probes & 15u) == 15u)
      __builtin_amdgcn_s_sleep(8);
  }
  if (status != nullptr)
    atomicOr(status, fail_bit);
  return 0xFFFFFFFFu;
}

struct CameraRegs
{
  V3 pos, horizontal, vertical, llc;
  double w_minus_1, h_minus_1, inv_w_minus_1, inv_h_minus_1;
};

__device__ __forceinline__ CameraRegs load_camera(const PtLaunch &L)
{
  CameraRegs c;
  c.pos = ld3(L.cam.pos);
  c.horizontal = ld3(L.cam.horizontal);
  c.vertical = ld3(L.cam.vertical);
  c.llc = ld3(L.cam.llc);
  c.w_minus_1 = L.w_minus_1;
  c.h_minus_1 = L.h_minus_1;
  c.inv_w_minus_1 = L.inv_w_minus_1;
  c.inv_h_minus_1 = L.inv_h_minus_1;
  return c;
}

/* The pooled kernels keep the camera in LDS instead: as kernel arguments its 16 doubles sit in 32 of the wave's
 * ~100 SGPRs for the whole trip loop although only the camera-sample batch (once per 64 jobs) reads them, and the
 * kernels are at the SGPR limit -- what does not fit is spilled to VGPR lanes and read back with v_readlane, a
 * VALU slot each, in loops that are bound by VALU issue.  camera_to_lds: once per workgroup, before a barrier. */
#define PT_CAM_LDS_DOUBLES 16
__device__ __forceinline__ void camera_to_lds(const PtLaunch &L, double *cam_lds)
{
  if (threadIdx.x < 12)
    cam_lds[threadIdx.x] = (&L.cam.pos[0])[threadIdx.x]; /* pos, horizontal, vertical, llc: contiguous (PtCamera) */
  else if (threadIdx.x < PT_CAM_LDS_DOUBLES)
    cam_lds[threadIdx.x] = threadIdx.x == 12 ? L.w_minus_1 : (threadIdx.x == 13 ? L.h_minus_1 : (threadIdx.x == 14 ? L.inv_w_minus_1 : L.inv_h_minus_1));
}
/* an index the compiler cannot see through: loads addressed with it stay where they are written (hoisted out of
 * the trip loop they would occupy vector registers for its whole length instead) */
__device__ __forceinline__ uint32_t opaque_zero()
{
  uint32_t z = 0;
  asm volatile("" : "+v"(z));
  return z;
}
__device__ __forceinline__ CameraRegs load_camera_lds(const double *cam_lds)
{
  const uint32_t z = opaque_zero();
  CameraRegs c;
  c.pos = {cam_lds[z + 0], cam_lds[z + 1], cam_lds[z + 2]};
  c.horizontal = {cam_lds[z + 3], cam_lds[z + 4], cam_lds[z + 5]};
  c.vertical = {cam_lds[z + 6], cam_lds[z + 7], cam_lds[z + 8]};
  c.llc = {cam_lds[z + 9], cam_lds[z + 10], cam_lds[z + 11]};
  c.w_minus_1 = cam_lds[z + 12];
  c.h_minus_1 = cam_lds[z + 13];
  c.inv_w_minus_1 = cam_lds[z + 14];
  c.inv_h_minus_1 = cam_lds[z + 15];
  return c;
}
__device__ __forceinline__ V3 load_camera_pos_lds(const double *cam_lds)
{
  const uint32_t z = opaque_zero();
  return {cam_lds[z + 0], cam_lds[z + 1], cam_lds[z + 2]};
}

/* a / b, correctly rounded, for a >= 0 and an INTEGER 1 <= b < 2^20, given y = RN(1/b)
 * (formed on the host): q0 = RN(a y); r = a - b q0 (exact: a multiple of ulp(q0) below
 * 2.01 b ulp(q0), so it fits 53 bits); q = RN(q0 + r y).
 *   |q0 - a/b| <= 2.01 2^-53 a/b, and q0 + r y = a/b + (r/b) eta with |eta| <= 2^-53, i.e. the
 *   final rounding sees a/b perturbed by <= 2.01 2^-106 a/b.  A quotient by an odd integer
 *   b is never a rounding midpoint, and its distance from one is >= ulp(q) / (2 b) >= 2^-74
 *   relative -- 2^32 times the perturbation -- so RN(q0 + r y) = RN(a/b).  (Even b = 2^k b':
 *   scale by 2^-k first, exact.)  3 instructions instead of the ~14 of an fp64 division
 *   (tests/test_host.py checks the identity with exact rational arithmetic). */
__device__ __forceinline__ double div_small_int(double a, double b, double y)
{
  const double q0 = a * y;
  const double r = __builtin_fma(-q0, b, a);
  return __builtin_fma(r, y, q0);
}

/* raytracer.c:203-206 + get_camera_ray :375-384, stream re-seeded per (pixel, sample) */
/* the sample half of the stream key (rt_rng.h, rt_rng_sample_state): pixel_key + 0xD1B5... * (sample + 1).  Callers whose
 * sample index is wave-uniform (a batch of a full tile = one sample index of every pixel) form it once, as scalar work:
 * as vector work it is two quarter-rate 64-bit multiplies per lane */
__device__ __forceinline__ uint64_t sample_term(uint32_t s)
{ /* (s + 1 in 32 bits -- sample indices are below 2^31 -- so that the product has no 64-bit addend: written as
   * C * ((uint64_t)s + 1) the compiler keeps C itself in a register pair for the "+ C" of s * C + C) */
  const uint32_t s1 = s + 1u;
  return 0xD1B54A32D192ED03ull * (uint64_t)s1;
}
/* the same for a wave-uniform sample index given in the lanes' registers: formed by the scalar unit, and pinned there (or
 * the compiler merges it with the per-lane form of the ragged-tile branch and multiplies in the vector unit after all) */
__device__ __forceinline__ uint64_t sample_term_uniform(uint32_t s_any_lane)
{
  const uint64_t t = sample_term((uint32_t)__builtin_amdgcn_readfirstlane((int)s_any_lane));
  uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t sample_state_from_term(uint64_t pixel_key, uint64_t term)
{
  const uint64_t h = rt_mix64(pixel_key + term);
  return h ? h : 0x9E3779B97F4A7C15ull; /* = rt_rng_sample_state(pixel_key, s) for term = sample_term(s) */
}
__device__ __forceinline__ void start_sample(Path &P, const CameraRegs &cam, uint64_t pixel_key, uint32_t px,
                                             uint32_t py, uint64_t term)
{
  P.rng = sample_state_from_term(pixel_key, term);
  /* (x + rnd) / (W - 1): exactly the reference's quotient, see div_small_int */
  const double u = div_small_int((double)px + rnd(P.rng), cam.w_minus_1, cam.inv_w_minus_1);
  const double v = div_small_int((double)py + rnd(P.rng), cam.h_minus_1, cam.inv_h_minus_1);
  const V3 on_plane = v_add(cam.llc, v_add(v_scale(cam.horizontal, u), v_scale(cam.vertical, v)));
  P.o = cam.pos;
  /* vec3_normalize (vector.h:53-58): w * (1.0 / sqrt(w.w)).  Where |w|^2 is in [1e-200, 1e200] -- every sane camera -- the
   * square root and the reciprocal are hipcc's own expansions without their range scaling (sqrt_unscaled, rcp_unscaled:
   * the same instructions on the same values, so the same doubles; the selftest compares them with the IEEE results):
   * ~25 instructions fewer per camera sample, which config 2 and 3 -- most of whose ray-bounces are first bounces --
   * notice.  Anything else takes the library forms. */
  const V3 w = v_sub(cam.pos, on_plane);
  const double ww = v_dot(w, w);
  P.d = (ww >= 1e-200 && ww <= 1e200) ? v_scale(w, rcp_unscaled(sqrt_unscaled(ww))) : v_scale(w, 1.0 / sqrt(ww));
  P.T = {1, 1, 1};
  P.Ls = {0, 0, 0};
  P.depth = 0;
}

#endif /* PT_SCENE_CTX_H */
