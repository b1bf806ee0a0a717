/* pt_body_static.h -- render_tiles_static: lane = (pixel, sample slice), fp64 partial sums; the body of pt_render_tiles_v0, of the M_REFRACTION
 * fallback kernels and of every cast_ray kernel.
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_BODY_STATIC_H
#define PT_BODY_STATIC_H

/* ---- static body: lane = (pixel, sample slice), fp64 partial sums ------------------------
 * Lane l of wave w: pixel (l >> 2) of the wave's 16, sample slice (l & 3): samples s = slice,
 * slice + 4, ...; the four slice sums of a pixel are combined by xor-shuffles in a fixed
 * order.  Floating-point sums have no range limit, which is what scenes with M_REFRACTION
 * need (see render_tiles_pooled); VARIANT 0 of it is the plain reference kernel
 * (RT_HIP_KERNEL_VARIANT=0 of the development build, librt_hip_dev.so). */
/* WHITTED: 0 = trace_path, 1 = cast_ray for scenes where no material has both M_REFLECTION and
 * M_REFRACTION (one child per hit at most: no pending-ray stack), 2 = cast_ray with the stack */
template <int VARIANT, bool REFRACT, bool CHECKER, bool TRIS, bool FILT_LDS, int WHITTED, bool GEOM_LDS>
__device__ __forceinline__ void render_tiles_static(const PtLaunch &L)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ float out_f[PT_TILE_PIXELS * 3];
  __shared__ uint8_t out_b[PT_TILE_PIXELS * 3 + 64];
  __shared__ unsigned long long wg_stats[2];

  SceneCtx S_init = stage_scene<GEOM_LDS, FILT_LDS>(L, lds);
  __shared__ double atan_tab[(CHECKER || WHITTED) ? PT_ATAN_TAB : 1];
  if (CHECKER || WHITTED)
  {
    atan_table_to_lds(atan_tab);
    S_init.atan_tab = atan_tab;
  }
  /* the leading wall-sized spheres pruned among themselves before the exact tests (BigPrune: the sign-form kernels of sphere
   * scenes), as in the pooled body -- round 4: the static kernels had gone without */
  __shared__ __attribute__((aligned(16))) float big_tab[12];
  if (WHITTED && FILT_LDS && !TRIS && L.big_pairs != 0u) /* (cast_ray only: in the static M_REFRACTION kernel -- a fallback now -- it costs 8 bytes of scratch at four waves) */
  {
    if (threadIdx.x < 2 + 2 * PT_BIG_PAIRS)
      big_tab[threadIdx.x] = threadIdx.x == 0 ? L.big_delta : (threadIdx.x == 1 ? L.big_tmin : L.big_qmin[threadIdx.x - 2]);
    S_init.big = BigPrune{big_tab, L.big_pairs};
  }
  const SceneCtx S = S_init;
  if (threadIdx.x < 2)
    wg_stats[threadIdx.x] = 0;
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t pix_in_tile = wave * 16u + (lane >> 2);
  const uint32_t slice = lane & (PT_SLICES - 1);
  const uint32_t tile = L.tile_first + blockIdx.x * L.tile_stride;
  const uint32_t px = (tile % L.tiles_x) * PT_TILE + (pix_in_tile & 7u);
  const uint32_t py = (tile / L.tiles_x) * PT_TILE + (pix_in_tile >> 3);
  const bool inside = px < (uint32_t)L.width && py < (uint32_t)L.height;
  const uint32_t pixel = py * (uint32_t)L.width + px;
  const uint32_t spp = (uint32_t)L.samples;
  const CameraRegs cam = load_camera(L);

  V3 acc = {0, 0, 0}; /* sum of finished samples of this lane's slice */
  Path P;
  P.o = {0, 0, 0};
  P.d = {0, 0, 1};
  P.T = {1, 1, 1};
  P.Ls = {0, 0, 0};
  P.rng = 1;
  P.depth = 0;
  uint32_t n_rays = 0, n_casts = 0;
  uint32_t s = inside ? slice : spp;
  bool fresh = true;
  /* kernels with two-child materials: the workgroup's slot of the pending-ray pool (PendStack) */
  constexpr bool STACKED = REFRACT || WHITTED == 2;
  __shared__ uint32_t pend_slot_lds;
  if (STACKED)
  {
    if (threadIdx.x == 0)
      pend_slot_lds = pt_pool_acquire(L.pend_flags, L.pend_slots_per_xcd, L.status, PT_FAIL_PEND_SLOT);
    __syncthreads();
  }
  const uint32_t pend_slot = STACKED ? pend_slot_lds : 0u;
  /* (no slot: a sizing bug of the pool -- the launcher refuses to launch without a pool.  The tile then comes
   * out NaN, bytes 255, rather than wrong (see the epilogue), and the status word says why: pt_pool_acquire) */
  const bool pend_ok = !STACKED || pend_slot != 0xFFFFFFFFu;
  const PendStack stack = {STACKED && pend_ok ? L.pend_ws + (size_t)pend_slot * L.pend_slot_doubles + threadIdx.x : nullptr,
                           STACKED && pend_ok ? (int)L.pend_entries : 0, PT_BLOCK, PT_PEND_FIELDS * PT_BLOCK};
  if (!pend_ok)
    s = spp;
  int stack_n = 0;
  unsigned long long *diag_ptr = L.stats;
  (void)diag_ptr;

  while (s < spp)
  {
    DIAG(0, 1);
    DIAG_LANES(1);
    if (fresh)
    {
      DIAG(6, 1);
      DIAG_LANES(7);
      start_sample(P, cam, rt_rng_pixel_key(L.seed, pixel), px, py, sample_term((uint32_t)s));
      fresh = false;
    }
    n_rays++;
    const bool finished = WHITTED ? whitted_step<TRIS, FILT_LDS, WHITTED == 2>(S, P, n_casts, diag_ptr, stack, stack_n)
                                  : trace_step<VARIANT, REFRACT, CHECKER, TRIS, FILT_LDS>(S, P, n_casts, diag_ptr,
                                                                                          stack, stack_n);
    if (finished)
    {
      acc = v_add(acc, P.Ls);
      s += PT_SLICES;
      fresh = true;
    }
  }

  /* per-pixel mean: fixed-order reduction over the 4 slice lanes */
  acc.x += __shfl_xor(acc.x, 1);
  acc.y += __shfl_xor(acc.y, 1);
  acc.z += __shfl_xor(acc.z, 1);
  acc.x += __shfl_xor(acc.x, 2);
  acc.y += __shfl_xor(acc.y, 2);
  acc.z += __shfl_xor(acc.z, 2);
  V3 mean = v_scale(acc, 1.0 / (double)spp); /* :215 */
  if (!pend_ok)
    mean.x = mean.y = mean.z = __longlong_as_double(0x7FF8000000000000ll);
  if (slice == 0)
  {
    out_f[3 * pix_in_tile + 0] = inside ? (float)mean.x : 0.f;
    out_f[3 * pix_in_tile + 1] = inside ? (float)mean.y : 0.f;
    out_f[3 * pix_in_tile + 2] = inside ? (float)mean.z : 0.f;
    out_b[3 * pix_in_tile + 0] = inside ? tonemap(mean.x) : 0;
    out_b[3 * pix_in_tile + 1] = inside ? tonemap(mean.y) : 0;
    out_b[3 * pix_in_tile + 2] = inside ? tonemap(mean.z) : 0;
  }
  if (n_rays)
  {
    atomicAdd(&wg_stats[0], (unsigned long long)n_rays);
    atomicAdd(&wg_stats[1], (unsigned long long)n_casts);
  }
  __syncthreads();
  store_tile(L, out_f, out_b, wg_stats, tile, blockIdx.x, S.n_sph + S.n_tri, true, true);
  if (STACKED && pend_ok && threadIdx.x == 0)
    atomicExch(&L.pend_flags[pend_slot], 0u); /* every lane is past its last pop (the barrier above) */
}

#endif /* PT_BODY_STATIC_H */
