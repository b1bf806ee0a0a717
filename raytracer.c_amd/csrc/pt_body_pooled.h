/* pt_body_pooled.h -- render_tiles_pooled: the shipped body of the headline kernel -- a pool of (pixel, sample) jobs per tile, paths swapped in and
 * out of an LDS waiting list, primary trips, fixed-point (or windowed) pixel sums in LDS.
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_BODY_POOLED_H
#define PT_BODY_POOLED_H

/* ---- shipped kernel: pooled samples, fixed-point pixel sums ------------------------------ */

/* PT_MIN_WAVES: waves per SIMD the register allocator must leave room for.  The loop is
 * VALU-issue bound and wants latency cover: on config 4 (1080p x 128 spp) 3 waves/SIMD took
 * 56.2 ms, 4: 49.8, 5: 48.0, 6: 47.2, 7: 48.1, 8: 54.1 when measured on revision c.  After the
 * uniform values moved to SGPRs and the checker code out of this kernel, 5 waves fit in 94
 * VGPRs with NO scratch (47.1 ms) and 6 waves need 56 B/lane of spills (46.7 ms, but 290 MB
 * of spill write-back per frame against 31 MB of algorithmic output): 5 it is. */
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 5
#endif
#ifndef PT_MIN_WAVES_TRI
#define PT_MIN_WAVES_TRI 5
#endif
#ifndef PT_MIN_WAVES_CHK
#define PT_MIN_WAVES_CHK 5 /* the M_CHECKERED sphere kernels: 96 VGPRs without scratch since atan2_tab (round 4; 128 before, no bound) */
#endif
/* postponed hierarchy walks of the pooled kernels: lanes that make a batch; trips the oldest waits */
#ifndef PT_MESH_BATCH
#define PT_MESH_BATCH 32
#endif
#ifndef PT_MESH_MAX_WAIT
#define PT_MESH_MAX_WAIT 16
#endif
/* rejection rounds per trip for the directions of diffuse hits (pooled kernels) */
#ifndef PT_DIR_ROUNDS
#define PT_DIR_ROUNDS 4
#endif
/* A/B knob (round 4), default off: stop the rounds early when fewer than PT_DIR_MIN_LANES lanes still want a sample (they
 * carry over like the stragglers of the fourth round).  A round is a wave's 44 instructions whoever still needs it and a
 * carried lane costs one lane's share of a trip, so a lane model put the break-even at ~6 lanes and promised -1 %; measured
 * with 4 / 6 / 8: headline +1.6 / +1.7 / +2.3 %, config 3 +0.6 / +1.3 / +2.8 %, config 5 +2.6 / +2.7 / +3.0 %, config 2 within
 * noise (profiles/r04_dir_min_lanes_ab.txt): a carried lane also holds up its pool's tail.  1 = every round. */
#ifndef PT_DIR_MIN_LANES
#define PT_DIR_MIN_LANES 1
#endif
/* Pooled kernel body.  Its fixed-point pixel sums rest on a throughput bounded by 1; scenes with M_REFRACTION have none
 * (fresnel = 0.1 + 0.9 (1 - facing)^3 reaches 7.3 when a surface is hit from inside, kt goes negative), so no fixed-point scale
 * can be fixed in advance: they take the REFR form of this body (below: windowed sums) or, where that does not apply, the
 * static body. */
/* REFR (pt_render_tiles_refr_pool, round 4): scenes with M_REFRACTION on the pooled body.  Two things kept them on the static
 * body: pixel sums need a bound on a term (here: win_add, order-free without one), and the second child of a refractive hit
 * waits on a per-lane stack while paths of this body move between lanes (here: the stack is addressed by a path ID that
 * travels with the path -- PendStack, pend_id_take). */
template <bool CHECKER, bool TRIS, bool FILT_LDS, bool GEOM_LDS, bool REFR = false>
__device__ __forceinline__ void render_tiles_pooled(const PtLaunch &L)
{
  static_assert(!REFR || (CHECKER && FILT_LDS && (GEOM_LDS || !TRIS)), "the pooled refraction kernels: sphere scenes (staged, or streamed from memory) and small staged mesh scenes, every material");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ float out_f[PT_TILE_PIXELS * 3];
  __shared__ uint8_t out_b[PT_TILE_PIXELS * 3 + 64];
  __shared__ unsigned long long wg_stats[2];
  __shared__ double cam_lds[PT_CAM_LDS_DOUBLES];             /* the camera (camera_to_lds) */
  __shared__ unsigned long long pix_sum[REFR ? 1 : PT_TILE_PIXELS * 3]; /* fixed-point radiance sums */
  __shared__ unsigned long long pix_win[REFR ? PT_TILE_PIXELS * 3 * PT_WIN_N : 1]; /* REFR: windowed sums without a bound on the terms (win_add) */
  __shared__ unsigned long long pend_free[REFR ? PT_BLOCK / 64 : 1][2];            /* REFR: per wave, the free ids of its 128 pending-ray stacks */
  __shared__ uint32_t pend_slot_lds;
  __shared__ unsigned long long pix_nan[3];                  /* per channel: pixels that received a NaN sample */
  __shared__ unsigned long long pix_key[PT_TILE_PIXELS];     /* per-pixel half of the RNG key */
  /* kernels with a triangle hierarchy postpone its walks in the lanes (see the loop): they keep round 2's job
   * hand-out, a queue of prepared camera samples; all others swap whole paths in and out (SWAP, see the loop) */
  constexpr bool DEFER_MESH = TRIS && !FILT_LDS;
  constexpr bool SWAP = !DEFER_MESH;
  /* !SWAP: per-wave queue of prepared camera samples: direction, RNG state, pixel slot (64 entries) */
  __shared__ double q_dir[SWAP ? 1 : PT_BLOCK / 64][SWAP ? 1 : 3 * 64];
  __shared__ unsigned long long q_rng[SWAP ? 1 : PT_BLOCK / 64][SWAP ? 1 : 64];
  __shared__ uint32_t q_pix[SWAP ? 1 : PT_BLOCK / 64][SWAP ? 1 : 64];
  /* SWAP: per-wave list of WAITING paths (up to 64): origin, direction (the normal while a direction is still to be
   * sampled), throughput, RNG state, [checker factor]; depth / pixel slot / need_dir; material slot of a pending direction */
  constexpr uint32_t WAIT_F = CHECKER ? 11u : 10u;
  __shared__ double w_f[SWAP ? PT_BLOCK / 64 : 1][SWAP ? WAIT_F : 1][SWAP ? 64 : 1];
  __shared__ uint32_t w_u[SWAP ? PT_BLOCK / 64 : 1][SWAP ? 2 : 1][SWAP ? 64 : 1];
  /* tile_cull: pairs a camera ray of this tile can reach, per chunk of 64 entries (scenes of more entries than the array
   * covers go without the culling: cull_ok) */
  constexpr uint32_t CULL_WORDS = GEOM_LDS ? PT_FILT_LDS_MAX / 64 : 256u;
  __shared__ uint32_t tile_pairs[CULL_WORDS];
  __shared__ uint32_t wg_next_job;                      /* SWAP: jobs of the tile's pool handed out so far */

#ifdef PT_PHASE
  if ((threadIdx.x & 63u) == 0u)
  {
    for (int k = 0; k < PT_PHASE_SLOTS; k++)
      pt_phase_acc[threadIdx.x >> 6][k] = 0;
    pt_phase_last[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime();
  }
#endif
  SceneCtx S_init = stage_scene<GEOM_LDS, FILT_LDS>(L, lds);
  __shared__ double atan_tab[CHECKER ? PT_ATAN_TAB : 1];
  if (CHECKER)
  {
    atan_table_to_lds(atan_tab);
    S_init.atan_tab = atan_tab;
  }
  PHASE(8); /* prologue: staging (scene -> LDS) */
  __shared__ __attribute__((aligned(16))) float big_tab[12]; /* BigPrune: delta, tmin, qmin of the leading wall-sized spheres */
  if (SWAP && FILT_LDS && !TRIS && L.big_pairs != 0u)
  {
    if (threadIdx.x < 2 + 2 * PT_BIG_PAIRS)
      big_tab[threadIdx.x] = threadIdx.x == 0 ? L.big_delta : (threadIdx.x == 1 ? L.big_tmin : L.big_qmin[threadIdx.x - 2]);
    S_init.big = BigPrune{big_tab, L.big_pairs};
  }
  const SceneCtx S = S_init;
  if (threadIdx.x < 2)
    wg_stats[threadIdx.x] = 0;
  if (threadIdx.x == 2)
    wg_next_job = 0;
  if (threadIdx.x < 3)
    pix_nan[threadIdx.x] = 0;
  if (!REFR && threadIdx.x < PT_TILE_PIXELS * 3)
    pix_sum[threadIdx.x] = 0;
  if (REFR)
  {
    for (uint32_t k = threadIdx.x; k < PT_TILE_PIXELS * 3 * PT_WIN_N; k += PT_BLOCK)
      pix_win[k] = 0;
    if (threadIdx.x < 2 * (PT_BLOCK / 64))
      pend_free[threadIdx.x >> 1][threadIdx.x & 1u] = ~0ull;
    if (threadIdx.x == 0)
      pend_slot_lds = pt_pool_acquire(L.pend_flags, L.pend_slots_per_xcd, L.status, PT_FAIL_PEND_SLOT);
  }
  if (threadIdx.x < PT_TILE_PIXELS)
  {
    const uint32_t t0 = L.tile_first + (blockIdx.x % L.tile_count) * L.tile_stride;
    const uint32_t kx = (t0 % L.tiles_x) * PT_TILE + (threadIdx.x & 7u), ky = (t0 / L.tiles_x) * PT_TILE + (threadIdx.x >> 3);
    pix_key[threadIdx.x] = rt_rng_pixel_key(L.seed, ky * (uint32_t)L.width + kx);
  }
  camera_to_lds(L, cam_lds);
  PHASE(9); /* prologue: pixel keys, camera, wall table */
  __syncthreads();
  PHASE(10); /* prologue: first barrier */

  /* ---- this wave's pixels and job pool ---- */
  const uint32_t wave = threadIdx.x >> 6;
  /* grid = tile_count x sample_chunks, chunk-major: consecutive workgroups are different
   * tiles, so the chunks of an expensive tile are spread over the launch */
  const uint32_t slot = blockIdx.x % L.tile_count, chunk = blockIdx.x / L.tile_count;
  const uint32_t tile = L.tile_first + slot * L.tile_stride;
  const bool cull_ok = S.n_sph + S.n_tri <= 64u * CULL_WORDS;
  if (SWAP && FILT_LDS && cull_ok)
  { /* the primitives a camera ray of this tile can reach at all: what the filter of a PRIMARY trip looks at */
    tile_cull(cam_lds, L.scene.entry_src, S.n_sph, S.n_sph + S.n_tri, (tile % L.tiles_x) * PT_TILE, (tile / L.tiles_x) * PT_TILE, tile_pairs);
    __syncthreads();
  }
  /* SWAP kernels: the four waves draw their 64-job batches from ONE pool, the tile's 64 pixels x samples (an LDS
   * counter): whichever wave is free takes the next batch, so the waves finish together whatever the rows of the
   * tile cost (a batch = one sample index of every pixel of the tile: its rays span exactly the tile's cone).  The
   * others keep round 2's split: wave w owns tile rows 2w, 2w + 1 and their samples. */
  const uint32_t tx0 = (tile % L.tiles_x) * PT_TILE, ty0 = (tile / L.tiles_x) * PT_TILE + (SWAP ? 0u : 2u * wave);
  /* valid sub-rectangle of the tile / of the wave's 8x2 strip (edge tiles of ragged images) */
  const uint32_t vcols = min((uint32_t)PT_TILE, (uint32_t)L.width - tx0);
  const uint32_t vrows = ty0 >= (uint32_t)L.height ? 0u : min(SWAP ? (uint32_t)PT_TILE : 2u, (uint32_t)L.height - ty0);
  const uint32_t n_valid = vcols * vrows;
  const uint32_t spp = (uint32_t)L.samples;
  /* this workgroup's share of the samples: [s_begin, s_end) of every pixel */
  const uint32_t s_begin = (uint32_t)(((uint64_t)chunk * spp) / L.sample_chunks);
  const uint32_t s_end = (uint32_t)(((uint64_t)(chunk + 1u) * spp) / L.sample_chunks);
  const uint32_t pool_jobs = n_valid * (s_end - s_begin); /* jobs: j -> pixel j % n_valid, sample s_begin + j / n_valid */

  Path P;
  P.o = {0, 0, 0};
  P.d = {0, 0, 1};
  P.T = {1, 1, 1};
  P.Ls = {0, 0, 0};
  P.rng = 1;
  P.depth = 0;
  uint32_t n_rays = 0, n_casts = 0;
  /* small-mesh kernels keep throughput and radiance in LDS across the scan (see the loop) */
  constexpr bool PARK_T = TRIS && FILT_LDS && !CHECKER;
  __shared__ double t_park[PARK_T ? 3 : 1][PARK_T ? PT_BLOCK : 1];
  HitRec hit;
  hit.min_t = 0;
  hit.bary_u = 0;
  hit.bary_v = 0;
  hit.best = -1;
  hit.depth_ok = false;
  hit.need_dir = false;
  hit.dir_slot = 0;
  hit.dir_scale = 1.0;
  hit.last.idx = -1;
  hit.last.u = 0;
  hit.last.v = 0;
  bool mesh_wait = false;
  uint32_t trip = 0, wait_since = 0xFFFFFFFFu; /* wave-uniform */
  uint32_t next_job = 0;     /* jobs handed out so far (wave-uniform) */
  uint32_t made_jobs = 0;    /* jobs whose camera ray sits in the wave's queue (wave-uniform) */
  uint32_t pix_slot = 0;     /* 0..63 inside the tile */
  bool busy = false;
  int stack_n = 0; /* REFR: pending second children of this lane's path (else no pending-ray stack in this body) */
  uint32_t pend_id = 0xFFu; /* REFR: the path's stack id, 0xFF = none yet */
  const PendStack no_stack = {nullptr, 0, 0u, 0u};
  /* REFR: this wave's 128 stacks in the workgroup's pool slot, [id][entry][field] (PendStack); no slot (a sizing bug of the
   * pool; RT_HIP_POOL_SLOTS=1 of the development build provokes it): the tile comes out NaN, as in the static body, and the launch is reported as
   * failed through the status word (pt_pool_acquire) */
  const uint32_t pend_slot = REFR ? pend_slot_lds : 0u;
  const bool pend_ok = !REFR || pend_slot != 0xFFFFFFFFu;
  const uint32_t pool = pend_ok ? pool_jobs : 0u;
  double *const pend_wave = REFR && pend_ok ? L.pend_ws + (size_t)pend_slot * L.pend_slot_doubles +
                                                  (size_t)(threadIdx.x >> 6) * 128u * L.pend_entries * PT_PEND_FIELDS
                                            : nullptr;
  unsigned long long *diag_ptr = L.stats;
  (void)diag_ptr;
  const uint32_t lane = threadIdx.x & 63u;
  double *const qd = q_dir[wave];
  unsigned long long *const qr = q_rng[wave];
  uint32_t *const qp = q_pix[wave];

  uint32_t n_wait = 0; /* SWAP: paths in this wave's waiting list (wave-uniform) */
  double *const wf = &w_f[SWAP ? wave : 0][0][0];
  uint32_t *const wu = &w_u[SWAP ? wave : 0][0][0];
  PHASE(11); /* prologue: tile_cull, its barrier, the wave's set-up */
  for (;;)
  {
    /* wave-uniform: this trip every busy lane holds a fresh camera ray of this tile (SWAP kernels) */
    bool primary_trip = false;
    if (SWAP)
    {
      /* ---- idle lanes take work: wave-synchronous, deterministic ----
       * Round 2 handed idle lanes camera rays that the whole wave had prepared 64 at a time; the FIRST BOUNCE of those
       * rays then ran in ordinary trips, a fifth of the lanes at a time, mixed with incoherent rays.  But camera rays
       * are the one coherent population there is: one origin, 64 directions inside one tile's narrow cone.  So the
       * wave now SWAPS: when lanes are idle, nobody waits in the list and jobs remain, every busy lane puts its path
       * on the wave's waiting list in LDS (o, d, T, RNG state, depth: 84 bytes) and ALL 64 lanes start fresh camera
       * samples -- a PRIMARY TRIP: full occupancy, a filter that only looks at the primitives the tile's cone can
       * reach (tile_cull: typically 3-5 pairs of the headline scene's 19), exact tests on coherent rays.  The fresh
       * paths that survive their first bounce stay in their lanes; lanes that fall idle in later trips pick up the
       * waiting paths (last in, first out), and when the list is dry the wave swaps again.  A sample's value depends
       * on its (seed, pixel, sample) stream alone and pixel sums are integers, so none of this can change a value. */
      unsigned long long idle = __ballot(!busy);
      if (idle != 0 && n_wait != 0u)
      {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (!busy && rank < n_wait)
        {
          const uint32_t e = n_wait - 1u - rank;
          P.o = {wf[0 * 64 + e], wf[1 * 64 + e], wf[2 * 64 + e]};
          P.d = {wf[3 * 64 + e], wf[4 * 64 + e], wf[5 * 64 + e]};
          P.T = {wf[6 * 64 + e], wf[7 * 64 + e], wf[8 * 64 + e]};
          P.rng = (uint64_t)__double_as_longlong(wf[9 * 64 + e]);
          if (CHECKER)
            hit.dir_scale = wf[(CHECKER ? 10 : 0) * 64 + e];
          const uint32_t meta = wu[e];
          hit.dir_slot = wu[64 + e];
          pix_slot = meta & 63u;
          hit.need_dir = (meta & 64u) != 0u;
          P.depth = (int)((meta >> 7) & 63u); /* (max_depth <= 32 in scenes with M_REFRACTION, rt_hip_render_tiles_chunked; others carry no more bits) */
          if (REFR)
          {
            stack_n = (int)((meta >> 13) & 63u);
            pend_id = (meta >> 19) & 0xFFu;
          }
          else
            P.depth = (int)(meta >> 7);
          busy = true;
        }
        n_wait -= min((uint32_t)__popcll(idle), n_wait);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        idle = __ballot(!busy);
      }
      PHASE(6); /* (of the trip's head: idle lanes take waiting paths from the list) */
      uint32_t batch = 0;
      if (idle != 0 && next_job < pool)
      { /* (idle lanes are left only when the list is dry: n_wait == 0 here) take the tile's next batch of 64 jobs */
        if (lane == 0)
          batch = atomicAdd(&wg_next_job, 64u);
        batch = (uint32_t)__builtin_amdgcn_readfirstlane((int)batch);
        next_job = batch < pool ? 0u : pool; /* the pool is dry: never ask again */
      }
      if (idle != 0 && next_job < pool)
      {
        /* the swap */
        const unsigned long long bm = __ballot(busy);
        if (busy)
        {
          const uint32_t e = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
          wf[0 * 64 + e] = P.o.x; wf[1 * 64 + e] = P.o.y; wf[2 * 64 + e] = P.o.z;
          wf[3 * 64 + e] = P.d.x; wf[4 * 64 + e] = P.d.y; wf[5 * 64 + e] = P.d.z;
          wf[6 * 64 + e] = P.T.x; wf[7 * 64 + e] = P.T.y; wf[8 * 64 + e] = P.T.z;
          wf[9 * 64 + e] = __longlong_as_double((long long)P.rng);
          if (CHECKER)
            wf[(CHECKER ? 10 : 0) * 64 + e] = hit.dir_scale;
          wu[e] = ((uint32_t)P.depth << 7) | (hit.need_dir ? 64u : 0u) | pix_slot |
                  (REFR ? (((uint32_t)stack_n << 13) | (pend_id << 19)) : 0u);
          wu[64 + e] = hit.dir_slot;
        }
        n_wait = (uint32_t)__popcll(bm);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        PHASE(7); /* (of the trip's head: the swap -- batch counter, busy lanes to the list) */
        const uint32_t job = batch + lane;
        busy = job < pool;
        if (busy)
        {
          DIAG(6, 1);
          DIAG_LANES(7);
          uint32_t idx;
          uint64_t term;
          if (n_valid == PT_TILE_PIXELS)
          {
            idx = job & 63u;
            /* (a batch of a full tile is one sample index of every pixel: wave-uniform, sample_term) */
            term = sample_term_uniform(s_begin + (job >> 6));
          }
          else
          {
            const uint32_t s = job / n_valid;
            idx = job - s * n_valid;
            term = sample_term(s_begin + s);
          }
          const uint32_t row = (vcols == PT_TILE) ? (idx >> 3) : (idx / vcols);
          const uint32_t col = idx - __umul24(row, vcols); /* (v_mul_u32_u24: full rate) */
          pix_slot = row * PT_TILE + col;
          start_sample(P, load_camera_lds(cam_lds), pix_key[pix_slot], tx0 + col, ty0 + row, term);
          hit.need_dir = false;
          if (REFR)
          { /* (the lane's previous path gave its id back when it ended, or took it along to the list) */
            stack_n = 0;
            pend_id = 0xFFu;
          }
        }
        primary_trip = FILT_LDS;
      }
    }
    else
    {
      /* ---- hand out jobs to idle lanes: wave-synchronous, deterministic ----
       * Idle lanes take jobs next_job, next_job + 1, ... in lane order.  The camera rays are
       * not generated by the few lanes that happen to be idle (about a fifth of the wave per
       * trip: start_sample would run on every trip at 20 % lane occupancy) but 64 at a time by
       * the whole wave into a queue in LDS, from which idle lanes only copy. */
      unsigned long long idle = __ballot(!busy);
      while (idle != 0 && next_job < pool)
      {
        if (next_job == made_jobs)
        {
          /* queue empty: every lane, busy or not, prepares job made_jobs + lane */
          const uint32_t job = made_jobs + lane;
          if (job < pool)
          {
            DIAG(6, 1);
            DIAG_LANES(7);
            uint32_t idx, s;
            if (n_valid == 16)
            {
              idx = job & 15u;
              s = job >> 4;
            }
            else
            {
              s = job / n_valid;
              idx = job - s * n_valid;
            }
            const uint32_t row = (vcols == PT_TILE) ? (idx >> 3) : (idx / vcols);
            const uint32_t col = idx - __umul24(row, vcols); /* (v_mul_u32_u24: full rate) */
            const uint32_t slot_in_tile = (2u * wave + row) * PT_TILE + col;
            Path Q;
            start_sample(Q, load_camera_lds(cam_lds), pix_key[slot_in_tile], tx0 + col, ty0 + row, sample_term(s_begin + s));
            qd[lane] = Q.d.x;
            qd[64 + lane] = Q.d.y;
            qd[128 + lane] = Q.d.z;
            qr[lane] = Q.rng;
            qp[lane] = slot_in_tile;
          }
          made_jobs = min(made_jobs + 64u, pool);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        const uint32_t job = next_job + rank;
        if (!busy && job < made_jobs)
        {
          const uint32_t q = job & 63u; /* batches start at multiples of 64 */
          P.o = load_camera_pos_lds(cam_lds);
          P.d = {qd[q], qd[64 + q], qd[128 + q]};
          P.rng = qr[q];
          pix_slot = qp[q];
          P.T = {1, 1, 1};
          P.Ls = {0, 0, 0};
          P.depth = 0;
          busy = true;
        }
        next_job = min(next_job + (uint32_t)__popcll(idle), made_jobs);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        idle = __ballot(!busy);
      }
    }
    if (__ballot(busy) == 0)
      break; /* pool dry and every lane drained (an idle lane would have taken a waiting path): the one exit, reached by all lanes together */
    const uint32_t *const prim_pairs = (SWAP && FILT_LDS && primary_trip && cull_ok) ? tile_pairs : nullptr;
    PHASE(0); /* the rest of the trip's head: the camera samples of a swap (start_sample) */

    bool step_done = false;
    /* lanes still sampling a direction from an earlier trip sit this trip's step out */
    const bool stepping = busy && !hit.need_dir;
    if (DEFER_MESH)
    {
      /* Scenes with a triangle hierarchy: only about a tenth of the rays enter the mesh's
       * bounds at all, and a walk costs several times a whole sphere-only trip -- done on the
       * spot it would run at ~8 % lane occupancy.  So a lane whose ray can reach the mesh
       * (bvh_probe) WAITS with its flat-scan result; the wave walks the hierarchy when enough
       * lanes wait (PT_MESH_BATCH), when nobody else can advance, or when the oldest has
       * waited PT_MESH_MAX_WAIT trips.  Waiting costs idle lanes in the trips between, a
       * batch runs the walk at several times the occupancy.  Results do not depend on when a
       * ray is walked. */
      if (stepping && !mesh_wait)
      {
        DIAG(0, 1);
        DIAG_LANES(1);
        n_rays++;
        (void)trace_step<1, false, CHECKER, TRIS, FILT_LDS, 1, true>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit);
        const bool far_origin = !(v_dot(P.o, P.o) <= S.near_R2);
        /* stale_uv: every triangle the ray passes matters, not only those closer than min_t (TriLast) */
        mesh_wait = hit.depth_ok && bvh_probe(S.bvh_nodes, S.n_bvh_nodes, far_origin, P.o, P.d,
                                              (CHECKER && S.stale_uv) ? S.t_start : hit.min_t, S.mesh_bound);
      }
      const uint32_t n_wait = (uint32_t)__popcll(__ballot(busy && mesh_wait));
      const uint32_t n_go = (uint32_t)__popcll(__ballot(busy && !mesh_wait));
      if (n_wait != 0 && wait_since == 0xFFFFFFFFu)
        wait_since = trip;
      const bool walk = n_wait != 0 && (n_wait >= PT_MESH_BATCH || n_go == 0 || trip - wait_since >= PT_MESH_MAX_WAIT);
      if (walk)
      {
        if (busy && mesh_wait)
        {
          const bool far_origin = !(v_dot(P.o, P.o) <= S.near_R2);
          bvh_traverse<CHECKER>(S.bvh_nodes, S.n_bvh_nodes, S.bvh_tri, S.tri, S.n_sph, far_origin, P.o, P.d, hit.min_t,
                                hit.best, hit.bary_u, hit.bary_v, diag_ptr, &hit.last, S.stale_uv, nullptr, S.tri32);
          mesh_wait = false;
        }
        wait_since = 0xFFFFFFFFu;
      }
      trip++;
      if (stepping && !mesh_wait)
        step_done = trace_step<1, false, CHECKER, TRIS, FILT_LDS, 2, true>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit);
    }
    else if (PARK_T)
    {
      /* small-mesh kernels: the scan (filter, fp32 pre-test, exact sphere and triangle tests) needs
       * every register it can get and does not touch the throughput, which sits in LDS while it runs
       * (the compiler otherwise spills registers to scratch around it): trace_path() in its two halves,
       * as in the hierarchy kernels */
      if (stepping)
      {
        DIAG(0, 1);
        DIAG_LANES(1);
        n_rays++;
        t_park[0][threadIdx.x] = P.T.x;
        t_park[1][threadIdx.x] = P.T.y;
        t_park[2][threadIdx.x] = P.T.z;
        asm volatile("" ::: "memory"); /* no store-to-load forwarding: the values must leave the registers */
        (void)trace_step<1, false, CHECKER, TRIS, FILT_LDS, 1, true>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit, prim_pairs);
        asm volatile("" ::: "memory");
        P.T = {t_park[0][threadIdx.x], t_park[1][threadIdx.x], t_park[2][threadIdx.x]};
        step_done = trace_step<1, false, CHECKER, TRIS, FILT_LDS, 2, true>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit);
      }
    }
    else if (stepping)
    {
      DIAG(0, 1);      /* wave-level loop iterations */
      DIAG_LANES(1);   /* lanes alive in them */
      n_rays++;
      if (REFR)
      {
        const PoolStack mine = {pend_wave, pend_free[wave], &pend_id, (int)L.pend_entries};
        step_done = trace_step<1, true, CHECKER, TRIS, FILT_LDS, 0, true, false, FILT_LDS && !GEOM_LDS, PoolStack>(S, P, n_casts, diag_ptr, mine, stack_n, &hit, prim_pairs);
      }
      else
        step_done = trace_step<1, false, CHECKER, TRIS, FILT_LDS, 0, true, false, FILT_LDS && !GEOM_LDS>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit, prim_pairs);
    }
    PHASE(3); /* hit record, roulette, material */
    /* ---- directions of diffuse hits: PT_DIR_ROUNDS rejection rounds per trip ----
     * A lane needs 1.91 rounds on average, but a loop that runs until the wave's last lane has
     * its sample takes ~6.2 (the maximum of ~45 geometric variables) at 20 % lane occupancy.
     * Here every lane that needs a direction -- from this trip's hit or still from an earlier
     * one -- gets PT_DIR_ROUNDS rounds; the ~5 % left without a sample carry on next trip and
     * skip that trip's step.  A sample depends on its stream alone, not on the trip it is
     * drawn in.  (No 100-round cap here: a lane that keeps failing simply keeps its turn; the
     * reference aborts at 100, probability 1e-32.) */
    if (busy && hit.need_dir)
    {
      V3 q;
      double len2;
      bool again = true;
      for (int round = 0; round < PT_DIR_ROUNDS && again; round++)
      {
        DIAG(10, 1);
        DIAG_LANES(11);
        again = rejection_round(P.rng, q, len2);
#if PT_DIR_MIN_LANES > 1
        /* (wave-uniform among the lanes still in the loop: they all leave together) */
        if (__popcll(__ballot(again)) < PT_DIR_MIN_LANES)
          break;
#endif
      }
      if (!again)
      {
        const double *m = S.mat + PT_MAT_STRIDE * (hit.dir_slot & ~PT_HULL_PLUS);
        V3 albedo = ld3(m + 1);
        if (CHECKER)
          albedo = v_scale(albedo, hit.dir_scale);
        const V3 n = P.d;
        double weight;
        P.d = hemisphere_from_sample(q, len2, n, weight);
        P.T = v_mul(P.T, v_scale(albedo, weight));
        hit.need_dir = false;
      }
    }
    PHASE(4); /* direction rounds */
    if (busy)
    {
      /* This trip's radiance terms (emission of a hit that goes on, or what ends the path) go to the
       * pixel's fixed-point sum at once: integer adds commute and associate, so the sum depends
       * neither on which lane finishes first nor on how a sample's terms are grouped -- and no
       * radiance lives in registers from one trip to the next. */
      if ((int)(P.Ls.x != 0.0) | (int)(P.Ls.y != 0.0) | (int)(P.Ls.z != 0.0))
      {
        if (REFR)
        { /* no bound on a term here: the windowed sums (win_add); a non-finite or oversized term flags the pixel.  (What they cost:
           * the same kernel adding plain fixed-point terms instead -- wrong for large terms, a timing experiment -- 19.9 against 20.3 ms) */
          unsigned long long *const pw = &pix_win[__umul24(pix_slot, 3u * PT_WIN_N)];
          if (P.Ls.x != 0.0 && !win_add(pw, P.Ls.x)) atomicOr(&pix_nan[0], 1ull << pix_slot);
          if (P.Ls.y != 0.0 && !win_add(pw + PT_WIN_N, P.Ls.y)) atomicOr(&pix_nan[1], 1ull << pix_slot);
          if (P.Ls.z != 0.0 && !win_add(pw + 2 * PT_WIN_N, P.Ls.z)) atomicOr(&pix_nan[2], 1ull << pix_slot);
        }
        else
        {
        /* (3 * pix_slot through v_mul_u32_u24: the compiler's v_mul_lo_u32 issues at a quarter of the rate) */
        unsigned long long *const px = &pix_sum[__umul24(pix_slot, 3u)];
        atomicAdd(&px[0], fixed_term(P.Ls.x, L.acc_scale));
        atomicAdd(&px[1], fixed_term(P.Ls.y, L.acc_scale));
        atomicAdd(&px[2], fixed_term(P.Ls.z, L.acc_scale));
        /* a NaN term (a ray through a degenerate normal, say) has no integer: flag the pixel, see finish_pixels */
        if ((int)(P.Ls.x != P.Ls.x) | (int)(P.Ls.y != P.Ls.y) | (int)(P.Ls.z != P.Ls.z))
        {
          if (P.Ls.x != P.Ls.x) atomicOr(&pix_nan[0], 1ull << pix_slot);
          if (P.Ls.y != P.Ls.y) atomicOr(&pix_nan[1], 1ull << pix_slot);
          if (P.Ls.z != P.Ls.z) atomicOr(&pix_nan[2], 1ull << pix_slot);
        }
        }
      }
      if (step_done)
      {
        busy = false;
        if (REFR && pend_id != 0xFFu)
        { /* the sample is complete (its stack is empty): the id goes back */
          pend_id_give(pend_free[wave], pend_id);
          pend_id = 0xFFu;
        }
      }
    }
    P.Ls = {0, 0, 0};
    PHASE(5); /* radiance to the pixel sums */
  }

  if (n_rays)
  {
    atomicAdd(&wg_stats[0], (unsigned long long)n_rays);
    atomicAdd(&wg_stats[1], (unsigned long long)n_casts);
  }
  __syncthreads();
  PHASE(12); /* epilogue: waiting for the workgroup's other waves */

  if (REFR)
  {
    if (!pend_ok && threadIdx.x < 3)
      pix_nan[threadIdx.x] = ~0ull;
    __syncthreads();
    if (L.sample_chunks == 1)
    {
      /* thread = (pixel, channel), as finish_pixels: the windowed sum -> mean -> float + tonemapped byte */
      if (threadIdx.x < PT_TILE_PIXELS * 3)
      {
        const uint32_t t = threadIdx.x / 3u, c = threadIdx.x - 3u * t;
        const bool inside = (tile % L.tiles_x) * PT_TILE + (t & 7u) < (uint32_t)L.width && (tile / L.tiles_x) * PT_TILE + (t >> 3) < (uint32_t)L.height;
        win_normalize(&pix_win[threadIdx.x * PT_WIN_N]);
        double mean = win_value(&pix_win[threadIdx.x * PT_WIN_N]) * (1.0 / (double)L.samples);
        mean = ((pix_nan[c] >> t) & 1ull) ? __longlong_as_double(0x7FF8000000000000ll) : mean;
        out_f[threadIdx.x] = inside ? (float)mean : 0.f;
        out_b[threadIdx.x] = inside ? tonemap(mean) : 0;
      }
      __syncthreads();
      store_tile(L, out_f, out_b, wg_stats, tile, slot, S.n_sph + S.n_tri, true, true);
    }
    else
    {
      /* one of several sample chunks of this tile: its windows, carry-normalised (words below 2^32: the tile's record takes one
       * piece per chunk and word), are added to the tile's record in HBM -- integer atomics: exact, order-free;
       * pt_resolve_tiles normalises the total and finishes the pixels */
      if (threadIdx.x < PT_TILE_PIXELS * 3)
      {
        unsigned long long *const w = &pix_win[threadIdx.x * PT_WIN_N];
        win_normalize(w);
        unsigned long long *const acc = L.acc_ws + ((size_t)slot * (PT_TILE_PIXELS * 3) + threadIdx.x) * PT_WIN_N;
#pragma unroll
        for (int k = 0; k < PT_WIN_N; k++)
          if (w[k] != 0ull)
            atomicAdd(&acc[k], w[k]);
      }
      if (threadIdx.x < 3 && pix_nan[threadIdx.x] != 0)
        atomicOr(&L.acc_ws[(size_t)L.tile_count * (PT_TILE_PIXELS * 3 * PT_WIN_N) + (size_t)slot * 3 + threadIdx.x], pix_nan[threadIdx.x]);
      store_tile(L, out_f, out_b, wg_stats, tile, slot, S.n_sph + S.n_tri, false, chunk == 0);
    }
    if (pend_ok && threadIdx.x == 0)
      atomicExch(&L.pend_flags[pend_slot], 0u); /* every lane is past its last pop (the barriers above) */
  }
  else if (L.sample_chunks == 1)
  {
    finish_pixels(L, pix_sum, pix_nan, tile, out_f, out_b);
    __syncthreads();
    store_tile(L, out_f, out_b, wg_stats, tile, slot, S.n_sph + S.n_tri, true, true);
    PHASE(13); /* epilogue: mean, tonemap, tile store */
#ifdef PT_PHASE
    /* one workgroup in 32 reports: same-address atomics from every wave would queue at one L2 channel and show up
     * in the very phases measured (they did: 480 k atomics on config 2, prologue and epilogue each 3x too long) */
    if ((threadIdx.x & 63u) == 0u && L.stats && (blockIdx.x & 31u) == 0u)
      for (int k = 0; k < PT_PHASE_SLOTS; k++)
        atomicAdd(&L.stats[64 + k], pt_phase_acc[threadIdx.x >> 6][k]);
#endif
  }
  else
  {
    /* one of several sample chunks of this tile: add the partial sums to the tile's record in
     * HBM (integer atomics: exact, order-independent); pt_resolve_tiles finishes the pixels */
    if (threadIdx.x < PT_TILE_PIXELS * 3 && pix_sum[threadIdx.x] != 0)
      atomicAdd(&L.acc_ws[(size_t)slot * (PT_TILE_PIXELS * 3) + threadIdx.x], pix_sum[threadIdx.x]);
    /* the NaN flags follow the sums of all tiles in the workspace */
    if (threadIdx.x < 3 && pix_nan[threadIdx.x] != 0)
      atomicOr(&L.acc_ws[(size_t)L.tile_count * (PT_TILE_PIXELS * 3) + (size_t)slot * 3 + threadIdx.x], pix_nan[threadIdx.x]);
    store_tile(L, out_f, out_b, wg_stats, tile, slot, S.n_sph + S.n_tri, false, chunk == 0);
  }
}

#endif /* PT_BODY_POOLED_H */
