/* pt_body_queued.h -- render_tiles_queued: hierarchy scenes -- the pooled body with PARKED walks (rays that can reach the mesh go to a per-wave ring in
 * global memory and are walked together, walk_parked).
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_BODY_QUEUED_H
#define PT_BODY_QUEUED_H

/* ---- hierarchy scenes: pooled samples + PARKED walks (pt_render_tiles_tri_queued[_chk][_sph][_refr]) ---------
 *
 * Scenes with a triangle hierarchy (more than PT_FILT_LDS_MAX primitives).  The pooled body above
 * makes a ray that can reach the mesh WAIT in its lane until enough lanes wait, then walks the
 * hierarchy with the waiting lanes only: the lanes in between idle (loop occupancy 55 % on config
 * 5) and a walk batch holds ~32 rays whose lengths range from 2 to 25 visits (12 % of the lanes
 * busy inside walks; profiles/archive/r02a_c5_pmc.txt: 30 % VALU lane utilisation overall).  Here such a
 * ray is PARKED instead: its state (origin, direction, throughput, RNG state, flat-scan result:
 * 96 bytes) goes to a per-wave ring in global memory and its lane takes the next job at once.
 * When PT_PARK_WALK rays are parked the whole wave turns to walking them: every lane takes a ray
 * from the ring, and a lane whose walk ends takes the next one (the walk lengths average out
 * over the ~2-4 rays a lane gets through), node visits and leaf tests batched apart
 * ("while-while" with refill).  Walked rays are picked up by idle lanes ahead of fresh camera
 * samples and continue with the shading half of trace_step.  None of this can change a value: a
 * sample depends on its (seed, pixel, sample) stream alone, per-pixel sums are integers.
 *
 * The ring: PT_PARK_Q entries of 128 bytes per wave, positions [head, head + n_done) hold walked rays, then n_new parked
 * ones; all three counters are wave-uniform.  It lives in a workspace slot the workgroup takes
 * from a pool at entry and returns at exit (pt_park_acquire): the pool is partitioned by XCD
 * (HW_REG_XCC_ID of the running wave, not an assumption about placement), so every owner a slot
 * ever has sits behind the same L2 -- plain stores, L1-bypassing loads, no cache write-backs.
 * Radiance is added to the pixel's fixed-point sum term by term (P.Ls is flushed every trip),
 * so a parked ray carries no partial radiance. */
/* Ring entries per wave (a power of two).  512, not 256, for a GUARANTEE: every live path of a wave is in exactly one place --
 * a lane (<= 64), the waiting list (<= 64), or the ring -- and new paths come only from a swap, which needs an empty list, no
 * walked ray left in the ring and fewer than PT_PARK_WALK parked ones: at most 63 + (PT_PARK_WALK - 1) paths live before it,
 * 64 more after.  So the ring never holds more than PT_PARK_WALK + 126 rays; with 512 entries it is never full, a ray that
 * wants a walk is always parked at once, and the `waiting` state below (a ray keeps its lane until the ring has room) cannot
 * occur -- it could otherwise starve a wave whose every lane waits while paths sit on its list.  (Measured against 256
 * entries, which a mesh-filling view could fill: same time, ring traffic 73 -> 75 GB per 4K x 256 spp launch.) */
#ifndef PT_PARK_Q
#define PT_PARK_Q 512u
#endif
#ifndef PT_PARK_WALK
#define PT_PARK_WALK 256u /* parked rays that turn the wave to walking (round 2's kernel at 4K x 256 spp: 32: 664 ms, 64: 553, 128: 529, 190: 521;
                          * round 3's last, with 512 ring entries: 128: 222.9, 190: 220.5, 256: 219.2, 320: 219.1, 384: 219.0 -- 256 is also the best at 64 spp) */
#endif
#ifndef PT_STAGE
#define PT_STAGE 32u /* walked rays copied from the ring to LDS at a time (<= 64) */
#endif
#ifndef PT_REFILL_BATCH
#define PT_REFILL_BATCH 16u /* free lanes that trigger a refill from the ring inside a walk phase */
#endif
#ifndef PT_LEAF_BATCH
#define PT_LEAF_BATCH 32u /* lanes holding a leaf that trigger a round of exact triangle tests (16: 223.1 ms, 24: 220.5, 32: 220.1 at 4K x 256 spp) */
#endif
#define PT_PARK_F64_FIELDS 13u /* o xyz, d xyz, T xyz, rng, min_t, (M_CHECKERED kernels: last u, v) */
#define PT_PARK_U32_FIELDS 4u  /* best, depth << 6 | pixel slot, (last index), pad */
static_assert(PT_PARK_WAVE_BYTES >= PT_PARK_Q * 128u + PT_TILE_PIXELS * 8u + PT_PARK_WIN_BYTES && PT_PARK_F64_FIELDS * 8u + PT_PARK_U32_FIELDS * 4u <= 128u, "ring bytes per wave");
static_assert(PT_PARK_WIN_BYTES == PT_TILE_PIXELS * 3u * 6u * 8u, "the refraction form's windows: PT_WIN_N words per pixel channel");
static_assert((PT_PARK_Q & (PT_PARK_Q - 1u)) == 0u && PT_PARK_WALK + 126u <= PT_PARK_Q, "ring size: see PT_PARK_Q");

/* Entry-major, in three regions per wave, by who touches what:
 *   HOT  [PT_PARK_Q] x 64 bytes: o, d, min_t, best, depth/pixel -- all the WALKER reads (one 64-byte line per ray) and
 *        writes (min_t, best: the same line), and what a resume reads first;
 *   COLD [PT_PARK_Q] x 32 bytes: T, RNG state -- written at the park, read at the resume, never seen by the walker;
 *   CHK  [PT_PARK_Q] x 32 bytes: hit.u / hit.v state of the M_CHECKERED kernels (TriLast).
 * Round 2 kept one 128-byte record per ray: the walker's loads pulled the cold half of every line through the L2 as
 * well, and its 12-byte result dirtied a 128-byte line. */
struct ParkRing
{
  double *f;   /* the wave's PT_PARK_Q x 128 bytes */
  uint32_t *u; /* the same memory as words */
};
__device__ __forceinline__ uint32_t ring_fi(uint32_t field, uint32_t e)
{ /* fields: 0-2 o, 3-5 d, 6-8 T, 9 rng, 10 min_t, 11-12 last u, v */
  return field < 6u ? e * 8u + field
                    : (field == 10u ? e * 8u + 6u
                                    : (field < 10u ? PT_PARK_Q * 8u + e * 4u + (field - 6u) : PT_PARK_Q * 12u + e * 4u + (field - 11u)));
}
__device__ __forceinline__ uint32_t ring_ui(uint32_t field, uint32_t e)
{ /* fields: 0 best, 1 depth << 6 | pixel slot (+ PT_DIAG flags), 2 last index, 3 (REFR kernels) stack id | stack height << 16 */
  return field < 2u ? e * 16u + 14u + field : PT_PARK_Q * 24u + e * 8u + 2u + field;
}

/* ring loads bypass the vector L1 (agent-scope relaxed = `sc1`): a slot's earlier owner on this CU
 * may have left lines of it there */
__device__ __forceinline__ double ring_ld(const ParkRing &r, uint32_t field, uint32_t e)
{
  return __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<unsigned long long *>(r.f + ring_fi(field, e)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ V3 ring_ld3(const ParkRing &r, uint32_t field, uint32_t e)
{
  return {ring_ld(r, field, e), ring_ld(r, field + 1u, e), ring_ld(r, field + 2u, e)};
}
__device__ __forceinline__ uint32_t ring_ldu(const ParkRing &r, uint32_t field, uint32_t e)
{
  return __hip_atomic_load(r.u + ring_ui(field, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ring_st(const ParkRing &r, uint32_t field, uint32_t e, double v) { r.f[ring_fi(field, e)] = v; }
__device__ __forceinline__ void ring_stu(const ParkRing &r, uint32_t field, uint32_t e, uint32_t v) { r.u[ring_ui(field, e)] = v; }
__device__ __forceinline__ void ring_st3(const ParkRing &r, uint32_t field, uint32_t e, const V3 &v)
{
  ring_st(r, field, e, v.x);
  ring_st(r, field + 1u, e, v.y);
  ring_st(r, field + 2u, e, v.z);
}

/* The workgroup's workspace slot, or 0xFFFFFFFF when there is none: no workspace (cannot happen: without one
 * pt_launch_render refuses the parked-walk kernels), or -- a sizing bug of the pool (pt_pool_slots_per_xcd gives it 25 % more
 * slots than workgroups can be resident; RT_HIP_POOL_SLOTS=1 of the development build gives it one) -- every slot of this XCD taken after a
 * bounded search.  The waves of such a workgroup render nothing and say so twice: PT_FAIL_PARK_SLOT in the device's status word,
 * which fails the launch for the host (rt_hip_launch_status), and every pixel of their tiles NaN (bytes 255;
 * render_tiles_queued) -- rather than spin for ever or walk rays from registers the kernel does not have.  Thread 0 only. */
__device__ __forceinline__ uint32_t lane_of_thread() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t pt_park_acquire(const PtLaunch &L)
{
  return pt_pool_acquire(L.park_ws == nullptr ? nullptr : L.park_flags, L.park_slots_per_xcd, L.status, PT_FAIL_PARK_SLOT);
}

/* The per-lane traversal stacks of the parked-walk kernels: 24-bit entries (a 16-bit and an 8-bit array, entry-major,
 * one entry per tree level and lane), because at four workgroups per CU every kilobyte of LDS counts there.  A reference
 * fits 24 bits while node indices stay below 2^23 and meshes below 2^(23 - PT_BVH_COUNT_BITS) triangles
 * (checked on the host, pt_pick_kernel: other meshes take the lane-waiting kernels). */
struct WalkStack
{
  uint16_t *lo; /* [levels][PT_BLOCK] */
  uint8_t *hi;  /* [levels][PT_BLOCK] */
};
#define PT_WALK_LEAF_FLAG24 0x800000u
__device__ __forceinline__ uint32_t walk_ref24(uint32_t ref) /* PT_BVH_LEAF_FLAG (bit 31) moves to bit 23 */
{
  return (ref & 0x7FFFFFu) | ((ref >> 8) & PT_WALK_LEAF_FLAG24);
}
__device__ __forceinline__ uint32_t walk_ref32(uint32_t r24) { return (r24 & 0x7FFFFFu) | ((r24 & PT_WALK_LEAF_FLAG24) << 8); }
__device__ __forceinline__ void walk_push(const WalkStack &st, uint32_t sp, uint32_t ref)
{
  const uint32_t r = walk_ref24(ref);
  st.lo[sp * PT_BLOCK + threadIdx.x] = (uint16_t)r;
  st.hi[sp * PT_BLOCK + threadIdx.x] = (uint8_t)(r >> 16);
}
__device__ __forceinline__ uint32_t walk_pop(const WalkStack &st, uint32_t sp)
{
  return walk_ref32((uint32_t)st.lo[sp * PT_BLOCK + threadIdx.x] | ((uint32_t)st.hi[sp * PT_BLOCK + threadIdx.x] << 16));
}


/* The wave walks the n_new parked rays at ring positions first, first + 1, ... (see the header
 * comment): refill, then either one node visit for the lanes that hold an inner node or the exact
 * triangle tests of the lanes that hold a leaf, until every ray has its result in the ring.
 * The caller has put every path the lanes held on the waiting list (render_tiles_queued): nothing of the trip loop
 * is live in registers while the wave walks. */
template <bool CHECKER>
__device__ __forceinline__ void walk_parked(const SceneCtx &S, const ParkRing &ring, uint32_t first, uint32_t n_new,
                                            const WalkStack &stack, unsigned long long *diag_ptr)
{
  /* parked state written by this wave's lanes (plain stores) must have reached L2 before other
   * lanes load it: workgroup-scope release = s_waitcnt vmcnt(0) */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  uint32_t next = 0; /* rays handed to lanes so far (wave-uniform) */
  bool have = false;
  uint32_t e = 0, sp = 0, ref = 0;
  V3 wo = {0, 0, 0}, wd = {0, 0, 1};
  double wmin_t = 0, bu = 0, bv = 0;
  int wbest = -1;
  bool far_origin = false;
  BvhRay R = bvh_ray(wo, wd);
  TriLast last = {-1, 0, 0};
  const bool no_prune = CHECKER && S.stale_uv;
  float wtmax = 0.f;        /* a float not below wmin_t (float_above), renewed when wmin_t changes: what the slab tests prune by */
#ifdef PT_DIAG
  uint32_t visits = 0;
  int wbest0 = wbest;
  bool outside_bound = false; /* the probe's bounding sphere would have kept this ray out: it must find nothing */
  bool origin_inside = false;
#endif
  for (;;)
  {
    /* refill in batches: every refill is a memory round trip the whole wave waits for, so free lanes
     * wait until PT_REFILL_BATCH of them are free (or nobody has a ray left) */
    const unsigned long long need = __ballot(!have);
    if (next < n_new && ((uint32_t)__popcll(need) >= PT_REFILL_BATCH || need == ~0ull))
    {
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
      if (!have && next + rank < n_new)
      {
        e = (first + next + rank) & (PT_PARK_Q - 1u);
        wo = ring_ld3(ring, 0u, e);
        wd = ring_ld3(ring, 3u, e);
        wmin_t = ring_ld(ring, 10u, e);
        wbest = (int)ring_ldu(ring, 0u, e);
        wtmax = no_prune ? 3.4028234663852886e38f : float_above(wmin_t);
        R = bvh_ray(wo, wd);
        far_origin = !(v_dot(wo, wo) <= S.near_R2);
        sp = 0;
        ref = 0; /* the root */
        last.idx = -1;
        have = true;
#ifdef PT_DIAG
        visits = 0;
        wbest0 = wbest;
        outside_bound = (ring_ldu(ring, 1u, e) & 0x80000000u) != 0u;
        origin_inside = (ring_ldu(ring, 1u, e) & 0x40000000u) != 0u;
#endif
      }
      next = min(n_new, next + (uint32_t)__popcll(need));
    }
    if (__ballot(have) == 0)
      break; /* every ray walked: the one exit, reached by all lanes together */
    /* Steps until the next refill is due, in a loop of their own: what belongs to the lane's ray (origin and direction in
     * fp64 and in the slab test's fp32 form, its ring entry: 27 registers) does not change in here.  In one loop with the
     * refill, which assigns them under a lane mask, the compiler moved all of them to other registers and back on every
     * iteration -- some forty v_mov per node visit of sixty instructions. */
    for (;;)
    {
    const unsigned long long active = __ballot(have);
    const bool at_leaf = have && (ref & PT_BVH_LEAF_FLAG) != 0u;
    const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
    const uint32_t n_inner = (uint32_t)__popcll(active) - n_leaf;
    bool finished = false;
    if (n_inner != 0u && n_leaf < PT_LEAF_BATCH)
    {
      if (have && !at_leaf)
      {
        DIAG(13, 1);
        DIAG_LANES(15);
#ifdef PT_DIAG
        visits++;
#endif
        bool hit0, hit1;
        float tn0, tn1;
        uint32_t r0, r1;
        bvh_test_children(S.bvh_nodes, ref, R, far_origin, wtmax, hit0, hit1, tn0, tn1, r0, r1);
        if (hit0 && hit1)
        {
          const bool zero_first = !(tn1 < tn0);
          walk_push(stack, sp, zero_first ? r1 : r0);
          sp++;
          ref = zero_first ? r0 : r1;
        }
        else if (hit0 || hit1)
          ref = hit0 ? r0 : r1;
        else if (sp == 0)
          finished = true;
        else
        {
          sp--;
          ref = walk_pop(stack, sp);
        }
      }
    }
    else if (at_leaf)
    {
      const uint32_t first_tri = (ref & ~PT_BVH_LEAF_FLAG) >> PT_BVH_COUNT_BITS, count = ref & ((1u << PT_BVH_COUNT_BITS) - 1u);
      uint32_t keep = leaf_pretest(S.tri32, first_tri, count, far_origin, R, wd, diag_ptr);
#ifdef PT_DIAG
      for (uint32_t k = 0; k < count; k++) /* re-check: a dropped triangle must fail the exact test */
      {
        double t_probe = 1.7976931348623157e308, pu = 0, pv = 0;
        int b_probe = -1;
        exact_triangle(S.tri_leaf + 9 * (size_t)(first_tri + k), 0u, wo, wd, t_probe, b_probe, pu, pv);
        if (!((keep >> k) & 1u) && b_probe >= 0)
          atomicAdd(&diag_ptr[4 + 12], 1ull);
      }
#endif
      while (keep != 0u)
      {
        DIAG(14, 1);
        DIAG_LANES(41);
        const uint32_t k = (uint32_t)__builtin_ctz(keep);
        keep &= keep - 1u;
        const uint32_t t = S.bvh_tri[first_tri + k];
        exact_triangle<true, CHECKER, true>(S.tri_leaf + 9 * (size_t)(first_tri + k), S.n_sph + t, wo, wd, wmin_t, wbest, bu, bv, &last); /* (parked-walk kernels: no wide-range scene) */
      }
      if (!no_prune)
        wtmax = float_above(wmin_t);
      if (sp == 0)
        finished = true;
      else
      {
        sp--;
        ref = walk_pop(stack, sp);
      }
    }
    if (finished)
    {
#ifdef PT_DIAG
      /* walked rays: those that come back with a triangle; walks of 1, 2-3, 4-6, more node visits */
      if (wbest >= (int)S.n_sph)
        atomicAdd(&diag_ptr[4 + 18], 1ull);
      /* parked rays by where they start (inside the bounding ball or not) and whether the walk found a closer triangle */
      atomicAdd(&diag_ptr[4 + 24 + (origin_inside ? 0 : 2) + (wbest != wbest0 ? 0 : 1)], 1ull);
      if (outside_bound && (wbest != wbest0 || last.idx >= 0))
        atomicAdd(&diag_ptr[4 + 12], 1ull); /* a violation of the conservative probe */
      atomicAdd(&diag_ptr[4 + (visits <= 1u ? 19 : (visits <= 3u ? 20 : (visits <= 6u ? 21 : 22)))], 1ull);
#endif
      ring_st(ring, 10u, e, wmin_t);
      ring_stu(ring, 0u, e, (uint32_t)wbest);
      if (CHECKER)
      {
        ring_stu(ring, 2u, e, (uint32_t)last.idx);
        ring_st(ring, 11u, e, last.u);
        ring_st(ring, 12u, e, last.v);
      }
      have = false;
    }
    const unsigned long long free_now = __ballot(!have);
    if (free_now == ~0ull || (next < n_new && (uint32_t)__popcll(free_now) >= PT_REFILL_BATCH))
      break; /* nobody holds a ray any more, or a batch of lanes is free and rays are left: back to the refill */
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* results in L2 before anyone resumes them */
}

/* REFR (pt_render_tiles_tri_queued_refr[_sph], end of round 4): hierarchy scenes with M_REFRACTION materials, until then on the
 * static body (every lane walks the hierarchy on the spot).  What render_tiles_pooled's REFR form needs, here: pixel sums without a
 * bound on a term (win_add: six signed 64-bit windows per channel -- 9 KB per wave's tile, kept at the end of the wave's
 * workspace region and added to with global integer atomics, see below), and pending second children in stacks addressed by a path id that travels with the
 * path -- through the waiting list (meta word) and, new here, through the ring (word 3 of an entry): a wave can hold 64 paths in
 * lanes, 64 on its list and up to 382 in its ring, so ids are 9 bits (PoolStackT<512, 511>: 511 ids for at most 510 paths) and a
 * workgroup's slot of the pending-ray pool holds 4 x 512 stacks (PtLaunch.pend_slot_doubles: pend_pool_for sizes it). */
typedef PoolStackT<512u, 511u> RingStack;
/* GEOM_LDS = false (pt_render_tiles_tri_queued_mem*): scenes whose spheres exceed the LDS staging budget AND that have a mesh of
 * more than 256 triangles -- sphere geometry and materials gathered from memory, the spheres' sign-form filter pairs read from
 * memory by scalar loads (as pt_render_tiles_pool_mem_s does for sphere-only scenes); everything else as the staged form. */
template <bool CHECKER, bool SPHERE_PROBE = false, bool REFR = false, bool GEOM_LDS = true>
__device__ __forceinline__ void render_tiles_queued(const PtLaunch &L)
{
  static_assert(!REFR || CHECKER, "the refraction form carries every material's code");
  constexpr bool TRIS = true, FILT_LDS = false;
  constexpr uint32_t NO_ID = 511u;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  /* ONE WAVE = ONE TILE here (a workgroup = four tiles, its waves independent of each other between the
   * barrier after staging and the one before the slot goes back): a wave's pool is its tile's 64 pixels
   * x samples, four times the 16-pixel strips of round 2's pooled body, so the tail in which the last paths of a
   * pool run on with most lanes idle -- each walk costs a park / walk / resume cycle, so the tail is long in
   * these kernels -- weighs a quarter as much.  (The image is 4K-sized or the scene's cost per ray is high
   * wherever these kernels run, so a quarter as many workgroups still fill the chip many times over.) */
  __shared__ unsigned long long pix_sum_all[REFR ? 1 : PT_BLOCK / 64][REFR ? 1 : PT_TILE_PIXELS * 3];
  __shared__ unsigned long long pix_nan_all[PT_BLOCK / 64][3];
  __shared__ unsigned long long pend_free[REFR ? PT_BLOCK / 64 : 1][8]; /* REFR: per wave, the free ids of its 511 pending-ray stacks */
  __shared__ uint32_t park_slot_lds, pend_slot_lds;
  __shared__ double cam_lds[PT_CAM_LDS_DOUBLES]; /* the camera (camera_to_lds) */
  /* The wave's WAITING LIST in LDS: up to 64 paths that wait for a lane (render_tiles_pooled's, with two more
   * tenants).  Who puts paths there: (1) the SWAP -- idle lanes, an empty list, jobs left: every busy lane leaves
   * its path here and all 64 lanes start fresh camera samples, a PRIMARY trip; (2) walked rays on their way back:
   * up to 64 at a time are copied from the ring in memory (one round trip for the batch; a few idle lanes taking
   * them straight from the ring would put that round trip at the head of every trip) and resume with the second
   * half of trace_step; (3) every path the lanes hold when the wave turns to WALKING the parked rays: the walk
   * needs the registers, and with the paths in LDS nothing of the trip loop is live while it runs -- round 2's kernel
   * spilled 136 bytes per lane to scratch memory around the walk, 40 GB per 4K x 256 spp frame.
   * Entry: o, d, T, RNG state, min_t of a scanned ray (or the checker factor of a pending direction) [+ hit.u / hit.v
   * state]; meta word; best of a scanned ray or material slot of a pending direction. */
  constexpr uint32_t WAIT_F = CHECKER ? 13u : 11u, WAIT_U = CHECKER ? 3u : 2u;
  __shared__ double w_f[PT_BLOCK / 64][WAIT_F][64];
  __shared__ uint32_t w_u[PT_BLOCK / 64][WAIT_U][64];
  /* meta: pixel slot (6 bits), then: a direction is still to be sampled; a walked ray (scan result known, second half
   * next); the ray leaves a hull facet for good; a scanned ray that found the ring full and waits to be parked (scan
   * result known, park next); then the depth */
  constexpr uint32_t META_NEED_DIR = 64u, META_RESUMED = 128u, META_LEAVING = 256u, META_WAITING = 512u, META_DEPTH_SHIFT = 10u;
  /* REFR: depth in six bits (max_depth <= 32 in scenes with M_REFRACTION), then the path's stack id (nine bits) and stack height (six) */
  constexpr uint32_t META_ID_SHIFT = 16u, META_STACK_SHIFT = 25u;

#ifdef PT_PHASE
  if ((threadIdx.x & 63u) == 0u)
  {
    for (int k = 0; k < PT_PHASE_SLOTS; k++)
      pt_phase_acc[threadIdx.x >> 6][k] = 0;
    pt_phase_last[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime();
  }
#endif
  SceneCtx S_init = stage_scene<GEOM_LDS, FILT_LDS, true>(L, lds);
  __shared__ double atan_tab[CHECKER ? PT_ATAN_TAB : 1];
  if (CHECKER)
  {
    atan_table_to_lds(atan_tab);
    S_init.atan_tab = atan_tab;
  }
  PHASE(8);
  __shared__ __attribute__((aligned(16))) float big_tab[12]; /* BigPrune: delta, tmin, qmin of the leading wall-sized spheres */
  if (L.big_pairs != 0u)
  {
    if (threadIdx.x < 2 + 2 * PT_BIG_PAIRS)
      big_tab[threadIdx.x] = threadIdx.x == 0 ? L.big_delta : (threadIdx.x == 1 ? L.big_tmin : L.big_qmin[threadIdx.x - 2]);
    S_init.big = BigPrune{big_tab, L.big_pairs};
  }
  const SceneCtx S = S_init;
  /* the traversal stacks follow the staged scene (geometry, materials, the spheres' filter pairs) in dynamic LDS:
   * one 24-bit entry per tree level and lane (WalkStack) */
  WalkStack stack;
  {
    const uint32_t levels = max(L.scene.bvh_depth, 1u);
    stack.lo = reinterpret_cast<uint16_t *>(lds + (GEOM_LDS ? (PT_GEOM_STRIDE * (size_t)S.n_sph + PT_MAT_STRIDE * (size_t)(L.scene.n_spheres + L.scene.n_meshes) +
                                                               pt_filt_pair_slots(S.n_sph))
                                                            : (size_t)0));
    stack.hi = reinterpret_cast<uint8_t *>(stack.lo + (size_t)levels * PT_BLOCK);
  }
  {
    unsigned long long *z = &pix_sum_all[0][0];
    for (uint32_t k = threadIdx.x; k < (REFR ? 1u : (PT_BLOCK / 64) * PT_TILE_PIXELS * 3); k += PT_BLOCK)
      z[k] = 0;
    if (threadIdx.x < (PT_BLOCK / 64) * 3)
      (&pix_nan_all[0][0])[threadIdx.x] = 0;
    if (REFR && threadIdx.x < (PT_BLOCK / 64) * 8)
      (&pend_free[0][0])[threadIdx.x] = (threadIdx.x & 7u) == 7u ? 0x7FFFFFFFFFFFFFFFull : ~0ull; /* ids 0 .. 510 */
  }
  if (threadIdx.x == 0)
  {
    park_slot_lds = pt_park_acquire(L);
    if (REFR)
      pend_slot_lds = pt_pool_acquire(L.pend_ws == nullptr ? nullptr : L.pend_flags, L.pend_slots_per_xcd, L.status, PT_FAIL_PEND_SLOT);
  }
  camera_to_lds(L, cam_lds);
  PHASE(9);
  __syncthreads();
  PHASE(10);

  const uint32_t wave = threadIdx.x >> 6;
  /* work units = tile_count x sample_chunks, chunk-major (consecutive units are different tiles; REFR: tile-major, below); wave w of
   * workgroup b takes unit 4 b + w; the last workgroup may have waves without a unit (pool = 0) */
  const uint32_t unit = blockIdx.x * (PT_BLOCK / 64) + wave;
  const bool has_unit = unit < L.tile_count * L.sample_chunks;
  /* (the refraction form: TILE-major -- a workgroup's four waves render four chunks of ONE tile: they walk the same part of the
   * hierarchy, and their windowed sums and pending-ray stacks touch the same lines; one rank's share of the glass mesh at N = 8,
   * 256 spp, four chunks: 45.9 -> 40.6 ms of an ideal 37.8.  The plain forms lose 3.5 % that way: profiles/r05_shard_order_experiments.txt) */
  constexpr bool TILE_MAJOR = REFR;
  const uint32_t slot = !has_unit ? 0u : (TILE_MAJOR ? unit / L.sample_chunks : unit % L.tile_count);
  const uint32_t chunk = !has_unit ? 0u : (TILE_MAJOR ? unit % L.sample_chunks : unit / L.tile_count);
  const uint32_t tile = L.tile_first + slot * L.tile_stride;
  const uint32_t tx0 = (tile % L.tiles_x) * PT_TILE, ty0 = (tile / L.tiles_x) * PT_TILE;
  const uint32_t vcols = min((uint32_t)PT_TILE, (uint32_t)L.width - tx0);
  const uint32_t vrows = min((uint32_t)PT_TILE, (uint32_t)L.height - ty0);
  const uint32_t n_valid = vcols * vrows;
  const uint32_t spp = (uint32_t)L.samples;
  const uint32_t s_begin = (uint32_t)(((uint64_t)chunk * spp) / L.sample_chunks);
  const uint32_t s_end = (uint32_t)(((uint64_t)(chunk + 1u) * spp) / L.sample_chunks);
  const uint32_t park_slot = park_slot_lds;
  const uint32_t pend_slot = REFR ? pend_slot_lds : 0u;
  /* (REFR: without its slot of the pending-ray pool -- a sizing bug of the pool, never seen -- a workgroup renders nothing either) */
  const bool ring_ok = park_slot != 0xFFFFFFFFu && (!REFR || pend_slot != 0xFFFFFFFFu);
  /* (the launcher takes these kernels only with a workspace: pt_launch_render; a slot can be missing only through a sizing
   * bug of the pool, never seen -- then nothing could be parked and rays that want a walk would wait for ever: the wave
   * renders nothing instead, and says so: every pixel of its tile comes out NaN, bytes 255) */
  const uint32_t pool = (has_unit && ring_ok) ? n_valid * (s_end - s_begin) : 0u;
  if (has_unit && !ring_ok && (threadIdx.x & 63u) < 3u)
    pix_nan_all[wave][threadIdx.x & 63u] = ~0ull;
  /* wave-uniform addresses and tile numbers that the trip loop needs now and then are formed where they are used, from
   * a wave index the compiler cannot see through (wave_now): hoisted out of the loop they each hold a vector register for
   * its whole length -- the kernel has none to spare at four waves per SIMD, they were what it spilled */
  auto wave_now = [] {
    uint32_t w = threadIdx.x >> 6;
    asm volatile("" : "+v"(w));
    return w;
  };
  ParkRing ring;
  {
    char *base = L.park_ws + ((size_t)(park_slot != 0xFFFFFFFFu ? park_slot : 0u) * (PT_BLOCK / 64) + wave) * PT_PARK_WAVE_BYTES;
    ring.f = reinterpret_cast<double *>(base);
    ring.u = reinterpret_cast<uint32_t *>(base);
  }
  /* REFR: the windowed pixel sums of this wave's tile live in the LAST bytes of its workspace region, not in LDS: in LDS the
   * four tiles' 36.9 KB left two workgroups per CU, and at two waves per SIMD this kernel waits (VALU busy 53 %: the walk's
   * dependent fetches, the ring's round trips).  Integer atomics at the XCD's L2 instead -- only lanes whose trip has a
   * non-zero term issue any, a few per cent of the L2's atomic rate --, zeroed here by the wave itself (its own earlier
   * stores and its later atomics reach the L2 in issue order), read back L1-bypassing in the epilogue. */
  auto pix_win_now = [&] {
    uint32_t off = PT_PARK_WAVE_BYTES - PT_PARK_WIN_BYTES;
    asm volatile("" : "+v"(off));
    return reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ring.f) + off);
  };
  auto pend_wave_now = [&] {
    return L.pend_ws + (size_t)pend_slot * L.pend_slot_doubles + (size_t)wave_now() * 512u * L.pend_entries * PT_PEND_FIELDS;
  };
  if (REFR && has_unit && ring_ok)
  {
    unsigned long long *const pix_win = pix_win_now();
    for (uint32_t k = lane_of_thread(); k < PT_TILE_PIXELS * 3 * PT_WIN_N; k += 64u)
      pix_win[k] = 0ull;
  }
  /* can a camera ray of this wave's tile reach the triangles' bounding ball at all?  (tile_cull's cone test, for the probe's
   * own ball: its r2_hi is the radius squared plus the filter's widening, far more than the centre's rounding to fp32) */
  const bool tile_sees_mesh =
      tile_cone_reaches_ball(cam_lds, tx0, ty0, V3{(double)S.mesh_bound.cx, (double)S.mesh_bound.cy, (double)S.mesh_bound.cz},
                             sqrt((double)S.mesh_bound.r2_hi) * (1.0 + 1e-5) + 1e-300);
  /* The tile's 64 per-pixel RNG keys (rt_rng_pixel_key: a splitmix64 finaliser, six quarter-rate multiplies) are formed
   * once, by lane = pixel slot, and kept behind the wave's ring (this kernel has neither a register pair nor 512 bytes of
   * LDS per wave to spare for them); a swap reads its lane's key back -- one load that hits the XCD's L2 -- instead
   * of hashing it again for every camera sample. */
  if (has_unit && ring_ok)
  {
    const uint32_t kx = tx0 + (lane_of_thread() & 7u), ky = ty0 + (lane_of_thread() >> 3);
    reinterpret_cast<unsigned long long *>(ring.f + PT_PARK_Q * 16u)[lane_of_thread()] = rt_rng_pixel_key(L.seed, ky * (uint32_t)L.width + kx);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* in L2 before any lane reads a key another lane wrote */
  }

  Path P;
  P.o = {0, 0, 0};
  P.d = {0, 0, 1};
  P.T = {1, 1, 1};
  P.Ls = {0, 0, 0};
  P.rng = 1;
  P.depth = 0;
  uint32_t n_rays = 0, n_casts = 0;
  HitRec hit;
  hit.min_t = 0;
  hit.bary_u = 0;
  hit.bary_v = 0;
  hit.best = -1;
  hit.depth_ok = false;
  hit.need_dir = false;
  hit.dir_slot = 0;
  hit.dir_scale = 1.0;
  hit.leaving = false;
  hit.last.idx = -1;
  hit.last.u = 0;
  hit.last.v = 0;
  uint32_t next_job = 0;                    /* camera samples started so far (wave-uniform) */
  uint32_t head = 0, n_done = 0, n_new = 0; /* the ring (wave-uniform) */
  uint32_t n_wait = 0;                      /* paths on the waiting list (wave-uniform) */
  uint32_t pix_slot = 0;
  bool busy = false;
  bool waiting = false; /* the lane's ray is scanned and wants a walk, but the ring was full: park it next trip */
  int stack_n = 0;        /* REFR: pending second children of this lane's path */
  uint32_t pend_id = NO_ID; /* REFR: the path's stack id */
  const PendStack no_stack = {nullptr, 0, 0u, 0u};
  /* (the wave's stacks in the pool slot and -- below -- its windows: addresses formed where they are used, see wave_now) */
  unsigned long long *diag_ptr = L.stats;
  (void)diag_ptr;
  const uint32_t lane = threadIdx.x & 63u;
  double *const wf = &w_f[wave][0][0];
  uint32_t *const wu = &w_u[wave][0][0];

  /* a busy lane's path -> list entry e (the swap, and before a walk) */
  auto put_on_list = [&](uint32_t e, bool resumed_now) {
    wf[0 * 64 + e] = P.o.x; wf[1 * 64 + e] = P.o.y; wf[2 * 64 + e] = P.o.z;
    wf[3 * 64 + e] = P.d.x; wf[4 * 64 + e] = P.d.y; wf[5 * 64 + e] = P.d.z;
    wf[6 * 64 + e] = P.T.x; wf[7 * 64 + e] = P.T.y; wf[8 * 64 + e] = P.T.z;
    wf[9 * 64 + e] = __longlong_as_double((long long)P.rng);
    uint32_t meta = ((uint32_t)P.depth << META_DEPTH_SHIFT) | (hit.need_dir ? META_NEED_DIR : 0u) | (hit.leaving ? META_LEAVING : 0u) | pix_slot;
    if (REFR)
      meta |= (pend_id << META_ID_SHIFT) | ((uint32_t)stack_n << META_STACK_SHIFT);
    if (resumed_now || waiting)
    { /* the scan's result travels with the ray */
      meta |= resumed_now ? META_RESUMED : META_WAITING;
      wf[10 * 64 + e] = hit.min_t;
      wu[64 + e] = (uint32_t)hit.best;
      if (CHECKER)
      {
        wf[(CHECKER ? 11 : 0) * 64 + e] = hit.last.u;
        wf[(CHECKER ? 12 : 0) * 64 + e] = hit.last.v;
        wu[(CHECKER ? 2 : 0) * 64 + e] = (uint32_t)hit.last.idx;
      }
    }
    else
    {
      wu[64 + e] = hit.dir_slot;
      if (CHECKER)
        wf[10 * 64 + e] = hit.dir_scale;
    }
    wu[e] = meta;
  };

  for (;;)
  {
    /* ---- idle lanes take work: waiting paths first (walked rays among them: that frees the ring), then, when the
     * list and the ring's walked part are empty, the swap (render_tiles_pooled) ---- */
    unsigned long long idle = __ballot(!busy);
    bool resumed = false;
    bool primary_trip = false; /* wave-uniform */
    for (int pass = 0; pass < 2 && idle != 0; pass++)
    {
      if (n_wait == 0u && n_done != 0u)
      {
        /* the next walked rays: lane l copies ring entry head + l to list entry l */
        const uint32_t k = min(64u, n_done);
        if (lane < k)
        {
          const uint32_t e = (head + lane) & (PT_PARK_Q - 1u);
          /* all thirteen loads first, then the stores: written as load / store pairs the compiler keeps each (atomic) load
           * and the LDS store of its value in program order, i.e. thirteen memory round trips one after the other */
          double fv[11];
#pragma unroll
          for (uint32_t f = 0; f < 11u; f++)
            fv[f] = ring_ld(ring, f, e);
          const uint32_t best_w = ring_ldu(ring, 0u, e);
          const uint32_t dp = ring_ldu(ring, 1u, e) & 0x3FFFFFFFu; /* depth << 6 | pixel slot; bits 31, 30: PT_DIAG's flags */
#pragma unroll
          for (uint32_t f = 0; f < 11u; f++)
            wf[f * 64u + lane] = fv[f];
          wu[64u + lane] = best_w;
          uint32_t meta_w = (dp & 63u) | META_RESUMED | ((dp >> 6) << META_DEPTH_SHIFT);
          if (REFR)
          {
            const uint32_t idw = ring_ldu(ring, 3u, e);
            meta_w |= ((idw & 0x1FFu) << META_ID_SHIFT) | ((idw >> 16) << META_STACK_SHIFT);
          }
          wu[lane] = meta_w;
          if (CHECKER)
          {
            wf[(CHECKER ? 11u : 0u) * 64u + lane] = ring_ld(ring, 11u, e);
            wf[(CHECKER ? 12u : 0u) * 64u + lane] = ring_ld(ring, 12u, e);
            wu[(CHECKER ? 2u : 0u) * 64u + lane] = ring_ldu(ring, 2u, e);
          }
        }
        n_wait = k;
        head = (head + k) & (PT_PARK_Q - 1u);
        n_done -= k;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (n_wait == 0u)
        break;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      if (!busy && rank < n_wait)
      {
        const uint32_t e = n_wait - 1u - rank;
        P.o = {wf[0 * 64 + e], wf[1 * 64 + e], wf[2 * 64 + e]};
        P.d = {wf[3 * 64 + e], wf[4 * 64 + e], wf[5 * 64 + e]};
        P.T = {wf[6 * 64 + e], wf[7 * 64 + e], wf[8 * 64 + e]};
        P.rng = (uint64_t)__double_as_longlong(wf[9 * 64 + e]);
        const double f10 = wf[10 * 64 + e];
        const uint32_t meta = wu[e], w1 = wu[64 + e];
        pix_slot = meta & 63u;
        P.depth = REFR ? (int)((meta >> META_DEPTH_SHIFT) & 63u) : (int)(meta >> META_DEPTH_SHIFT);
        if (REFR)
        {
          pend_id = (meta >> META_ID_SHIFT) & 0x1FFu;
          stack_n = (int)((meta >> META_STACK_SHIFT) & 63u);
        }
        P.Ls = {0, 0, 0};
        hit.need_dir = (meta & META_NEED_DIR) != 0u;
        hit.leaving = (meta & META_LEAVING) != 0u;
        waiting = (meta & META_WAITING) != 0u;
        if (meta & (META_RESUMED | META_WAITING))
        { /* the scan's result is known: a walked ray goes on with the second half, a waiting one with the park */
          hit.min_t = f10;
          hit.best = (int)w1;
          hit.depth_ok = true;
          if (CHECKER)
          {
            hit.last.u = wf[(CHECKER ? 11 : 0) * 64 + e];
            hit.last.v = wf[(CHECKER ? 12 : 0) * 64 + e];
            hit.last.idx = (int)wu[(CHECKER ? 2 : 0) * 64 + e];
          }
          resumed = (meta & META_RESUMED) != 0u;
        }
        else
        {
          hit.dir_slot = w1;
          if (CHECKER)
            hit.dir_scale = f10;
        }
        busy = true;
      }
      n_wait -= min((uint32_t)__popcll(idle), n_wait);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      idle = __ballot(!busy);
    }
    PHASE(6); /* walked rays from the ring to the list; idle lanes take waiting paths */
    /* enough rays are parked (or the ring is full): the wave owes them a walk.  It happens as soon as the list is
     * empty -- every live path is then in a lane and the list can take them all; until then no swap brings new paths */
    const bool walk_due = n_new >= PT_PARK_WALK || n_new + n_done >= PT_PARK_Q;
    if (!walk_due && idle != 0 && n_wait == 0u && n_done == 0u && next_job < pool)
    {
      /* the swap: busy lanes leave their paths on the list, all 64 lanes start fresh camera samples */
      const unsigned long long bm = __ballot(busy);
      if (busy)
        put_on_list(__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)), resumed);
      n_wait = (uint32_t)__popcll(bm);
      resumed = false;
      waiting = false;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t job = next_job + lane;
      busy = job < pool;
      if (REFR)
      { /* fresh samples: no stack yet */
        pend_id = NO_ID;
        stack_n = 0;
      }
      if (busy)
      {
        DIAG(6, 1);
        DIAG_LANES(7);
        uint32_t idx;
        uint64_t term;
        if (n_valid == PT_TILE_PIXELS)
        {
          idx = job & 63u;
          term = sample_term_uniform(s_begin + (job >> 6)); /* (wave-uniform: see render_tiles_pooled) */
        }
        else
        { /* ragged edge tiles only: the divisors go through a register the compiler cannot see through, or it forms their
           * reciprocals ahead of the trip loop and keeps them (in scratch memory: the kernel has no register to spare) */
          uint32_t nv = n_valid;
          asm volatile("" : "+v"(nv));
          const uint32_t s = job / nv;
          idx = job - s * nv;
          uint32_t sv = s_begin + s;
          asm volatile("" : "+v"(sv)); /* (or the constant part of the product is formed ahead of the loop and kept, in scratch memory) */
          term = sample_term(sv);
        }
        uint32_t vc = vcols;
        if (vcols != PT_TILE)
          asm volatile("" : "+v"(vc));
        const uint32_t row = (vcols == PT_TILE) ? (idx >> 3) : (idx / vc);
        const uint32_t col = idx - __umul24(row, vcols); /* (v_mul_u32_u24: full rate) */
        pix_slot = row * PT_TILE + col;
        /* (the keys' offset behind the ring through a register the compiler cannot see through: hoisted out of the loop, the
         * sum would be one more address held for its whole length -- in scratch memory, this kernel has no register left) */
        uint32_t key_at = PT_PARK_Q * 16u + pix_slot;
        asm volatile("" : "+v"(key_at));
#ifndef PT_KEYS_RECOMPUTED /* (A/B: hash the key again for every camera sample, as before) */
        const uint64_t pixel_key = __hip_atomic_load(reinterpret_cast<unsigned long long *>(ring.f) + key_at, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT); /* (L1-bypassing, like every ring load) */
#else
        const uint64_t pixel_key = rt_rng_pixel_key(L.seed, (ty0 + row) * (uint32_t)L.width + tx0 + col);
#endif
        start_sample(P, load_camera_lds(cam_lds), pixel_key, tx0 + col, ty0 + row, term);
        hit.need_dir = false;
        hit.leaving = false;
      }
      next_job = min(next_job + 64u, pool);
      primary_trip = true;
    }
    PHASE(7); /* the swap and its camera samples */
    /* nobody holds a ray: the pool is dry, the list and the ring's walked part are empty (an idle lane would have
     * taken from them).  Parked rays, if any, are walked now; otherwise this is the one exit. */
    const bool drained = __ballot(busy) == 0;
    if (drained && n_new == 0u)
      break;

    /* ---- the wave turns to walking: every path the lanes hold goes to the list first, so that nothing of this
     * loop is live in registers while walk_parked runs ---- */
    if (drained || (walk_due && n_wait == 0u))
    {
      const unsigned long long bm = __ballot(busy);
      if (busy)
        put_on_list(__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)), resumed);
      n_wait = (uint32_t)__popcll(bm);
      busy = false;
      waiting = false;
      /* the lanes' paths are dead from here (they come back from the list): say so to the register allocator */
      P.o = {0, 0, 0};
      P.d = {0, 0, 1};
      P.T = {1, 1, 1};
      P.rng = 1;
      P.depth = 0;
      pix_slot = 0;
      pend_id = NO_ID;
      stack_n = 0;
      hit.min_t = 0;
      hit.best = -1;
      hit.need_dir = false;
      hit.leaving = false;
      hit.dir_slot = 0;
      hit.dir_scale = 1.0;
      hit.last.idx = -1;
      hit.last.u = 0;
      hit.last.v = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      PHASE(14); /* every path to the list before a walk */
      walk_parked<CHECKER>(S, ring, (head + n_done) & (PT_PARK_Q - 1u), n_new, stack, diag_ptr);
      PHASE(15); /* walking the parked rays */
      n_done += n_new;
      n_new = 0u;
      continue;
    }

    /* ---- first half of trace_path(): depth test + flat scan over the spheres, then the probe ---- */
    PHASE(0);
    const bool stepping = busy && !hit.need_dir && !resumed && !waiting;
    bool want_walk = waiting;
#ifdef PT_DIAG
    bool diag_in_sphere = true;
#endif
    if (stepping)
    {
      DIAG(0, 1);
      DIAG_LANES(1);
      n_rays++;
      /* a ray that left a hull facet on its outer side cannot meet a triangle: no probe, no walk (set by the
       * second half of the previous step; the first half does not touch it) */
      /* ... and a fresh camera ray of a tile whose cone cannot reach the triangles' bounding ball (tile_sees_mesh, once
       * per wave: a primary trip's 64 rays are all such rays) cannot either: most of the image's primary trips skip the probe */
      const bool no_mesh = (hit.leaving && !(CHECKER && S.stale_uv)) || (primary_trip && !tile_sees_mesh);
      (void)trace_step<1, false, CHECKER, TRIS, FILT_LDS, 1, true, true, !GEOM_LDS>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit); /* (the first half never touches the stack) */
      const bool far_origin = !(v_dot(P.o, P.o) <= S.near_R2);
#ifdef PT_DIAG
      /* RT_HIP_DIAG_WALK_REJECTED=1: rays the bounding sphere rejects are parked and walked all the same, and any
       * that comes back with a triangle counts as a violation; otherwise the build parks what the shipped one parks */
      want_walk = hit.depth_ok && bvh_probe<SPHERE_PROBE>(S.bvh_nodes, S.n_bvh_nodes, far_origin, P.o, P.d,
                                                          (CHECKER && S.stale_uv) ? S.t_start : hit.min_t, S.mesh_bound,
                                                          (L.diag_flags & 1u) ? &diag_in_sphere : nullptr);
      if (hit.depth_ok && !no_mesh)
        DIAG_LANES(42); /* lane-level probe evaluations of the shipped build */
      if (no_mesh)
      { /* walked all the same under RT_HIP_DIAG_WALK_REJECTED=1, and counted as a violation if it finds a triangle */
        diag_in_sphere = false;
        want_walk = want_walk && (L.diag_flags & 1u) != 0u;
        if (hit.leaving)
          DIAG_LANES(28); /* rays leaving a hull facet */
        else
          DIAG_LANES(38); /* camera rays of tiles that cannot see the mesh */
      }
#else
      want_walk = hit.depth_ok && !no_mesh &&
                  bvh_probe<SPHERE_PROBE>(S.bvh_nodes, S.n_bvh_nodes, far_origin, P.o, P.d,
                                          (CHECKER && S.stale_uv) ? S.t_start : hit.min_t, S.mesh_bound);
#endif
    }
    /* ---- rays that can reach the mesh are parked; their lanes are idle from here on.  A ray that finds the ring
     * full keeps its lane and its scan result and tries again next trip (`waiting`): a full ring makes the walk due,
     * so room comes within a few trips ---- */
    const unsigned long long wants = __ballot(want_walk);
    if (wants != 0)
    {
      const uint32_t space = PT_PARK_Q - n_done - n_new;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wants >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wants, 0u));
      if (want_walk && rank < space)
      {
        const uint32_t e = (head + n_done + n_new + rank) & (PT_PARK_Q - 1u);
        ring_st3(ring, 0u, e, P.o);
        ring_st3(ring, 3u, e, P.d);
        ring_st3(ring, 6u, e, P.T);
        ring_st(ring, 9u, e, __longlong_as_double((long long)P.rng));
        ring_st(ring, 10u, e, hit.min_t);
        ring_stu(ring, 0u, e, (uint32_t)hit.best);
#ifdef PT_DIAG
        const double diag_lx = (double)S.mesh_bound.cx - P.o.x, diag_ly = (double)S.mesh_bound.cy - P.o.y,
                     diag_lz = (double)S.mesh_bound.cz - P.o.z; /* bit 30: the ray starts inside the bounding ball */
        const bool diag_origin_inside = diag_lx * diag_lx + diag_ly * diag_ly + diag_lz * diag_lz <= (double)S.mesh_bound.r2_hi;
        /* (a ray that waited a trip for room lost its diag_in_sphere: it counts as inside, i.e. is not checked) */
        ring_stu(ring, 1u, e, ((uint32_t)P.depth << 6) | pix_slot | ((diag_in_sphere || waiting) ? 0u : 0x80000000u) |
                                  (diag_origin_inside ? 0x40000000u : 0u));
        if (diag_in_sphere)
          DIAG_LANES(23);
#else
        ring_stu(ring, 1u, e, ((uint32_t)P.depth << 6) | pix_slot);
#endif
        if (REFR)
          ring_stu(ring, 3u, e, pend_id | ((uint32_t)stack_n << 16));
        if (CHECKER)
        { /* (the walk overwrites these; a defined value for rays it finds nothing for) */
          ring_stu(ring, 2u, e, (uint32_t)hit.last.idx);
          ring_st(ring, 11u, e, hit.last.u);
          ring_st(ring, 12u, e, hit.last.v);
        }
        waiting = false;
        busy = false;
        DIAG_LANES(17);
      }
      else if (want_walk)
        waiting = true;
      n_new += min((uint32_t)__popcll(wants), space);
    }

    PHASE(11); /* depth test, the mesh probe, parking (what the sphere scan's two marks leave) */
    /* ---- second half: hit record, roulette, material -- for rays scanned now and not parked, and for
     * walked rays resumed at the top of this trip ---- */
    bool step_done = false;
    if (busy && (stepping || resumed) && !waiting)
    {
      if constexpr (REFR)
      {
        const RingStack mine = {pend_wave_now(), pend_free[wave_now()], &pend_id, (int)L.pend_entries};
        step_done = trace_step<1, true, CHECKER, TRIS, FILT_LDS, 2, true, true, !GEOM_LDS, RingStack>(S, P, n_casts, diag_ptr, mine, stack_n, &hit);
      }
      else
        step_done = trace_step<1, false, CHECKER, TRIS, FILT_LDS, 2, true, true, !GEOM_LDS>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit);
    }
    PHASE(3);

    /* ---- directions of diffuse hits (see render_tiles_pooled) ---- */
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
        hit.leaving = (hit.dir_slot & PT_HULL_PLUS) != 0u && weight > S.hull_margin; /* weight = the new direction . n */
      }
    }
    PHASE(4);
    if (busy)
    {
      /* this trip's radiance terms go to the pixel's fixed-point sum at once (integer adds commute and
       * associate: the sum does not depend on the order or the grouping of the terms) */
      if ((int)(P.Ls.x != 0.0) | (int)(P.Ls.y != 0.0) | (int)(P.Ls.z != 0.0))
      {
        if (REFR)
        { /* no bound on a term: the windowed sums (win_add); a non-finite or oversized term flags the pixel */
          unsigned long long *const pw = pix_win_now() + __umul24(pix_slot, 3u * PT_WIN_N);
          unsigned long long *const pix_nan = pix_nan_all[wave_now()];
          if (P.Ls.x != 0.0 && !win_add(pw, P.Ls.x)) atomicOr(&pix_nan[0], 1ull << pix_slot);
          if (P.Ls.y != 0.0 && !win_add(pw + PT_WIN_N, P.Ls.y)) atomicOr(&pix_nan[1], 1ull << pix_slot);
          if (P.Ls.z != 0.0 && !win_add(pw + 2 * PT_WIN_N, P.Ls.z)) atomicOr(&pix_nan[2], 1ull << pix_slot);
        }
        else
        {
        unsigned long long *const pix_sum = pix_sum_all[wave_now()];
        /* (3 * pix_slot through v_mul_u32_u24: the compiler's v_mul_lo_u32 issues at a quarter of the rate) */
        unsigned long long *const px = &pix_sum[__umul24(pix_slot, 3u)];
        atomicAdd(&px[0], fixed_term(P.Ls.x, L.acc_scale));
        atomicAdd(&px[1], fixed_term(P.Ls.y, L.acc_scale));
        atomicAdd(&px[2], fixed_term(P.Ls.z, L.acc_scale));
        if ((int)(P.Ls.x != P.Ls.x) | (int)(P.Ls.y != P.Ls.y) | (int)(P.Ls.z != P.Ls.z))
        {
          unsigned long long *const pix_nan = pix_nan_all[wave_now()];
          if (P.Ls.x != P.Ls.x) atomicOr(&pix_nan[0], 1ull << pix_slot);
          if (P.Ls.y != P.Ls.y) atomicOr(&pix_nan[1], 1ull << pix_slot);
          if (P.Ls.z != P.Ls.z) atomicOr(&pix_nan[2], 1ull << pix_slot);
        }
        }
        P.Ls = {0, 0, 0};
      }
      if (step_done)
      {
        busy = false;
        if (REFR && pend_id != NO_ID)
        { /* the sample is complete (its stack is empty): the id goes back */
          pend_id_give(pend_free[wave_now()], pend_id);
          pend_id = NO_ID;
        }
      }
    }
    PHASE(5);
  }

#ifdef PT_PHASE
  if ((threadIdx.x & 63u) == 0u && L.stats && (blockIdx.x & 31u) == 0u)
    for (int k = 0; k < PT_PHASE_SLOTS; k++)
      atomicAdd(&L.stats[64 + k], pt_phase_acc[threadIdx.x >> 6][k]);
#endif
  /* ---- this wave's tile: counters, then the pixels (thread = pixel) ---- */
  if (has_unit)
  {
    /* the tile's numbers once more (see wave_now) */
    const uint32_t unit_e = blockIdx.x * (PT_BLOCK / 64) + wave_now();
    const uint32_t slot = TILE_MAJOR ? unit_e / L.sample_chunks : unit_e % L.tile_count;
    const uint32_t chunk = TILE_MAJOR ? unit_e % L.sample_chunks : unit_e / L.tile_count;
    const uint32_t tile_e = L.tile_first + slot * L.tile_stride;
    const uint32_t vcols = min((uint32_t)PT_TILE, (uint32_t)L.width - (tile_e % L.tiles_x) * PT_TILE);
    const uint32_t vrows = min((uint32_t)PT_TILE, (uint32_t)L.height - (tile_e / L.tiles_x) * PT_TILE);
    const uint32_t n_valid = vcols * vrows;
    unsigned long long *const pix_sum = pix_sum_all[REFR ? 0u : wave_now()];
    unsigned long long *const pix_nan = pix_nan_all[wave_now()];
    uint32_t rays_w = n_rays, casts_w = n_casts;
    for (int off = 32; off > 0; off >>= 1)
    {
      rays_w += (uint32_t)__shfl_xor((int)rays_w, off);
      casts_w += (uint32_t)__shfl_xor((int)casts_w, off);
    }
    if (L.stats && lane == 0)
    {
      atomicAdd(&L.stats[0], (unsigned long long)rays_w);
      atomicAdd(&L.stats[1], (unsigned long long)casts_w);
      atomicAdd(&L.stats[2], (unsigned long long)casts_w * (unsigned long long)(S.n_sph + S.n_tri));
      if (chunk == 0)
        atomicAdd(&L.stats[3], (unsigned long long)n_valid * (unsigned long long)L.samples);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); /* the wave's own LDS atomics are done: sums are final */
    if (REFR)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); /* ... and its global ones have reached the L2 */
    __builtin_amdgcn_wave_barrier();
    if (L.sample_chunks == 1)
    {
      const uint32_t t = lane;
      const bool inside = (t & 7u) < vcols && (t >> 3) < vrows;
      const double inv_s = 1.0 / (double)L.samples;
      const double quiet_nan = __longlong_as_double(0x7FF8000000000000ll);
      V3 mean;
      if (REFR)
      {
        /* (the wave's atomics have reached the L2: agent-scope release = s_waitcnt vmcnt(0); the loads bypass the L1) */
        unsigned long long w[3 * PT_WIN_N];
#pragma unroll
        for (uint32_t k = 0; k < 3u * PT_WIN_N; k++)
          w[k] = __hip_atomic_load(pix_win_now() + t * (3u * PT_WIN_N) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        win_normalize(w);
        win_normalize(w + PT_WIN_N);
        win_normalize(w + 2 * PT_WIN_N);
        mean.x = win_value(w) * inv_s;
        mean.y = win_value(w + PT_WIN_N) * inv_s;
        mean.z = win_value(w + 2 * PT_WIN_N) * inv_s;
      }
      else
      {
      mean.x = ((double)(long long)pix_sum[3 * t + 0] * L.acc_inv_scale) * inv_s;
      mean.y = ((double)(long long)pix_sum[3 * t + 1] * L.acc_inv_scale) * inv_s;
      mean.z = ((double)(long long)pix_sum[3 * t + 2] * L.acc_inv_scale) * inv_s;
      }
      mean.x = ((pix_nan[0] >> t) & 1ull) ? quiet_nan : mean.x; /* see finish_pixels */
      mean.y = ((pix_nan[1] >> t) & 1ull) ? quiet_nan : mean.y;
      mean.z = ((pix_nan[2] >> t) & 1ull) ? quiet_nan : mean.z;
      float *of = L.tiles_rgb + (size_t)slot * (PT_TILE_PIXELS * 3) + 3 * t;
      of[0] = inside ? (float)mean.x : 0.f;
      of[1] = inside ? (float)mean.y : 0.f;
      of[2] = inside ? (float)mean.z : 0.f;
      if (L.tiles_rgb8)
      {
        uint8_t *ob = L.tiles_rgb8 + (size_t)slot * (PT_TILE_PIXELS * 3) + 3 * t;
        ob[0] = inside ? tonemap(mean.x) : 0;
        ob[1] = inside ? tonemap(mean.y) : 0;
        ob[2] = inside ? tonemap(mean.z) : 0;
      }
    }
    else if (REFR)
    {
      /* one of several sample chunks of this tile, windowed form: the wave's windows (in its workspace region; its atomics have
       * reached the L2, the loads bypass the L1), carry-normalised per pixel channel, added to the tile's record */
      for (uint32_t pc = lane; ring_ok && pc < PT_TILE_PIXELS * 3; pc += 64) /* (no slot: no windows of its own; the NaN masks below say so) */
      {
        unsigned long long w[PT_WIN_N];
#pragma unroll
        for (uint32_t k = 0; k < PT_WIN_N; k++)
          w[k] = __hip_atomic_load(pix_win_now() + pc * PT_WIN_N + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        win_normalize(w);
        unsigned long long *const acc = L.acc_ws + ((size_t)slot * (PT_TILE_PIXELS * 3) + pc) * PT_WIN_N;
#pragma unroll
        for (uint32_t k = 0; k < PT_WIN_N; k++)
          if (w[k] != 0ull)
            atomicAdd(&acc[k], w[k]);
      }
      if (lane < 3 && pix_nan[lane] != 0)
        atomicOr(&L.acc_ws[(size_t)L.tile_count * (PT_TILE_PIXELS * 3 * PT_WIN_N) + (size_t)slot * 3 + lane], pix_nan[lane]);
    }
    else
    {
      /* one of several sample chunks of this tile: exact integer partial sums to the tile's record */
      for (uint32_t k = lane; k < PT_TILE_PIXELS * 3; k += 64)
        if (pix_sum[k] != 0)
          atomicAdd(&L.acc_ws[(size_t)slot * (PT_TILE_PIXELS * 3) + k], pix_sum[k]);
      if (lane < 3 && pix_nan[lane] != 0)
        atomicOr(&L.acc_ws[(size_t)L.tile_count * (PT_TILE_PIXELS * 3) + (size_t)slot * 3 + lane], pix_nan[lane]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && park_slot != 0xFFFFFFFFu)
    atomicExch(&L.park_flags[park_slot], 0u); /* every wave is past its last ring access */
  if (REFR && threadIdx.x == 0 && pend_slot != 0xFFFFFFFFu)
    atomicExch(&L.pend_flags[pend_slot], 0u); /* ... and past its last pop */
}

#endif /* PT_BODY_QUEUED_H */
