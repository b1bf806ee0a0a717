/* pt_filter.h -- the scene scan of intersect() (raytracer.c:393-464) as filter-then-compact: the conservative packed-fp32 phase-1 filter in its
 * three forms, wall pruning (BigPrune), per-tile culling of primary trips, the fp32 triangle pre-test, scan_filtered.
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_FILTER_H
#define PT_FILTER_H

struct SceneCtx;

/* VARIANT 1 scan over ALL primitives.  The filter table has one entry per primitive in
 * scan order (spheres, then triangles): a sphere is its own bound; a triangle is bounded by
 * a sphere around its centroid (a ray that hits the triangle passes through that sphere,
 * and the sphere's centre is at most its radius behind the origin, hence the entry's
 * tca threshold -(R + tol)).  The table lives in HBM and is read with a wave-uniform index,
 * i.e. by scalar loads through the constant cache into SGPRs: no LDS traffic, no VGPRs,
 * and no size limit -- a 10k-triangle mesh streams through at 20 B per primitive. */
/* word = 2 * word + keep, keep = !(tca < neg_tol) && !(d2 > r2_hi), in three VALU instructions:
 * the two compares (NaN-aware: a NaN keeps the primitive, as it passes both reference tests),
 * and an add-with-carry that shifts the result bit in.  The compiler's own sequence for
 * `word |= keep << k` is compare, compare, move, select, or. */
__device__ __forceinline__ uint32_t push_keep_bit(uint32_t word, float tca, float neg_tol, float d2, float r2_hi)
{
  unsigned long long tmp;
  asm("v_cmp_nlt_f32 vcc, %2, %3\n\t"
               "v_cmp_ngt_f32 %1, %4, %5\n\t"
               "s_and_b64 vcc, vcc, %1\n\t"
               "v_addc_co_u32 %0, vcc, %0, %0, vcc"
               : "+v"(word), "=&s"(tmp)
               : "v"(tca), "v"(neg_tol), "v"(d2), "v"(r2_hi)
               : "vcc", "scc"); /* s_and_b64 also writes SCC */
  return word;
}

/* A per-ray value as the LOW half of a packed-fp32 operand.  The sign-test filter multiplies two spheres (the halves of
 * one register pair) by the same per-ray value; the compiler's way is to copy that value into both halves first -- eight
 * v_mov per trip -- although the hardware can read the low half for both results (op_sel_hi = 0).  The compiler does not
 * use that, so these few instructions are written out; the high half of such an operand is never read. */
__device__ __forceinline__ f32x2 lo_half(float x)
{
  f32x2 r;
  r.x = x; /* (the high half stays undefined on purpose: nothing initialises it, nothing keeps it alive) */
  return r;
}
/* a * b.lo + c.lo, a * b.lo + c, a + b.lo -- per half of a.  S0: `a` arrives in a scalar register pair (a table entry read from
 * memory through scalar loads, ConstPair below) and is used from there: one constant-bus operand per instruction is allowed */
template <bool S0 = false>
__device__ __forceinline__ f32x2 pk_fma_lo_lo(f32x2 a, f32x2 b, f32x2 c)
{
  f32x2 r;
  if (S0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "s"(a), "v"(b), "v"(c));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <bool S0 = false>
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 a, f32x2 b, f32x2 c)
{
  f32x2 r;
  if (S0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "s"(a), "v"(b), "v"(c));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <bool S0 = false>
__device__ __forceinline__ f32x2 pk_add_lo(f32x2 a, f32x2 b)
{
  f32x2 r;
  if (S0)
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "s"(a), "v"(b));
  else
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

/* The filter table where it is read from MEMORY (every kernel whose table is not staged in LDS): through the constant address
 * space, so that the wave-uniform reads become SCALAR loads (s_load: the scalar cache, values in SGPRs, no vector-memory
 * instruction).  As plain global loads the compiler could not prove the table unwritten and issued a vector load of one and the
 * same address for all 64 lanes, three per pair of spheres -- which made these kernels texture-unit bound: a CU's four SIMDs can
 * filter a pair every ~11 cycles, its one address unit took ~32 for those loads (round 4, tools/many_spheres.py: the scalar-table
 * kernels cost 2.0x the LDS-table kernel per sphere test).  The table is written by pt_build_filter in an earlier launch and is
 * immutable while a render reads it (rt_hip_shim.hip, TableSet): constant for the kernel's lifetime, which is what the
 * address space asserts. */
typedef const f32x2 __attribute__((address_space(4))) *ConstPair;
__device__ __forceinline__ ConstPair const_pairs(const f32x2 *p) { return reinterpret_cast<ConstPair>(reinterpret_cast<uintptr_t>(p)); }

/* the ray as phase 1 of scan_filtered sees it: fp32 (round to nearest: relative error <= 2^-24,
 * part of the bound), origin pulled back by filt_shift along d in the sign-test form */
struct FiltRay
{
  float ox, oy, oz;
  f32x2 dx, dy, dz;
  /* sign-test form: o'.d, |o'|^2 and -2 o' of the pulled-back origin o' (see filter_chunk) */
  float od, oo, m2ox, m2oy, m2oz;
  /* (a 32-bit member, not a bool: next to a bool the compiler took the neighbouring float to pieces, byte by byte, when it
   * copied the struct -- nine instructions of shifts and byte selects per trip to put m2oz back together) */
  uint32_t far_origin;
};

template <bool SHIFT, bool FAR32 = false>
__device__ __forceinline__ FiltRay filter_ray(const V3 &o, const V3 &d, double filt_shift, double near_R2)
{
  FiltRay r;
  /* (fused: the pulled-back origin only feeds the conservative filter, where one fp64 ulp is 2^-29 of the fp32 rounding that follows) */
  r.ox = SHIFT ? (float)__builtin_fma(-filt_shift, d.x, o.x) : (float)o.x;
  r.oy = SHIFT ? (float)__builtin_fma(-filt_shift, d.y, o.y) : (float)o.y;
  r.oz = SHIFT ? (float)__builtin_fma(-filt_shift, d.z, o.z) : (float)o.z;
  r.dx = {(float)d.x, (float)d.x};
  r.dy = {(float)d.y, (float)d.y};
  r.dz = {(float)d.z, (float)d.z};
  if (SHIFT)
  {
    r.od = __builtin_fmaf(r.oz, r.dz.x, __builtin_fmaf(r.oy, r.dy.x, r.ox * r.dx.x));
    r.oo = __builtin_fmaf(r.oz, r.oz, __builtin_fmaf(r.oy, r.oy, r.ox * r.ox));
    r.m2ox = -2.0f * r.ox;
    r.m2oy = -2.0f * r.oy;
    r.m2oz = -2.0f * r.oz;
  }
  else
    r.od = r.oo = r.m2ox = r.m2oy = r.m2oz = 0.f;
  /* "the origin is beyond near_R": outside the table's error bounds, the ray keeps every primitive.  The sign-test form has
   * |o'|^2 in fp32 already (o' = o pulled back by filt_shift ~ 1e-6 near_R: |o'|^2 and |o|^2 agree to ~3e-6 relative, fp32
   * rounding included), so it asks that instead of a second, fp64 dot product: with a margin of 1e-4 a ray it lets through
   * has |o|^2 <= near_R2 for certain; the thin shell it turns away loses only the filter's help, never a hit.  NaN: true. */
  if (SHIFT && FAR32) /* (the parked-walk kernels need the fp64 dot product for their probe anyway: they keep it) */
    r.far_origin = !(r.oo <= (float)(near_R2 * 0.9999));
  else
    r.far_origin = !(v_dot(o, o) <= near_R2); /* also true for NaN */
  return r;
}

/* Phase 1 of scan_filtered for one chunk of up to 64 primitives starting at `base` (a multiple
 * of 64): the conservative packed-fp32 filter, all lanes on the same pair.  Returns the lane's
 * keep mask (bit k = primitive base + k survives).  The ray arrives in fp32, SHIFTed where the
 * sign-test form applies (see scan_filtered); far_origin lanes keep everything. */
/* PRUNING the wall-sized spheres among themselves (sign-test kernels; the scene's LEADING pairs of spheres with radius
 * >= 1000, PtLaunch.big_pairs <= PT_BIG_PAIRS of them).  A ray inside a room of six such walls points at about half of
 * them, every one a true hit the filter must keep, and the exact test -- the kernel's largest block -- then runs for all
 * of them although only the nearest can win (config 4: 2.47 of a ray's 2.73 candidates are walls, 4.7 exact-test
 * iterations per trip where 2.9 would do).  The filter already holds, per lane and wall, tca32' and q32 ~ thc^2, and the hit
 * distance is t = tca - thc ~ t32 := tca32' - sqrt(q32).  With e = 2^-24, A = |c| + near_R + tol, W = r2_hi' - r^2 (the table's
 * widening), E = 28 e A^2 + 6 e | |c|^2 - r^2 | >= |q32 - (thc^2 + W)| (pt_build_filter; W >= E), and tol |d|^2 the pull-back of
 * the filter's origin, the same for every sphere of a ray:
 *   LOWER bound, any wall with q32 >= 0:  sqrt(q32) >= thc (1 - 2 e), so  t >= t32 - tol |d|^2 - 11.2 e A
 *     (8.2 e A the filter's bound on tca32', 3 e A one ulp of v_sqrt_f32 and the rounding of the difference);
 *   UPPER bound, a wall whose half-chord is at least r / 16 (q32 > qmin = (r / 16)^2 + W + E: the ray meets it within 86 degrees
 *     of its normal -- which also makes the hit certain: d2 <= r^2 with room to spare) and that lies ahead (t32 > tmin =
 *     2 (tol + 11.2 e A), so tca > 0 and t > EPSILON):  sqrt(thc^2 + W + E) - thc <= (W + E) / (2 thc) <= 8 (W + E) / r, so
 *     t <= t32 - tol |d|^2 + 11.2 e A + 8 (W + E) / r.
 * So with delta = 1.5 max_k (22.4 e A_k + 8 (W_k + E_k) / r_k), formed on the host (rt_hip_shim.hip, big_prune_for):
 * t32_j > t32_i + delta, for a wall i that satisfies the conditions of the upper bound and ANY wall j, means t_j > t_i by a
 * margin that dwarfs the reference's own fp64 rounding (~1e-12 A): wall j can neither be the closest hit nor tie with it,
 * and its candidate bit is cleared.  (A wall the ray starts inside has t32 < 0: never pruned, never pruning.)  Config 4:
 * delta = 0.42 in a room of 40 x 20 x 60: two walls survive together only within that distance of a room edge, or when
 * the nearer one is met at a grazing angle.  The PT_DIAG build puts every pruned wall through the exact test after the
 * scan: it must come out strictly farther than the scan's result (tests/test_gpu_diag.py). */
#define PT_BIG_PAIRS 4u
struct BigPrune
{
  const float *tab; /* LDS, 16-byte aligned: delta, tmin, qmin[2 PT_BIG_PAIRS], pad */
  uint32_t n_pairs; /* 0: off (wave-uniform) */
};

/* TABLE_MEM: the FILT_LDS form of the filter with its table in memory (pt_render_tiles_pool_mem_s); the other form's table
 * always is */
template <bool TRIS, bool FILT_LDS, bool TABLE_MEM = false>
__device__ __forceinline__ void filter_chunk(const f32x2 *__restrict__ filt, uint32_t base, uint32_t chunk, const FiltRay &fr,
                                             uint32_t &cand_lo, uint32_t &cand_hi, BigPrune big = BigPrune{nullptr, 0u},
                                             uint32_t *pruned_out = nullptr)
{
  constexpr bool SHIFT = FILT_LDS && !TRIS;
  constexpr bool MEM = TABLE_MEM || !FILT_LDS;
  const float ox = fr.ox, oy = fr.oy, oz = fr.oz;
  const f32x2 dx = fr.dx, dy = fr.dy, dz = fr.dz;
  const bool far_origin = fr.far_origin;
  /* sign-test form: per-ray terms of the expanded products, both halves alike */
  const f32x2 dxl = lo_half(fr.dx.x), dyl = lo_half(fr.dy.x), dzl = lo_half(fr.dz.x), neg_odl = lo_half(-fr.od), ool = lo_half(fr.oo),
              m2oxl = lo_half(fr.m2ox), m2oyl = lo_half(fr.m2oy), m2ozl = lo_half(fr.m2oz);
  /* ---- phase 1: conservative packed-fp32 filter, all lanes on the same pair ---- */
  cand_lo = 0;
  cand_hi = 0;
  struct PairRec
  {
    f32x2 cx, cy, cz, r2_hi, neg_tol; /* sign-test form: r2_hi holds kq = |c|^2 - r2_hi instead, neg_tol is not read */
  };
  auto load_pair = [&](uint32_t pair) -> PairRec {
    const size_t at = PT_FILT_STRIDE * (size_t)((base >> 1) + pair);
    if (MEM)
    {
      const ConstPair g = const_pairs(filt) + at;
      if (SHIFT)
        return {g[0], g[1], g[2], g[5], g[5]};
      return {g[0], g[1], g[2], g[3], g[4]};
    }
    const f32x2 *g = filt + at;
    if (SHIFT)
      return {g[0], g[1], g[2], g[5], g[5]};
    return {g[0], g[1], g[2], g[3], g[4]};
  };
  auto filter_pair = [&](const PairRec &g, uint32_t &word, uint32_t shift) {
    if (SHIFT)
    {
      /* the products expanded: tca' = c.d - o'.d and |c - o'|^2 - r2_hi = (|c|^2 - r2_hi) + |o'|^2 - 2 c.o', so the
       * per-sphere work is two 3-term chains on c alone (8 packed ops per pair instead of 10; |c|^2 - r2_hi comes
       * exact-then-rounded from the table, which also spares the walls' |L|^2 ~ 1e8 its fp32 rounding) */
      const f32x2 tca = pk_fma_lo<MEM>(g.cz, dzl, pk_fma_lo<MEM>(g.cy, dyl, pk_fma_lo_lo<MEM>(g.cx, dxl, neg_odl)));
      const f32x2 ll = pk_fma_lo<MEM>(g.cz, m2ozl, pk_fma_lo<MEM>(g.cy, m2oyl, pk_fma_lo<MEM>(g.cx, m2oxl, pk_add_lo<MEM>(g.r2_hi, ool))));
      /* ONE sign decides: q'' = tca |tca| - ll.  Where tca32 >= 0 it is q = tca^2 - ll, the reject "d2 > r2_hi" as
       * before.  Where tca32 < 0 the reference rejects the sphere whatever q says (the pulled-back origin makes
       * tca32' > 0 for every tca >= 0, scan_filtered), so any sign is right there: -tca^2 - ll is negative for an
       * origin outside the sphere (ll > 0: dropped, as the tca test did) and may come out positive for an origin
       * inside it (kept: the exact test rejects it).  Two single fmas with an |.| source modifier (packed
       * instructions have none) replace one packed fma and the two ORs of the sign words. */
      const float qx = __builtin_fmaf(tca.x, __builtin_fabsf(tca.x), -ll.x), qy = __builtin_fmaf(tca.y, __builtin_fabsf(tca.y), -ll.y);
      /* pairs arrive in DESCENDING order: shifting sign bits in leaves bit k = primitive k;
       * a set bit means DROP here, the word is inverted after the loop */
      word = __builtin_amdgcn_alignbit(word, __float_as_uint(qy), 31);
      word = __builtin_amdgcn_alignbit(word, __float_as_uint(qx), 31);
      return;
    }
    const f32x2 lx = g.cx - ox, ly = g.cy - oy, lz = g.cz - oz;
    const f32x2 tca = __builtin_elementwise_fma(lz, dz, __builtin_elementwise_fma(ly, dy, lx * dx));
    const f32x2 ll = __builtin_elementwise_fma(lz, lz, __builtin_elementwise_fma(ly, ly, lx * lx));
    const f32x2 d2 = __builtin_elementwise_fma(-tca, tca, ll);
    if (FILT_LDS)
    { /* pairs arrive in DESCENDING order, so shifting bits in leaves bit k = primitive k */
      word = push_keep_bit(word, tca.y, g.neg_tol.y, d2.y, g.r2_hi.y);
      word = push_keep_bit(word, tca.x, g.neg_tol.x, d2.x, g.r2_hi.x);
    }
    else
    {
      /* bitwise |: no short-circuit branch.  NaNs compare false and stay candidates. */
      const bool drop0 = (bool)((int)(tca.x < g.neg_tol.x) | (int)(d2.x > g.r2_hi.x));
      const bool drop1 = (bool)((int)(tca.y < g.neg_tol.y) | (int)(d2.y > g.r2_hi.y));
      word |= (drop0 ? 0u : (1u << shift)) | (drop1 ? 0u : (2u << shift));
    }
  };
  const uint32_t n_pairs = (chunk + 1u) >> 1;
  const uint32_t pairs_lo = min(n_pairs, 16u);
  if (FILT_LDS)
  {
    /* descending pair order (see filter_pair); the LDS reads of the next pair are issued
     * before the current one computes */
    /* (unrolled by hand: inline asm is convergent, which rules out runtime unrolling) */
    auto run_desc = [&](uint32_t top, uint32_t count, uint32_t &word) {
      uint32_t q = 0;
      for (; q + 2 <= count; q += 2)
      {
        const PairRec a = load_pair(top - q), b = load_pair(top - q - 1u);
        filter_pair(a, word, 0);
        filter_pair(b, word, 0);
      }
      for (; q < count; q++)
        filter_pair(load_pair(top - q), word, 0);
    };
    /* the leading wall pairs come last (descending order) and by a loop of their own, which also estimates their hit distances */
    const uint32_t nb = (SHIFT && base == 0u) ? min(big.n_pairs, pairs_lo) : 0u;
    run_desc(pairs_lo - 1u, pairs_lo - nb, cand_lo);
    uint32_t pruned = 0u;
    if (SHIFT && nb != 0u)
    {
      /* delta, tmin, then qmin per sphere: three 16-byte reads */
      const float4 c0 = *reinterpret_cast<const float4 *>(big.tab), c1 = *reinterpret_cast<const float4 *>(big.tab + 4),
                   c2 = *reinterpret_cast<const float4 *>(big.tab + 8);
      const float delta = c0.x, tmin = c0.y;
      const float qmin[2 * PT_BIG_PAIRS] = {c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y};
      const float quiet_nan = __uint_as_float(0x7FC00000u);
      /* NaN stands for "takes no part": v_min ignores it and no comparison with it holds */
      float t32[2 * PT_BIG_PAIRS];
      float m = __builtin_inff();
#pragma unroll
      for (int p = (int)PT_BIG_PAIRS - 1; p >= 0; p--)
      {
        t32[2 * p] = quiet_nan;
        t32[2 * p + 1] = quiet_nan;
        if ((uint32_t)p < nb) /* wave-uniform */
        {
          const PairRec g = load_pair((uint32_t)p);
          const f32x2 tca = pk_fma_lo<MEM>(g.cz, dzl, pk_fma_lo<MEM>(g.cy, dyl, pk_fma_lo_lo<MEM>(g.cx, dxl, neg_odl)));
          const f32x2 ll = pk_fma_lo<MEM>(g.cz, m2ozl, pk_fma_lo<MEM>(g.cy, m2oyl, pk_fma_lo<MEM>(g.cx, m2oxl, pk_add_lo<MEM>(g.r2_hi, ool))));
          const float qx = __builtin_fmaf(tca.x, __builtin_fabsf(tca.x), -ll.x), qy = __builtin_fmaf(tca.y, __builtin_fabsf(tca.y), -ll.y);
          cand_lo = __builtin_amdgcn_alignbit(cand_lo, __float_as_uint(qy), 31);
          cand_lo = __builtin_amdgcn_alignbit(cand_lo, __float_as_uint(qx), 31);
          /* the distance estimates (q'' = q32 where tca32' > 0; NaN where q'' < 0: such a wall is dropped anyway) ... */
          const float tx = tca.x - __builtin_amdgcn_sqrtf(qx), ty = tca.y - __builtin_amdgcn_sqrtf(qy);
          t32[2 * p] = tx;
          t32[2 * p + 1] = ty;
          /* ... and, of the walls that may PRUNE (a certain hit ahead with a half-chord of r / 16 at least), the nearest */
          const float px = ((qx > qmin[2 * p]) & (tx > tmin)) ? tx : quiet_nan;     /* (t32 > tmin > 0 implies tca32' > 0) */
          const float py = ((qy > qmin[2 * p + 1]) & (ty > tmin)) ? ty : quiet_nan;
          m = hw_min(m, hw_min(px, py)); /* (v_min_f32 itself: NaN-ignoring, and no canonicalising v_max x, x before it) */
        }
      }
      const float thr = m + delta;
#pragma unroll
      for (int k = 0; k < 2 * (int)PT_BIG_PAIRS; k++)
        if ((uint32_t)k < 2u * nb) /* wave-uniform */
          pruned |= (t32[k] > thr) ? (1u << k) : 0u;
      cand_lo |= pruned; /* drop bits here */
      if (far_origin)
        pruned = 0u;
    }
    if (pruned_out)
      *pruned_out = pruned;
    run_desc(n_pairs - 1u, n_pairs - pairs_lo, cand_hi);
    if (SHIFT)
    { /* drop bits -> keep bits */
      cand_lo = ~cand_lo;
      cand_hi = ~cand_hi;
    }
  }
  else
  {
    /* software pipeline: the scalar loads of pair p+1 are in flight while pair p computes
     * (the table is padded to a whole number of pairs, and one pair past the end) */
    PairRec cur = load_pair(0);
#pragma unroll 2
    for (uint32_t p = 0; p < pairs_lo; p++)
    {
      const PairRec nxt = load_pair(p + 1);
      filter_pair(cur, cand_lo, 2u * p);
      cur = nxt;
    }
#pragma unroll 2
    for (uint32_t p = 16; p < n_pairs; p++)
    {
      const PairRec nxt = load_pair(p + 1);
      filter_pair(cur, cand_hi, 2u * (p - 16u));
      cur = nxt;
    }
  }
  /* entries that exist in this chunk (an odd count leaves one padding slot) */
  const uint32_t valid_lo = chunk >= 32u ? 0xFFFFFFFFu : ((1u << chunk) - 1u);
  const uint32_t valid_hi = chunk >= 64u ? 0xFFFFFFFFu : (chunk > 32u ? ((1u << (chunk - 32u)) - 1u) : 0u);
  cand_lo = far_origin ? valid_lo : (cand_lo & valid_lo);
  cand_hi = far_origin ? valid_hi : (cand_hi & valid_hi);
}

/* Phase 1 for a PRIMARY trip of the pooled kernels (render_tiles_pooled: all 64 lanes hold fresh camera rays of one
 * 8x8 tile): only the pairs of `pair_mask` (wave-uniform; bit p = pair p of this chunk holds a primitive that some
 * camera ray of the tile can reach at all, tile_cull below) go through the packed-fp32 test; every other primitive
 * of the chunk is dropped for all lanes.  Same arithmetic and thresholds as filter_chunk, so a listed primitive gets
 * the keep bit it would get there; keep bits are placed by position instead of shifted in, because pairs are skipped. */
template <bool SHIFT, bool MEM = false>
__device__ __forceinline__ void filter_chunk_listed(const f32x2 *__restrict__ filt, uint32_t base, uint32_t chunk, uint32_t pair_mask,
                                                    const FiltRay &fr, uint32_t &cand_lo, uint32_t &cand_hi)
{
  const f32x2 dx = fr.dx, dy = fr.dy, dz = fr.dz;
  const f32x2 dxl = lo_half(fr.dx.x), dyl = lo_half(fr.dy.x), dzl = lo_half(fr.dz.x), neg_odl = lo_half(-fr.od), ool = lo_half(fr.oo),
              m2oxl = lo_half(fr.m2ox), m2oyl = lo_half(fr.m2oy), m2ozl = lo_half(fr.m2oz);
  unsigned long long keep = 0;
  uint32_t pm = (uint32_t)__builtin_amdgcn_readfirstlane((int)pair_mask);
  while (pm != 0u)
  {
    const uint32_t p = (uint32_t)__builtin_ctz(pm);
    pm &= pm - 1u;
    const size_t at = PT_FILT_STRIDE * (size_t)((base >> 1) + p);
    const f32x2 *g = filt + at;
    uint32_t two;
    if (SHIFT)
    {
      f32x2 cx, cy, cz, kq;
      if (MEM)
      { /* (p comes from the wave-uniform mask: a scalar, so these are scalar loads) */
        const ConstPair gc = const_pairs(filt) + at;
        cx = gc[0]; cy = gc[1]; cz = gc[2]; kq = gc[5];
      }
      else
      {
        cx = g[0]; cy = g[1]; cz = g[2]; kq = g[5];
      }
      const f32x2 tca = pk_fma_lo<MEM>(cz, dzl, pk_fma_lo<MEM>(cy, dyl, pk_fma_lo_lo<MEM>(cx, dxl, neg_odl)));
      const f32x2 ll = pk_fma_lo<MEM>(cz, m2ozl, pk_fma_lo<MEM>(cy, m2oyl, pk_fma_lo<MEM>(cx, m2oxl, pk_add_lo<MEM>(kq, ool))));
      /* a set sign bit of q'' = tca |tca| - ll means DROP (filter_chunk) */
      const float qx = __builtin_fmaf(tca.x, __builtin_fabsf(tca.x), -ll.x), qy = __builtin_fmaf(tca.y, __builtin_fabsf(tca.y), -ll.y);
      const uint32_t d0 = __float_as_uint(qx) >> 31, d1 = __float_as_uint(qy) >> 31;
      two = (d0 | (d1 << 1)) ^ 3u;
    }
    else
    {
      const f32x2 cx = g[0], cy = g[1], cz = g[2], r2_hi = g[3], neg_tol = g[4];
      const f32x2 lx = cx - fr.ox, ly = cy - fr.oy, lz = cz - fr.oz;
      const f32x2 tca = __builtin_elementwise_fma(lz, dz, __builtin_elementwise_fma(ly, dy, lx * dx));
      const f32x2 ll = __builtin_elementwise_fma(lz, lz, __builtin_elementwise_fma(ly, ly, lx * lx));
      const f32x2 d2 = __builtin_elementwise_fma(-tca, tca, ll);
      /* NaNs compare false and stay candidates */
      const bool drop0 = (bool)((int)(tca.x < neg_tol.x) | (int)(d2.x > r2_hi.x));
      const bool drop1 = (bool)((int)(tca.y < neg_tol.y) | (int)(d2.y > r2_hi.y));
      two = (drop0 ? 0u : 1u) | (drop1 ? 0u : 2u);
    }
    keep |= (unsigned long long)two << (2u * p);
  }
  const uint32_t valid_lo = chunk >= 32u ? 0xFFFFFFFFu : ((1u << chunk) - 1u);
  const uint32_t valid_hi = chunk >= 64u ? 0xFFFFFFFFu : (chunk > 32u ? ((1u << (chunk - 32u)) - 1u) : 0u);
  cand_lo = fr.far_origin ? valid_lo : ((uint32_t)keep & valid_lo);
  cand_hi = fr.far_origin ? valid_hi : ((uint32_t)(keep >> 32) & valid_hi);
}

/* Which primitives can a camera ray of tile (tx0, ty0) reach at all?  -> pairs[c]: bit p set = pair p of chunk c (entries
 * 64 c + 2 p, + 1) holds such a primitive.  Once per workgroup, thread = entry, before a barrier.
 *   The camera rays of the tile are d = normalize(w), w(u, v) = pos - (llc + H u + V v) (get_camera_ray :377-383) with
 *   u in [tx0, tx0 + 8] / (W - 1), v in [ty0, ty0 + 8] / (H - 1) (pixel + jitter in [0, 1), raytracer.c:203-204): w is
 *   affine in (u, v), so every direction lies in the convex cone of the four corner vectors, i.e. within the angle
 *   theta of the centre direction a that the farthest corner makes.  A ray from pos with direction within theta of a
 *   can touch the ball (c, R) only if the angle between a and c - pos is at most theta + asin(R / |c - pos|) (or pos is
 *   inside the ball).  R is the sphere's radius, or the radius of a triangle's bounding sphere (entry_src).  Everything
 *   in fp64, square roots and quotients through the hardware's ~2^-26 seeds (errors ~1e-7 relative in all), with margins
 *   of 1e-5 in the radius, in cos(theta) and in the final comparison: conservative by orders of magnitude over both that
 *   and the fp64 rounding of the exact tests that decide, which can accept nothing farther than ~1e-12 |c| outside a
 *   primitive.  Non-finite anything: keep.  (The PT_DIAG build re-checks
 *   every primitive dropped this way with the exact test, like every other dropped primitive.) */
/* (the test itself, for one ball: centre c, radius R already widened by its margin) */
__device__ __forceinline__ bool tile_cone_reaches_ball(const double *cam_lds, uint32_t tx0, uint32_t ty0, const V3 &c, double R)
{
  /* 1 / sqrt and 1 / x from the hardware's seed instructions (v_rsq_f64, v_rcp_f64: ~2^-26 relative): the margins
   * below are 1e-5, and the correctly rounded expansions of ten square roots and divisions cost several hundred
   * instructions per workgroup -- 2 % of a low-spp frame */
  auto rsq = [](double x) { return __builtin_amdgcn_rsq(x); };
  auto root = [&](double x) { return x > 0.0 ? x * rsq(x) : 0.0; };
  const V3 pos = {cam_lds[0], cam_lds[1], cam_lds[2]}, Hh = {cam_lds[3], cam_lds[4], cam_lds[5]},
           Vv = {cam_lds[6], cam_lds[7], cam_lds[8]}, llc = {cam_lds[9], cam_lds[10], cam_lds[11]};
  const double u0 = (double)tx0 * cam_lds[14], u1 = (double)(tx0 + PT_TILE) * cam_lds[14]; /* x 1 / (W - 1), 1 / (H - 1) */
  const double v0 = (double)ty0 * cam_lds[15], v1 = (double)(ty0 + PT_TILE) * cam_lds[15];
  V3 w[4];
  for (int k = 0; k < 4; k++)
  {
    const double u = (k & 1) ? u1 : u0, v = (k & 2) ? v1 : v0;
    w[k] = v_sub(pos, v_add(llc, v_add(v_scale(Hh, u), v_scale(Vv, v))));
  }
  V3 a = v_add(v_add(w[0], w[1]), v_add(w[2], w[3]));
  a = v_scale(a, rsq(v_dot(a, a)));
  double cos_t = 1.0;
  for (int k = 0; k < 4; k++)
    cos_t = fmin(cos_t, v_dot(a, w[k]) * rsq(v_dot(w[k], w[k])));
  cos_t -= 1e-5;
  const double sin_t = root(1.0 - cos_t * cos_t);
  const V3 L = v_sub(c, pos);
  const double inv_len = rsq(v_dot(L, L));
  const double sin_p = R * inv_len; /* NaN / inf: the comparisons below keep the primitive */
  if (!(sin_p < 0.99999) || !(cos_t > 0.0))
    return true; /* the camera inside (or on, or within 1e-5 of) the ball; a degenerate cone */
  const double cos_p = root(1.0 - sin_p * sin_p);
  const double cos_a = v_dot(a, L) * inv_len;
  return !(cos_a < cos_t * cos_p - sin_t * sin_p - 1e-5);
}

__device__ __forceinline__ void tile_cull(const double *cam_lds, const double *entry_src, uint32_t n_sph, uint32_t n_entries,
                                          uint32_t tx0, uint32_t ty0, uint32_t *pairs)
{
  auto root = [](double x) { return x > 0.0 ? x * __builtin_amdgcn_rsq(x) : 0.0; };
  /* thread = entry, PT_BLOCK entries per pass (one pass for the small scenes whose table is in LDS; scenes of thousands of
   * spheres -- pt_render_tiles_pool_mem_s -- take several): wave w of pass p writes the word of chunk 4 p + w */
  for (uint32_t base = 0; base < n_entries; base += PT_BLOCK)
  {
    const uint32_t i = base + threadIdx.x;
    bool keep = false;
    if (i < n_entries)
    {
      const double *e = entry_src + PT_ENTRY_SRC_STRIDE * (size_t)i; /* cx cy cz R2 |c| Rb */
      const double R = (i < n_sph ? root(e[3]) : e[5]) * (1.0 + 1e-5) + 1e-300;
      keep = tile_cone_reaches_ball(cam_lds, tx0, ty0, ld3(e), R);
    }
    unsigned long long m = __ballot(keep);
    /* entry mask -> pair mask: OR neighbouring bits, then gather the even positions */
    m = (m | (m >> 1)) & 0x5555555555555555ull;
    m = (m | (m >> 1)) & 0x3333333333333333ull;
    m = (m | (m >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    m = (m | (m >> 4)) & 0x00FF00FF00FF00FFull;
    m = (m | (m >> 8)) & 0x0000FFFF0000FFFFull;
    m = (m | (m >> 16)) & 0x00000000FFFFFFFFull;
    if ((threadIdx.x & 63u) == 0u && base + (threadIdx.x & ~63u) < n_entries)
      pairs[(base >> 6) + (threadIdx.x >> 6)] = (uint32_t)m;
  }
}

/* Per-lane fp32 pre-test of one triangle candidate (small scenes: the flat filter passes a
 * triangle through its bounding sphere, which is loose -- a ray near a cube passes the spheres of
 * most of its 12 triangles; measured on config 3: 6.9 exact tests per wave trip for 1.7 candidates
 * per ray).  Moeller-Trumbore in fp32 with every accept/reject widened by a bound on the fp32
 * error, so it can only keep extra triangles, never drop one the exact test accepts:
 *   a = e1.(d x e2), U = s.(d x e2), V = d.(s x e1), T = e2.(s x e1), s = o - v0;
 *   the exact test accepts iff |a| >= 1e-8, 0 <= U/a <= 1, V/a >= 0, (U+V)/a <= 1, T/a > 1e-8.
 * With e = 2^-24, |d| <= 1.0001, |s| <= S := near_R + |v0| (rays from farther out skip the filter
 * altogether), inputs rounded to fp32 and fused 3-term products:
 *   |a32 - a| <= 12 e |e1||e2|,  |U32 - U| <= 14 e S |e2|,  |V32 - V| <= 15 e S |e1|,
 *   |T32 - T| <= 15 e S |e1||e2|;
 * the table stores Ea = 16 e |e1||e2|, KU = 20 e S |e2|, KV = 20 e S |e1|, KT = 20 e S |e1||e2|
 * (rounded up; >= 25 % slack over the bounds, which also swallows the reference's own fp64
 * rounding, ~1e-16 of the same magnitudes).  If |a32| <= Ea the sign of a is not certain: keep.
 * Otherwise, with everything multiplied by sign(a): drop iff U < -KU, or U > |a| + Ea + KU, or
 * V < -KV, or U + V > |a| + Ea + KU + KV, or T < -KT -- each a certain violation of one of the
 * exact test's conditions.  NaNs compare false: kept.  Triangles whose products could overflow
 * fp32 get Ea = +inf in the table: always kept. */
__device__ __forceinline__ bool tri_may_hit32(const float4 &r0, const float4 &r1, const float4 &r2, float r3x, float ox,
                                              float oy, float oz, float dx, float dy, float dz);
__device__ __forceinline__ bool tri_may_hit32(const float4 *__restrict__ rec, float ox, float oy, float oz, float dx,
                                              float dy, float dz)
{
  return tri_may_hit32(rec[0], rec[1], rec[2], rec[3].x, ox, oy, oz, dx, dy, dz);
}
__device__ __forceinline__ bool tri_may_hit32(const float4 &r0, const float4 &r1, const float4 &r2, float r3x, float ox,
                                              float oy, float oz, float dx, float dy, float dz)
{
  const float v0x = r0.x, v0y = r0.y, v0z = r0.z, e1x = r0.w, e1y = r1.x, e1z = r1.y, e2x = r1.z, e2y = r1.w, e2z = r2.x;
  const float Ea = r2.y, KU = r2.z, KV = r2.w, KT = r3x;
  const float hx = __builtin_fmaf(dy, e2z, -(dz * e2y)), hy = __builtin_fmaf(dz, e2x, -(dx * e2z)),
              hz = __builtin_fmaf(dx, e2y, -(dy * e2x));
  const float a = __builtin_fmaf(e1z, hz, __builtin_fmaf(e1y, hy, e1x * hx));
  const float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
  float U = __builtin_fmaf(sz, hz, __builtin_fmaf(sy, hy, sx * hx));
  const float qx = __builtin_fmaf(sy, e1z, -(sz * e1y)), qy = __builtin_fmaf(sz, e1x, -(sx * e1z)),
              qz = __builtin_fmaf(sx, e1y, -(sy * e1x));
  float V = __builtin_fmaf(dz, qz, __builtin_fmaf(dy, qy, dx * qx));
  float T = __builtin_fmaf(e2z, qz, __builtin_fmaf(e2y, qy, e2x * qx));
  const float abs_a = fabsf(a);
  if (!(abs_a > Ea))
    return true; /* near-parallel, or NaN: the sign of a is not certain */
  const uint32_t sgn = __float_as_uint(a) & 0x80000000u;
  U = __uint_as_float(__float_as_uint(U) ^ sgn);
  V = __uint_as_float(__float_as_uint(V) ^ sgn);
  T = __uint_as_float(__float_as_uint(T) ^ sgn);
  const float lim = abs_a + Ea;
  const bool drop = (U < -KU) | (U > lim + KU) | (V < -KV) | (U + V > lim + KU + KV) | (T < -KT);
  return !drop;
}

/* Two triangles per call: the same arithmetic as tri_may_hit32, element for element (so the same decisions), with triangle
 * A in the low and B in the high half of packed-fp32 registers -- the cross and dot products, 27 of the ~40 operations of a
 * pre-test, cost one instruction for both.  (Small-mesh kernels: a lane's candidates go through two at a time.) */
__device__ __forceinline__ void tri_may_hit32_x2(const float4 *__restrict__ ra, const float4 *__restrict__ rb, f32x2 ox, f32x2 oy,
                                                 f32x2 oz, f32x2 dx, f32x2 dy, f32x2 dz, bool &may_a, bool &may_b)
{
  const float4 a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = rb[0], b1 = rb[1], b2 = rb[2];
  const float a3 = ra[3].x, b3 = rb[3].x;
  const f32x2 v0x = {a0.x, b0.x}, v0y = {a0.y, b0.y}, v0z = {a0.z, b0.z}, e1x = {a0.w, b0.w}, e1y = {a1.x, b1.x}, e1z = {a1.y, b1.y},
              e2x = {a1.z, b1.z}, e2y = {a1.w, b1.w}, e2z = {a2.x, b2.x};
  const f32x2 hx = __builtin_elementwise_fma(dy, e2z, -(dz * e2y)), hy = __builtin_elementwise_fma(dz, e2x, -(dx * e2z)),
              hz = __builtin_elementwise_fma(dx, e2y, -(dy * e2x));
  const f32x2 a = __builtin_elementwise_fma(e1z, hz, __builtin_elementwise_fma(e1y, hy, e1x * hx));
  const f32x2 sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
  const f32x2 U = __builtin_elementwise_fma(sz, hz, __builtin_elementwise_fma(sy, hy, sx * hx));
  const f32x2 qx = __builtin_elementwise_fma(sy, e1z, -(sz * e1y)), qy = __builtin_elementwise_fma(sz, e1x, -(sx * e1z)),
              qz = __builtin_elementwise_fma(sx, e1y, -(sy * e1x));
  const f32x2 V = __builtin_elementwise_fma(dz, qz, __builtin_elementwise_fma(dy, qy, dx * qx));
  const f32x2 T = __builtin_elementwise_fma(e2z, qz, __builtin_elementwise_fma(e2y, qy, e2x * qx));
  auto decide = [](float a_, float U_, float V_, float T_, float Ea, float KU, float KV, float KT) -> bool {
    const float abs_a = fabsf(a_);
    if (!(abs_a > Ea))
      return true; /* near-parallel, or NaN: the sign of a is not certain */
    const uint32_t sgn = __float_as_uint(a_) & 0x80000000u;
    U_ = __uint_as_float(__float_as_uint(U_) ^ sgn);
    V_ = __uint_as_float(__float_as_uint(V_) ^ sgn);
    T_ = __uint_as_float(__float_as_uint(T_) ^ sgn);
    const float lim = abs_a + Ea;
    const bool drop = (U_ < -KU) | (U_ > lim + KU) | (V_ < -KV) | (U_ + V_ > lim + KU + KV) | (T_ < -KT);
    return !drop;
  };
  may_a = decide(a.x, U.x, V.x, T.x, a2.y, a2.z, a2.w, a3);
  may_b = decide(a.y, U.y, V.y, T.y, b2.y, b2.z, b2.w, b3);
}

/* SPH_LDS (hierarchy kernels with parked walks): the flat filter covers the spheres only, and their part of the pair
 * table is staged in LDS and used in the sign-test form, as in the sphere-only kernels. */
template <bool TRIS, bool BVH, bool FILT_LDS, bool WALK = true, bool LAST = false, bool SPH_LDS = false, bool FILT_MEM = false>
__device__ __forceinline__ void scan_filtered(const double *geom, const double *tri_geom,
                                              const f32x2 *__restrict__ filt, double near_R2, uint32_t n_sph,
                                              uint32_t n_entries, const V3 &o, const V3 &d, double &min_t,
                                              int &best, double &bary_u, double &bary_v,
                                              unsigned long long *diag_ptr, const float *bvh_nodes = nullptr,
                                              uint32_t n_bvh_nodes = 0, const uint32_t *bvh_tri = nullptr,
                                              double filt_shift = 0.0, TriLast *last = nullptr, bool no_prune = false,
                                              const float4 *tri32 = nullptr, const uint32_t *prim_pairs = nullptr,
                                              BigPrune big = BigPrune{nullptr, 0u}, const MeshBound *mesh_bound = nullptr)
{
  /* prim_pairs (wave-uniform; pooled kernels' primary trips, FILT_LDS only): per chunk the pairs that a camera ray of
   * this tile can reach (tile_cull); nullptr: every pair */
  /* with a hierarchy the flat filter covers the spheres only */
  if (BVH)
    n_entries = n_sph;
  /* SHIFT form of the filter (small sphere-only scenes, i.e. the headline kernel): both
   * rejects become SIGN tests, so a primitive's keep bit costs two integer instructions
   * (or, funnel shift) instead of two compares, a scalar and, and an add-with-carry.
   *   - "tca < -tol": the filter's ray starts tol_max = filt_shift behind the real origin,
   *     o' = o - tol_max d.  That adds tol_max |d|^2 to every tca and leaves the distance of
   *     a centre from the ray's line, d2, where it was (to (1 - |d|^2) (2 tol tca + tol^2),
   *     ~1e-14); with tol_max = 12 e (max |c| + near_R), tca >= 0 implies
   *     tca32' >= 0.9998 tol_max - 8.2 e (A + 1.0001 tol_max) > 0: sign clear.
   *   - "d2 > r2_hi": q = tca'^2 - (|c - o'|^2 - r2_hi') = r2_hi' - d2 and the reject is q < 0.
   *   Both come from products EXPANDED around the centre (filter_chunk): c.d - o'.d and
   *   (|c|^2 - r2_hi') + |o'|^2 - 2 c.o', with the per-ray terms o'.d, |o'|^2, -2 o' formed once
   *   (filter_ray) and |c|^2 - r2_hi' in the table; the error bound that r2_hi' is widened by
   *   stands at pt_build_filter.
   * A NaN's sign is arbitrary: rays with non-finite o skip the filter (far_origin), rays with
   * non-finite d hit nothing in the exact test either, and scenes whose centres or radii are
   * outside fp32's comfortable range never use this form (pt_filter_in_lds). */
  static_assert(!SPH_LDS || (BVH && !FILT_LDS), "SPH_LDS is the sphere filter of the hierarchy kernels");
  constexpr bool SHIFT = (FILT_LDS && !TRIS) || SPH_LDS;
  const FiltRay fr = filter_ray<SHIFT, SHIFT && !SPH_LDS>(o, d, filt_shift, near_R2);
  const float ox = fr.ox, oy = fr.oy, oz = fr.oz;
  const f32x2 dx = fr.dx, dy = fr.dy, dz = fr.dz;
  const bool far_origin = fr.far_origin;

  for (uint32_t base = 0; base < n_entries; base += 64)
  {
    const uint32_t chunk = min(64u, n_entries - base);
    /* ---- phase 1: conservative packed-fp32 filter, all lanes on the same pair ---- */
    uint32_t cand_lo, cand_hi;
    uint32_t pruned = 0u; /* wall-sized spheres of this chunk that cannot be the closest hit (BigPrune): PT_DIAG re-checks them */
    if (SPH_LDS)
      filter_chunk<false, true, FILT_MEM>(filt, base, chunk, fr, cand_lo, cand_hi, big, &pruned);
    else if (FILT_LDS && prim_pairs != nullptr)
      filter_chunk_listed<SHIFT, FILT_MEM>(filt, base, chunk, prim_pairs[base >> 6], fr, cand_lo, cand_hi);
    else
      filter_chunk<TRIS, FILT_LDS, FILT_MEM>(filt, base, chunk, fr, cand_lo, cand_hi, big, &pruned);
    /* triangle candidates of this chunk: bits from entry n_sph on */
    uint32_t tri_lo = 0, tri_hi = 0;
    if (TRIS && !BVH)
    {
      const uint32_t first_tri_bit = n_sph > base ? min(n_sph - base, 64u) : 0u;
      const uint32_t m_lo = first_tri_bit >= 32u ? 0u : (0xFFFFFFFFu << first_tri_bit);
      const uint32_t m_hi = first_tri_bit >= 64u ? 0u : (first_tri_bit > 32u ? (0xFFFFFFFFu << (first_tri_bit - 32u)) : 0xFFFFFFFFu);
      tri_lo = cand_lo & m_lo;
      tri_hi = cand_hi & m_hi;
      cand_lo &= ~m_lo; /* what is left in cand_*: sphere candidates */
      cand_hi &= ~m_hi;
      if (FILT_LDS && !far_origin)
      {
        /* ---- phase 1b: per-lane fp32 pre-test of the lane's own triangle candidates (tri_may_hit32) ---- */
#ifdef PT_DIAG
        DIAG(34, (wave_max_u32((uint32_t)(__popc(tri_lo) + __popc(tri_hi))) + 1u) / 2u); /* wave-level pre-test iterations: two candidates each */
        {
          uint32_t tot = (uint32_t)(__popc(tri_lo) + __popc(tri_hi));
          for (int off = 32; off > 0; off >>= 1)
            tot += (uint32_t)__shfl_xor((int)tot, off);
          DIAG(35, tot); /* lane-level pre-tests */
        }
#endif
        /* two candidates per iteration (tri_may_hit32_x2); a lane with an odd one left tests it twice */
        unsigned long long w = ((unsigned long long)tri_hi << 32) | tri_lo, keep = 0;
        const f32x2 pox = {ox, ox}, poy = {oy, oy}, poz = {oz, oz};
        while (w != 0)
        {
          const uint32_t bit_a = (uint32_t)__builtin_ctzll(w);
          const unsigned long long rest = w & (w - 1ull);
          const uint32_t bit_b = rest != 0 ? (uint32_t)__builtin_ctzll(rest) : bit_a;
          w = rest & (rest - 1ull);
          const uint32_t ta = base + bit_a - n_sph, tb = base + bit_b - n_sph;
          bool may_a, may_b;
          tri_may_hit32_x2(tri32 + (PT_TRI32_STRIDE / 4) * (size_t)ta, tri32 + (PT_TRI32_STRIDE / 4) * (size_t)tb, pox, poy, poz, dx, dy, dz,
                           may_a, may_b);
          keep |= (may_a ? (1ull << bit_a) : 0ull) | (may_b ? (1ull << bit_b) : 0ull);
        }
        tri_lo = (uint32_t)keep;
        tri_hi = (uint32_t)(keep >> 32);
      }
    }
#ifdef PT_DIAG
    /* lane-level filter evaluations: (lane, primitive) pairs that went through the packed-fp32 test */
    DIAG(39, (unsigned long long)__popcll(__ballot(1)) *
                 ((FILT_LDS && !SPH_LDS && prim_pairs != nullptr) ? 2u * (uint32_t)__popc(prim_pairs[base >> 6]) : chunk));
    {
      /* exactness check of the filter: any primitive it dropped that the exact test accepts? */
      uint32_t violations = 0;
      for (uint32_t k = 0; k < chunk; k++)
      {
        const bool kept = k < 32 ? (((cand_lo | tri_lo | pruned) >> k) & 1u) : (((cand_hi | tri_hi) >> (k - 32u)) & 1u);
        double t_probe = 1.7976931348623157e308, pu = 0, pv = 0;
        int b_probe = -1;
        const uint32_t i = base + k;
        if (!TRIS || BVH || i < n_sph)
          exact_sphere(geom + PT_GEOM_STRIDE * i, i, o, d, t_probe, b_probe);
        else
          exact_triangle(tri_geom + 9 * (size_t)(i - n_sph), i, o, d, t_probe, b_probe, pu, pv);
        violations += (!kept && b_probe >= 0) ? 1u : 0u;
      }
      for (int off = 32; off > 0; off >>= 1)
        violations += (uint32_t)__shfl_xor((int)violations, off);
      DIAG(12, violations);
      const uint32_t mine = (uint32_t)(__popc(cand_lo) + __popc(cand_hi) + __popc(tri_lo) + __popc(tri_hi));
      DIAG(2, wave_max_u32((uint32_t)(__popc(cand_lo) + __popc(cand_hi))) + wave_max_u32((uint32_t)(__popc(tri_lo) + __popc(tri_hi)))); /* wave-level phase-2 iterations */
      DIAG(36, wave_max_u32((uint32_t)(__popc(tri_lo) + __popc(tri_hi)))); /* of them: exact triangle tests */
      uint32_t tot = mine;
      for (int off = 32; off > 0; off >>= 1)
        tot += (uint32_t)__shfl_xor((int)tot, off);
      DIAG(3, tot);                /* lane-level candidates */
      if (!TRIS)
      { /* what pruning the wall-sized spheres (r > 1000) among themselves could reach: iterations if every lane
         * kept one wall, the lanes' other candidates, the walls alone */
        uint32_t walls = 0, others = 0, lo = cand_lo, hi = cand_hi;
        while (lo | hi)
        {
          const bool in_lo = lo != 0;
          const uint32_t word = in_lo ? lo : hi;
          const uint32_t k = (uint32_t)__builtin_ctz(word) + (in_lo ? 0u : 32u);
          lo = in_lo ? (word & (word - 1u)) : 0u;
          hi = in_lo ? hi : (word & (word - 1u));
          const bool wall = geom[PT_GEOM_STRIDE * (base + k) + 3] > 1e6;
          walls += wall ? 1u : 0u;
          others += wall ? 0u : 1u;
        }
        DIAG(29, wave_max_u32(min(walls, 1u) + others));
        DIAG(30, wave_max_u32(others));
        DIAG(31, wave_max_u32(walls));
        uint32_t tw = walls, to = others;
        for (int off = 32; off > 0; off >>= 1)
        {
          tw += (uint32_t)__shfl_xor((int)tw, off);
          to += (uint32_t)__shfl_xor((int)to, off);
        }
        DIAG(32, tw);
        DIAG(33, to);
      }
    }
#endif
    PHASE(1); /* phase 1: the filter */
    /* ---- phase 2: the exact test on each lane's own candidates, in index order: spheres ... ---- */
    /* (chunks of at most 32 entries -- every configuration but the headline's 38 spheres -- have no high word: the
     * candidate loop is then a bit scan of one register, five instructions per iteration less than the two-word form) */
    if (chunk <= 32u)
      while (cand_lo)
      {
        const uint32_t k = (uint32_t)__builtin_ctz(cand_lo);
        cand_lo &= cand_lo - 1u;
        DIAG_LANES(43); /* lane-level exact sphere tests */
        exact_sphere(geom + PT_GEOM_STRIDE * (base + k), base + k, o, d, min_t, best);
      }
    while (cand_lo | cand_hi)
    {
      /* lowest set bit of the 64-bit mask, branch-free */
      const bool in_lo = cand_lo != 0;
      const uint32_t word = in_lo ? cand_lo : cand_hi;
      const uint32_t k = (uint32_t)__builtin_ctz(word) + (in_lo ? 0u : 32u);
      const uint32_t cleared = word & (word - 1u);
      cand_lo = in_lo ? cleared : 0u;
      cand_hi = in_lo ? cand_hi : cleared;
      const uint32_t i = base + k;
      DIAG_LANES(43);
      exact_sphere(geom + PT_GEOM_STRIDE * i, i, o, d, min_t, best);
    }
    PHASE(2); /* phase 2: exact tests */
#ifdef PT_DIAG
    /* a pruned wall must lose STRICTLY against what the scan found */
    for (uint32_t k = 0; k < 2u * PT_BIG_PAIRS; k++)
      if ((pruned >> k) & 1u)
      {
        double t_probe = 1.7976931348623157e308;
        int b_probe = -1;
        exact_sphere(geom + PT_GEOM_STRIDE * (base + k), base + k, o, d, t_probe, b_probe);
        if (b_probe >= 0 && !(t_probe > min_t))
          atomicAdd(&diag_ptr[4 + 12], 1ull);
        atomicAdd(&diag_ptr[4 + 37], 1ull); /* pruned walls */
      }
#endif
    /* then its triangle candidates (all of higher index than any sphere: the scan order holds).  Two
     * loops, not one with a branch inside: a wave holding both kinds would pay for both tests in
     * every iteration */
    if (TRIS && !BVH)
      while (tri_lo | tri_hi)
      {
        const bool in_lo = tri_lo != 0;
        const uint32_t word = in_lo ? tri_lo : tri_hi;
        const uint32_t k = (uint32_t)__builtin_ctz(word) + (in_lo ? 0u : 32u);
        const uint32_t cleared = word & (word - 1u);
        tri_lo = in_lo ? cleared : 0u;
        tri_hi = in_lo ? tri_hi : cleared;
        const uint32_t i = base + k;
        DIAG_LANES(41);
        exact_triangle<false, LAST, FILT_LDS>(tri_geom + 9 * (size_t)(i - n_sph), i, o, d, min_t, best, bary_u, bary_v, last); /* (FILT_LDS: no wide-range scene) */
      }
  }
  if (BVH && WALK) /* WALK = false: the caller walks the hierarchy itself, later (render_tiles_pooled) */
    bvh_traverse<LAST>(bvh_nodes, n_bvh_nodes, bvh_tri, tri_geom, n_sph, far_origin, o, d, min_t, best, bary_u, bary_v,
                       diag_ptr, last, no_prune, nullptr, tri32); /* tri32: in leaf order for hierarchy scenes */
}

#endif /* PT_FILTER_H */
