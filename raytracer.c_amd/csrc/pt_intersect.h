/* pt_intersect.h -- the exact fp64 primitives of intersect() (raytracer.c:77-174: exact_sphere, exact_triangle, in the reference's operation
 * order) and the triangle hierarchy in packed fp32 (slab test, probe, per-lane walk, leaf pre-tests).
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_INTERSECT_H
#define PT_INTERSECT_H

/* ---- the sphere scan of intersect() (raytracer.c:401-412) ---------------------------
 *
 * VARIANT 0: the literal scan -- every lane runs intersect_sphere() on sphere i, in
 *   lock step; the sqrt / t0,t1 block sits under a per-lane branch.  Correct but
 *   wasteful on a 64-wide wavefront: with incoherent rays almost every sphere is
 *   passed by SOME lane, so the wave pays for the expensive block at ~10-20 % lane
 *   occupancy (measured: VALU busy 89 %, average 41 % of lanes active).
 *
 * VARIANT 1 (shipped): filter, then compact.
 *   Phase 1, wave-uniform over spheres, two spheres per instruction: a CONSERVATIVE
 *     version of the two early rejects of intersect_sphere (tca < 0, d2 > r*r) in packed
 *     fp32 with fused multiply-adds (v_pk_fma_f32: 5 packed ops per sphere instead of 15
 *     fp64 ops).  fp32 values differ from the reference's fp64 ones by a bounded amount;
 *     the thresholds are widened by a rigorous bound on that difference (derivation at
 *     pt_build_filter), so phase 1 NEVER drops a sphere the reference accepts -- it can only
 *     let extra ones through.  It records, per lane, a bit mask of surviving spheres.
 *     Rays that start farther out than the staging assumed (|o| > near_R) skip the filter
 *     and keep every sphere.
 *   Phase 2, per lane over its own set bits: the EXACT intersect_sphere() (fp64, reference
 *     operation order, no fusion) on that lane's next candidate, sphere data gathered
 *     from LDS by index.  Lanes test different spheres in the same instruction, so the
 *     sqrt block runs at (mean / max candidates per lane) occupancy instead of (lanes
 *     passing sphere i) / 64.  Visiting candidates in increasing index order with strict <
 *     keeps the reference's first-index-wins tie rule.
 *   Exactness: every accept/reject that reaches the result is made by phase 2's exact
 *   arithmetic; phase 1 can only add work, never change an outcome (PT_DIAG builds
 *   re-check every dropped sphere with the exact test and count violations: zero).
 */
typedef float f32x2 __attribute__((ext_vector_type(2)));

/* Correctly rounded sqrt for x == 0 or x >= 2^-767: hipcc's own fp64 expansion (v_rsq_f64 +
 * two Goldschmidt steps + two residual corrections) minus its input/output scaling, which
 * only acts below 2^-767.  Same instructions on the same values => the same result as
 * sqrt(x) there.  In intersect_sphere x = r*r - d2 is zero or at least half an ulp of r*r, and
 * rt_hip_scene_create rejects radii below 1e-100, so the precondition always holds. */
/* x == 0 without a select: the seed is taken of max(x, 4.9e-324) (the integer inline constant 1 read as a double: the least
 * denormal; v_max_f64 ignores a NaN operand, and fp64 denormals are on in this mode), so it is finite where 1 / sqrt(0) would
 * be inf, and everything after it multiplies by x itself: g = 0 * y = 0, both corrections are 0, the result is x (+0 or -0),
 * as IEEE sqrt has it.  For x >= 2^-767 the maximum is x: nothing changes.  NaN still comes out NaN (g = x * y).  One
 * instruction instead of a compare and two selects in every exact sphere test, every normal and every accepted direction. */
__device__ __forceinline__ double sqrt_unscaled(double x)
{
  double xs;
  asm("v_max_f64 %0, %1, 1" : "=v"(xs) : "v"(x));
  const double y = __builtin_amdgcn_rsq(xs);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d0 = __builtin_fma(-g, g, x);
  g = __builtin_fma(d0, h, g);
  const double d1 = __builtin_fma(-g, g, x);
  g = __builtin_fma(d1, h, g);
  return g;
}

/* 1.0 / x, correctly rounded, for 2^-500 <= x <= 2^500: hipcc's fp64 division expansion
 * (v_rcp_f64, two Newton steps, quotient, residual, final fma) minus v_div_scale /
 * v_div_fmas' scaling / v_div_fixup, which only act on operands outside that range (or
 * zero / inf / NaN).  Same instructions on the same values => the same quotient.  Used where
 * the range is known: the length of an accepted rejection sample is in [2^-30, 1]. */
__device__ __forceinline__ double rcp_unscaled(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = 1.0 * r;
  const double rem = __builtin_fma(-x, q, 1.0);
  return __builtin_fma(rem, r, q);
}

/* the same double for |a|^2 in [1e-200, 1e200] through the expansions without their range scaling (sqrt_unscaled, rcp_unscaled:
 * the same instructions on the same values; start_sample normalises the camera ray this way), the library forms otherwise:
 * ~25 instructions fewer where the vector is known to be of ordinary length -- the two children of an M_REFRACTION hit */
__device__ __forceinline__ V3 v_normalize_fast(V3 a)
{
  const double aa = v_dot(a, a);
  return (aa >= 1e-200 && aa <= 1e200) ? v_scale(a, rcp_unscaled(sqrt_unscaled(aa))) : v_scale(a, 1.0 / sqrt(aa));
}

/* intersect_sphere :82-117, exact.  Updates (min_t, best) with strict <. */
__device__ __forceinline__ void exact_sphere(const double *g, uint32_t index, const V3 &o, const V3 &d,
                                             double &min_t, int &best)
{
  V3 Lv = {g[0] - o.x, g[1] - o.y, g[2] - o.z};
  double tca = v_dot(Lv, d);
  double d2 = v_dot(Lv, Lv) - tca * tca;
  double r2 = g[3];
  if (!(tca < 0) && !(d2 > r2))
  {
    double thc = sqrt_unscaled(r2 - d2);
    /* t0 <= t1 always (thc >= 0 or NaN): the reference's swap (:95-100) is dead code */
    double t0 = tca - thc, t1 = tca + thc;
    if (t0 < 0)
      t0 = t1;
    if (t0 > kEps && t0 < min_t)
    {
      min_t = t0;
      best = (int)index;
    }
  }
}

/* intersect_triangle :132-150 (Moeller-Trumbore, two-sided), exact.  g = v0, e1, e2.
 * TIE = false: candidates arrive in increasing index order, strict < keeps the first (the
 * reference's rule).  TIE = true: they arrive in hierarchy order, so an equal t from a LOWER
 * index must still win: the result is then the linear scan's, whatever the visiting order. */
/* LAST: also remember the highest-index triangle the ray passes at t > EPSILON, closest or not.
 * intersect_triangle() writes the texture coordinates into the caller's Hit on every such hit
 * (:165-166), before the scan's `local.t < min_t` test (:426), and the scan never restores them:
 * after it hit.u / hit.v belong to the LAST passing triangle in scan order (oracle/ref_harness.c
 * revives the block around the compiled primitives and shows it).  Only M_CHECKERED reads them. */
struct TriLast
{
  int idx; /* scan index (n_sph + triangle) of the last passing triangle, -1: none */
  double u, v; /* its barycentrics */
};

/* UNSCALED: 1.0 / a through rcp_unscaled -- the same quotient for 2^-500 <= |a| <= 2^500 (either sign: the scaling steps it
 * leaves out act on magnitudes only; checked on the device, test_device_math_shortcuts_are_bit_exact).  |a| >= 1e-8 here, and
 * |a| <= |e1||e2||d| < 2e30: every launch refuses near_R >= 1e15 (rt_hip_render_tiles_chunked; the static_assert next to
 * RT_NEAR_R_LIMIT in rt_hip_shim.hip does the arithmetic), and every vertex lies within near_R / 1.5 of the origin -- that
 * check, not PtSceneView.wide_range (which speaks of spheres only), is what the range rests on.  (a = NaN or inf: no hit
 * either way -- t comes out NaN or 0, never above EPSILON.)  Four instructions less per test than the general division. */
template <bool TIE = false, bool LAST = false, bool UNSCALED = false>
__device__ __forceinline__ void exact_triangle(const double *g, uint32_t index, const V3 &o, const V3 &d,
                                               double &min_t, int &best, double &bary_u, double &bary_v,
                                               TriLast *last = nullptr)
{
  V3 v0 = ld3(g), e1 = ld3(g + 3), e2 = ld3(g + 6);
  V3 h = v_cross(d, e2);
  double a = v_dot(e1, h);
  if (!(a > -kEps && a < kEps))
  {
    double f = UNSCALED ? rcp_unscaled(a) : 1.0 / a;
    V3 sv = v_sub(o, v0);
    double u = f * v_dot(sv, h);
    if (!(u < 0.0 || u > 1.0))
    {
      V3 q = v_cross(sv, e1);
      double v = f * v_dot(d, q);
      if (!(v < 0.0 || u + v > 1.0))
      {
        double t = f * v_dot(e2, q);
        if (LAST && t > kEps && (int)index > last->idx)
        {
          last->idx = (int)index;
          last->u = u;
          last->v = v;
        }
        if (t > kEps && (t < min_t || (TIE && t == min_t && (int)index < best)))
        {
          min_t = t;
          best = (int)index;
          bary_u = u;
          bary_v = v;
        }
      }
    }
  }
}

/* the ray as the hierarchy's slab tests use it: fp32, both halves of a pair alike */
struct BvhRay
{
  f32x2 ox, oy, oz, ix, iy, iz;
};

__device__ __forceinline__ BvhRay bvh_ray(const V3 &o, const V3 &d)
{
  /* v_rcp_f32: 1 ulp (IEEE division: 10 instructions each); bvh_test_children's widening covers it */
  const float ixs = __builtin_amdgcn_rcpf((float)d.x), iys = __builtin_amdgcn_rcpf((float)d.y), izs = __builtin_amdgcn_rcpf((float)d.z);
  return {{(float)o.x, (float)o.x}, {(float)o.y, (float)o.y}, {(float)o.z, (float)o.z}, {ixs, ixs}, {iys, iys}, {izs, izs}};
}

/* v_min / v_max / v_min3 / v_max3 as the hardware has them: a NaN operand is ignored (the other comes back), which is
 * what the slab test relies on (bvh_traverse).  Through the builtins the compiler first "canonicalises" every operand
 * it cannot prove quiet (v_max_f32 x, x, x): twelve extra instructions per node visit. */
__device__ __forceinline__ float hw_min(float a, float b)
{
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_max(float a, float b)
{
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_min3(float a, float b, float c)
{
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float hw_max3(float a, float b, float c)
{
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

/* A float that is not below x, for x > 0 (inf and NaN pass through): the closest hit so far as the slab tests see it.
 * RN(x) lies at most half an ulp below x, so one ulp up is above it; callers keep the value per ray and renew it when
 * min_t changes (__double2float_ru is a 15-instruction sequence on this target, and it ran in every node visit). */
__device__ __forceinline__ float float_above(double x)
{
  const float f = (float)x;
  return f < __builtin_inff() ? __uint_as_float(__float_as_uint(f) + 1u) : f;
}

/* One visit: the boxes of node `ref`'s two children against the ray (see bvh_traverse for
 * the bounds that make it conservative).  tmax = a float not below the closest hit so far (float_above). */
__device__ __forceinline__ void bvh_test_children(const float *__restrict__ nodes, uint32_t ref, const BvhRay &R,
                                                  bool far_origin, float tmax, bool &hit0, bool &hit1, float &tn0,
                                                  float &tn1, uint32_t &r0, uint32_t &r1)
{
  const float widen = 6.0f * 5.9604644775390625e-08f;
  const float4 *node = reinterpret_cast<const float4 *>(nodes + PT_BVH_NODE_WORDS * (size_t)ref);
  const float4 px = node[0], py = node[1], pz = node[2], tail = node[3];
  /* (min, max) planes of (child 0, child 1) */
  const f32x2 tx1 = (f32x2{px.x, px.y} - R.ox) * R.ix, tx2 = (f32x2{px.z, px.w} - R.ox) * R.ix;
  const f32x2 ty1 = (f32x2{py.x, py.y} - R.oy) * R.iy, ty2 = (f32x2{py.z, py.w} - R.oy) * R.iy;
  const f32x2 tz1 = (f32x2{pz.x, pz.y} - R.oz) * R.iz, tz2 = (f32x2{pz.z, pz.w} - R.oz) * R.iz;
  tn0 = hw_max3(hw_min(tx1.x, tx2.x), hw_min(ty1.x, ty2.x), hw_min(tz1.x, tz2.x));
  float tf0 = hw_min3(hw_max(tx1.x, tx2.x), hw_max(ty1.x, ty2.x), hw_max(tz1.x, tz2.x));
  tn1 = hw_max3(hw_min(tx1.y, tx2.y), hw_min(ty1.y, ty2.y), hw_min(tz1.y, tz2.y));
  float tf1 = hw_min3(hw_max(tx1.y, tx2.y), hw_max(ty1.y, ty2.y), hw_max(tz1.y, tz2.y));
  tn0 -= fabsf(tn0) * widen;
  tf0 += fabsf(tf0) * widen;
  tn1 -= fabsf(tn1) * widen;
  tf1 += fabsf(tf1) * widen;
  /* a box starting beyond the closest hit so far cannot matter */
  hit0 = far_origin || (tf0 >= tn0 && tf0 >= 0.0f && tn0 <= tmax);
  hit1 = far_origin || (tf1 >= tn1 && tf1 >= 0.0f && tn1 <= tmax);
  r0 = __float_as_uint(tail.x);
  r1 = __float_as_uint(tail.y);
}

/* The bounding sphere of all triangles as the probe sees it: the compare form of the flat filter's test for a
 * bounding entry (scan_filtered: keep unless tca < -(R + tol) or d2 > r2_hi), thresholds widened on the host for
 * the launch's near_R by the bounds of pt_build_filter (rt_hip_shim.hip, mesh_bound_for). */
struct MeshBound
{
  float cx, cy, cz, r2_hi, neg_tol;
};

/* Could the ray reach a triangle closer than min_t at all?  The root's two child boxes -- and the triangles'
 * bounding sphere: boxes are loose around anything round (the two half-boxes of a sphere-like mesh show a ray
 * about twice the silhouette of the mesh itself), and every ray let through costs a park / walk / resume cycle
 * of ~15 node visits to find nothing. */
/* SPHERE_ONLY (pt_render_tiles_tri_queued_sph, scenes whose bounding sphere is at least as tight as the root's
 * boxes, PtSceneView.mesh_round): the boxes are left to the walk's first visit, which tests them anyway. */
template <bool SPHERE_ONLY = false>
__device__ __forceinline__ bool bvh_probe(const float *__restrict__ nodes, uint32_t n_nodes, bool far_origin,
                                          const V3 &o, const V3 &d, double min_t, const MeshBound &mb, bool *in_sphere = nullptr)
{
  if (n_nodes == 0)
    return false;
  bool hit0 = true, hit1 = true;
  if (!SPHERE_ONLY)
  {
    float tn0, tn1;
    uint32_t r0, r1;
    bvh_test_children(nodes, 0u, bvh_ray(o, d), far_origin, float_above(min_t), hit0, hit1, tn0, tn1, r0, r1);
  }
  const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
  const float lx = mb.cx - (float)o.x, ly = mb.cy - (float)o.y, lz = mb.cz - (float)o.z;
  const float tca = __builtin_fmaf(lz, dz, __builtin_fmaf(ly, dy, lx * dx));
  const float ll = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
  const float d2 = __builtin_fmaf(-tca, tca, ll);
  /* NaNs compare false: kept */
  const bool inside = far_origin | (!(tca < mb.neg_tol) & !(d2 > mb.r2_hi));
  if (in_sphere)
  { /* PT_DIAG: the caller walks the ray anyway and checks that it finds nothing */
    *in_sphere = inside;
    return hit0 || hit1;
  }
  return (hit0 || hit1) && inside;
}

/* Ordered walk of the triangle hierarchy (pt_device.h: bvh_nodes).  Per lane and per visit:
 * the widened fp32 boxes of the node's TWO children against the ray by the slab test, both
 * in the same packed-fp32 instructions, made conservative --
 *   boxes were widened at launch by 4 e (near_R + |b|) (covers rounding o to fp32 and the
 *   subtraction b - o), and the slab distances are widened by 6 e |t| (covers rounding d,
 *   the reciprocal and the product; e = 2^-24) --
 * so a box that contains an exact hit closer than min_t is never skipped.  v_min/v_max
 * ignore NaN (0 * inf on an axis-parallel ray touching a slab plane), which leaves the
 * other, correct bound.  The nearer child is entered first and the other waits on a
 * per-lane stack in LDS (the tree is balanced: at most PT_BVH_STACK deep), so the first
 * leaves reached usually hold the closest hit and min_t prunes what lies behind it.  Leaves
 * run the exact fp64 triangle test; with the (t, index) tie rule the outcome does not depend
 * on the visiting order. */
/* the per-lane traversal stacks: ONE array per workgroup, whichever instantiations of
 * bvh_traverse a kernel contains (a function-local __shared__ array in the template would be
 * allocated once per instantiation) */
__device__ __forceinline__ uint32_t (*bvh_stack_lds())[PT_BLOCK]
{
  __shared__ uint32_t stack[PT_BVH_STACK][PT_BLOCK]; /* entry-major: conflict-free per wave */
  return stack;
}

__device__ __forceinline__ bool tri_may_hit32(const float4 &r0, const float4 &r1, const float4 &r2, float r3x, float ox,
                                              float oy, float oz, float dx, float dy, float dz);

/* A leaf's triangles through the per-lane fp32 pre-test (tri_may_hit32: conservative; the table tri32 is in LEAF
 * order here, in HBM behind the pair table) -> bit k set: triangle first + k needs the exact test.  A leaf holds
 * ~5 triangles of which the ray passes one or none, and the exact fp64 test costs the wave its full length while
 * any lane's triangle needs it.  The walk is bound by memory round trips as much as by instructions: triangle
 * k + 1's record is on its way while k is tested (a leaf's records are consecutive).  Rays that start beyond
 * near_R are outside the table's error bounds: every triangle is kept. */
__device__ __forceinline__ uint32_t leaf_pretest(const float4 *__restrict__ tri32, uint32_t first, uint32_t count,
                                                 bool far_origin, const BvhRay &R, const V3 &d,
                                                 unsigned long long *diag_ptr)
{
  uint32_t keep = (1u << count) - 1u;
  if (far_origin || tri32 == nullptr)
    return keep;
  const float fdx = (float)d.x, fdy = (float)d.y, fdz = (float)d.z;
  const float4 *rec = tri32 + (PT_TRI32_STRIDE / 4) * (size_t)first;
  /* two buffers used in turn, two triangles per iteration: record k + 1 is on its way while k is tested, and no record is
   * copied from a "next" to a "current" set of registers (as one buffer pair the loop spent 13 v_mov per triangle on that) */
  float4 a0 = rec[0], a1 = rec[1], a2 = rec[2];
  float a3 = rec[3].x;
  float4 b0 = a0, b1 = a1, b2 = a2;
  float b3 = a3;
  for (uint32_t k = 0; k < count; k += 2u)
  {
    DIAG(16, 1);
    DIAG_LANES(40); /* lane-level leaf pre-tests */
    const bool second = k + 1u < count;
    if (second)
    {
      b0 = rec[4];
      b1 = rec[5];
      b2 = rec[6];
      b3 = rec[7].x;
    }
    if (!tri_may_hit32(a0, a1, a2, a3, R.ox.x, R.oy.x, R.oz.x, fdx, fdy, fdz))
      keep &= ~(1u << k);
    if (second)
    {
      DIAG(16, 1);
      DIAG_LANES(40);
      if (k + 2u < count)
      {
        a0 = rec[8];
        a1 = rec[9];
        a2 = rec[10];
        a3 = rec[11].x;
      }
      if (!tri_may_hit32(b0, b1, b2, b3, R.ox.x, R.oy.x, R.oz.x, fdx, fdy, fdz))
        keep &= ~(2u << k);
    }
    rec += 2 * (PT_TRI32_STRIDE / 4);
  }
  (void)diag_ptr;
  return keep;
}

/* LAST / no_prune: scenes with M_CHECKERED materials and triangles need every triangle the ray
 * passes, not only those closer than the closest hit so far (TriLast): no pruning by min_t then. */
template <bool LAST = false, bool OWN_STACK = false>
__device__ __forceinline__ void bvh_traverse(const float *__restrict__ nodes, uint32_t n_nodes,
                                             const uint32_t *__restrict__ tri_order, const double *tri_geom,
                                             uint32_t n_sph, bool far_origin, const V3 &o, const V3 &d,
                                             double &min_t, int &best, double &bary_u, double &bary_v,
                                             unsigned long long *diag_ptr, TriLast *last = nullptr,
                                             bool no_prune = false, uint32_t (*stack)[PT_BLOCK] = nullptr,
                                             const float4 *__restrict__ tri32_leaf = nullptr)
{
  if (!OWN_STACK) /* default: the workgroup's static array (the queued kernels pass their own, sized by the tree) */
    stack = bvh_stack_lds();
  if (n_nodes == 0)
    return;
  const BvhRay R = bvh_ray(o, d);
  uint32_t sp = 0;
  uint32_t ref = 0; /* the root node */
  bool done = false;
  float tmax = (LAST && no_prune) ? 3.4028234663852886e38f : float_above(min_t); /* renewed after every leaf */
  /* "while-while": lanes first descend until each holds a leaf (or has finished), then the
   * leaves are tested together -- the exact triangle test, the expensive block, runs with all
   * the lanes that have one instead of whenever a single lane happens to reach a leaf */
  for (;;)
  {
    while (!done && !(ref & PT_BVH_LEAF_FLAG))
    {
      DIAG(13, 1);
      DIAG_LANES(15);
      bool hit0, hit1;
      float tn0, tn1;
      uint32_t r0, r1;
      bvh_test_children(nodes, ref, R, far_origin, tmax, hit0, hit1, tn0, tn1, r0, r1);
      if (hit0 && hit1)
      {
        const bool zero_first = !(tn1 < tn0);
        stack[sp][threadIdx.x] = zero_first ? r1 : r0;
        sp++;
        ref = zero_first ? r0 : r1;
      }
      else if (hit0 || hit1)
        ref = hit0 ? r0 : r1;
      else if (sp == 0)
        done = true;
      else
      {
        sp--;
        ref = stack[sp][threadIdx.x];
      }
    }
    if (done)
      break;
    const uint32_t first = (ref & ~PT_BVH_LEAF_FLAG) >> PT_BVH_COUNT_BITS, count = ref & ((1u << PT_BVH_COUNT_BITS) - 1u);
    uint32_t keep = leaf_pretest(tri32_leaf, first, count, far_origin, R, d, diag_ptr);
    while (keep != 0u)
    {
      DIAG(14, 1);
      DIAG_LANES(41); /* lane-level exact triangle tests */
      const uint32_t t = tri_order[first + (uint32_t)__builtin_ctz(keep)];
      keep &= keep - 1u;
      exact_triangle<true, LAST>(tri_geom + 9 * (size_t)t, n_sph + t, o, d, min_t, best, bary_u, bary_v, last);
    }
    if (!(LAST && no_prune))
      tmax = float_above(min_t);
    if (sp == 0)
      break;
    sp--;
    ref = stack[sp][threadIdx.x];
  }
}

#endif /* PT_INTERSECT_H */
