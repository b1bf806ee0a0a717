/* pt_kernel.hip -- the path-tracing hot path, hand-written for gfx950 (CDNA4).
 *
 * Replaces, per pixel, the loop nest of the reference's render()
 * (gue-ni/raytracer.c raytracer.c:184-222) and everything it reaches:
 * get_camera_ray :375-384, trace_path :482-554 (the recursion rewritten as an
 * iterative bounce loop carrying a throughput), intersect :393-464,
 * intersect_sphere :77-118, intersect_triangle :120-174, the RNG helpers
 * :227-253, reflect :349-352, checkered_texture :386-391, the sample mean and
 * gamma-5 tonemap :212-220.
 *
 * pt_render_tiles (shipped).  One workgroup = one 8x8 pixel tile = 4 wavefronts;
 * wavefront w owns tile rows 2w, 2w+1 (16 pixels) and their 16*spp samples as a POOL of
 * jobs.  Each lane runs a flattened state machine: one loop iteration = one
 * trace_path() call of the reference; a lane whose path ends adds its sample to the
 * pixel's accumulator and pulls the next (pixel, sample) job of the pool in the same
 * iteration slot (wave-synchronous: ballot + prefix count, no atomics), so all 64 lanes
 * stay busy until the pool is dry whatever the individual path lengths are.  The scene
 * scan -- 80 % of the work -- is split into a wave-uniform conservative filter and a
 * per-lane exact test over the survivors (scan_filtered below).  Per-pixel sums are
 * kept in LDS as 64-bit FIXED-POINT integers (power-of-two scale chosen per launch from a
 * bound on the radiance, ~2^-40 relative resolution): integer addition is associative,
 * so the image is bit-identical under any lane / tile / GPU assignment although samples
 * finish in a data-dependent order.  The tile leaves as one coalesced 768-byte float3
 * store (+192 tonemapped bytes).
 *   Three more things keep lanes from idling in divergent code (render_tiles_pooled):
 * camera samples are prepared 64 at a time by the whole wave into an LDS queue instead of by
 * whichever few lanes are idle; the direction of a diffuse hit gets four rejection rounds per
 * trip and the rare lane still without a sample carries on next trip instead of the wave
 * looping on it; and, in scenes with a triangle hierarchy, rays that can reach the mesh wait
 * until a batch of them walks it together.  None of this can change a value: a sample
 * depends only on its (seed, pixel, sample) stream.
 *
 * Kernel family: pt_render_tiles[_tri][_big][_chk] (pooled body, by scene content), pt_render_tiles_pool_mem* (the same body
 * with geometry read from memory: scenes beyond the LDS staging budget, and sphere scenes beyond ~85 spheres by preference),
 * pt_render_tiles_refr_pool (the same body for small sphere scenes with M_REFRACTION: windowed pixel sums, pending rays that
 * travel with a path), pt_render_tiles_tri_queued* (hierarchy scenes: parked walks), pt_render_tiles[..]_refr and
 * pt_whitted_tiles[..] (static body: refraction's two-child tree where the pooled kernel does not apply, and cast_ray,
 * raytracer.c:556-641), see pt_pick_kernel.
 *
 * pt_render_tiles_v0 (kept for A/B and as the plainest statement of the algorithm): static
 * assignment lane = (pixel, sample slice), literal scan, fp64 partial sums combined by
 * xor-shuffles in a fixed order.
 *
 * Numerics: everything on the decision path (hit / miss, closest index, Russian
 * roulette, rejection sampling, hemisphere flip) is fp64 in exactly the reference's
 * operation order, compiled with -ffp-contract=off, IEEE sqrt and division -- so every
 * branch decision, hence every RNG draw and the ray / test counters, equals the CPU
 * reference's bit for bit.  Only the radiance VALUE is accumulated differently (forward:
 * L += T*e; T *= albedo*cos instead of the recursive nesting; fixed-point sample sum), a
 * ~1e-12 relative difference, far inside the float32 output's rounding.
 *
 * No MFMA: branchy fp64 scalar-per-lane math with no dense contraction.  The bounding
 * roof is the fp64 VALU issue rate.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pt_device.h"
#include "rt_rng.h"

namespace
{

struct V3
{
  double x, y, z;
};

__device__ __forceinline__ V3 v_add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 v_sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 v_mul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 v_scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
/* reference vector.h:25-26: (ax*bx + ay*by) + az*bz */
__device__ __forceinline__ double v_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* reference vector.h:43-48 */
__device__ __forceinline__ V3 v_cross(V3 a, V3 b)
{
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
/* reference vector.h:53-58: v * (1.0 / sqrt(dot)) */
__device__ __forceinline__ V3 v_normalize(V3 a) { return v_scale(a, 1.0 / sqrt(v_dot(a, a))); }

__device__ __forceinline__ V3 ld3(const double *p) { return {p[0], p[1], p[2]}; }

/* The 31-bit draw as a double.  The empty asm keeps the value a 32-bit one for the compiler:
 * seeing (double)(uint32_t)(x >> 33) it otherwise converts the 64-bit shift result, i.e.
 * cvt(low half) + ldexp(cvt(high half), 32) with a high half that is always zero -- one
 * wasted fp64 add per draw. */
__device__ __forceinline__ double draw31(uint64_t &state)
{
  uint32_t r = rt_rng_next31(&state);
  asm("" : "+v"(r));
  return (double)r;
}

/* raytracer.c:227: r / 2^31, exact */
__device__ __forceinline__ double rnd(uint64_t &state) { return draw31(state) * (1.0 / 2147483648.0); }

/* random_range(-1, 1) (raytracer.c:229, :239) = rnd * (1 - -1) + -1.  rnd = r * 2^-31 and
 * the product by 2 are exact, so the only rounding is the final add: one fused
 * r * 2^-30 - 1 is the same double. */
__device__ __forceinline__ double rnd_pm1(uint64_t &state)
{
  return __builtin_fma(draw31(state), 1.0 / 1073741824.0, -1.0);
}

/* One radiance term as the fixed-point integer that goes into a pixel's sum: RN(x * scale), in two's
 * complement.  The launch picks the scale so that |x * scale| < 2^51 for every term a sample can
 * produce (rt_hip_render_tiles_chunked), so the integer can be read off the mantissa: adding
 * 1.5 * 2^52 rounds x * scale to an integer (round-to-nearest-even, as a conversion would) and leaves
 * it, offset by the constant's bit pattern, in the sum's low bits -- one fp64 add and one 64-bit
 * subtract instead of the ~10-instruction double -> int64 conversion sequence.  (NaN: any value; the
 * pixel is flagged apart.) */
__device__ __forceinline__ unsigned long long fixed_term(double x, double scale)
{
  const double magic = 6755399441055744.0; /* 1.5 * 2^52 */
  return (unsigned long long)(__double_as_longlong(__builtin_fma(x, scale, magic)) - __double_as_longlong(magic));
}

/* raytracer.c:218-220 */
__device__ __forceinline__ uint8_t tonemap(double x)
{
  double g = pow(x, 1 / 5.0);
  double lo = (g < 1) ? g : 1.0; /* MIN(x, 1): NaN -> 1 */
  double cl = (0 > lo) ? 0.0 : lo; /* MAX(0, .) */
  return (uint8_t)(255.0 * cl);
}

/* ---- math of the material code without library calls inside the trip loops -------------------------------------
 * The device library's atan2 / pow / fmod are long polynomial sequences whose dozen-odd fp64 constants the compiler
 * hoists out of the sample loop into registers -- and, in kernels at their register limit, spills from there: the
 * static-body kernels' scratch traffic at three waves per SIMD was exactly the thirteen coefficients of atan2, stored
 * once and re-loaded at every checker hit (round 4, read off the ISA).  None of the three decides anything -- they
 * shape VALUES (a texture coordinate, a fresnel weight, a specular term) -- and the device library does not round like
 * glibc anyway ("where exactness ends"), so:
 *   cube(x), pow10(x)   products instead of pow(x, 3.0) / pow(x, 10.0): within 1.5 / 4 ulp of the exact power;
 *   frac1(x)            x - trunc(x), with x's sign = fmod(x, 1.0) EXACTLY (the difference of a double and its integer part
 *                       is representable; inf -> NaN, NaN -> NaN, as fmod has it);
 *   atan2_tab(y, x)     fdlibm's atan2 / atan (Sun Microsystems' freely distributable algorithm, e_atan2.c / s_atan.c:
 *                       argument reduction at 7/16, 11/16, 19/16, 39/16, an odd polynomial of degree 23 in two
 *                       interleaved Horner chains; error below one ulp of the result) with its twenty coefficients read
 *                       from a table in LDS through an index the compiler cannot see through, so that they stay where
 *                       they are used.  Against glibc's atan2 on 2.4e7 unit normals and random arguments (numpy, the same
 *                       unfused operations): 84 % equal, 16 % one ulp off, 2.5e-7 two ulps at a binade boundary -- the
 *                       same class as the device library's own; rt_hip_selftest_math op 6 compares it on the device. */
__device__ __forceinline__ double cube(double x) { return x * x * x; }
__device__ __forceinline__ double pow10(double x)
{
  const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
  return x8 * x2;
}
__device__ __forceinline__ double frac1(double x) { return __builtin_copysign(x - trunc(x), x); } /* (a zero result takes x's sign, as fmod's) */

#define PT_ATAN_TAB 22 /* doubles: aT[0..10], atanhi[0..3], atanlo[0..3], pi, pi_lo, 1 / (2 PI) is NOT here: the reference divides */
__constant__ double kAtanTab[PT_ATAN_TAB] = {
    3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02,
    4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00,
    2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17,
    3.1415926535897931160e+00, 1.2246467991473531772e-16, 0.0};
/* once per workgroup, before a barrier */
__device__ __forceinline__ void atan_table_to_lds(double *tab)
{
  if (threadIdx.x < PT_ATAN_TAB)
    tab[threadIdx.x] = kAtanTab[threadIdx.x];
}
__device__ __forceinline__ double atan2_tab(double y, double x, const double *tab)
{
  uint32_t z0 = 0;
  asm volatile("" : "+v"(z0)); /* (opaque_zero, defined further down) */
  const double ay = fabs(y), ax = fabs(x);
  /* atan(|y| / |x|); 0 / 0 counts as 0 (atan2(+-0, +-0) = +-0 or +-pi), |x| = 0 gives inf -> pi / 2 through the last interval */
  double q = ay / ax;
  if (ay == 0.0)
    q = 0.0;
  /* s_atan.c's argument reduction, one division for all five intervals; NaN falls through to the last and stays NaN */
  int id = 3;
  double num = -1.0, den = q;
  if (q < 2.4375) { id = 2; num = q - 1.5; den = 1.0 + 1.5 * q; }
  if (q < 1.1875) { id = 1; num = q - 1.0; den = q + 1.0; }
  if (q < 0.6875) { id = 0; num = 2.0 * q - 1.0; den = 2.0 + q; }
  if (q < 0.4375) { id = -1; num = q; den = 1.0; }
  const double xr = num / den;
  const double z = xr * xr, w = z * z;
  const double s1 = z * (tab[z0 + 0] + w * (tab[z0 + 2] + w * (tab[z0 + 4] + w * (tab[z0 + 6] + w * (tab[z0 + 8] + w * tab[z0 + 10])))));
  const double s2 = w * (tab[z0 + 1] + w * (tab[z0 + 3] + w * (tab[z0 + 5] + w * (tab[z0 + 7] + w * tab[z0 + 9]))));
  const uint32_t k = (uint32_t)(id < 0 ? 0 : id);
  const double hi = tab[z0 + 11 + k], lo = tab[z0 + 15 + k];
  const double t = xr * (s1 + s2);
  const double r = id < 0 ? xr - t : hi - ((t - lo) - xr);
  /* e_atan2.c's quadrants: the sign bit of x counts -- also of -0 when y is a zero too (atan2(+-0, -0) = +-pi) --, but
   * x = +-0 with y != 0 is +-pi / 2 whatever the zero's sign */
  const bool x_neg = __double2hiint(x) < 0 && (ax != 0.0 || ay == 0.0);
  const bool y_neg = __double2hiint(y) < 0;
  const double pi = tab[z0 + 19], pi_lo = tab[z0 + 20];
  const double left = y_neg ? (r - pi_lo) - pi : pi - (r - pi_lo);
  const double right = y_neg ? -r : r;
  return x_neg ? left : right;
}

/* PT_DIAG builds (make shim-diag, tools/diag.py) count wave-level events into stats[4..];
 * the shipped build compiles every DIAG(...) away. */
#ifdef PT_DIAG
#define DIAG(slot, value)                                                                   \
  do                                                                                        \
  {                                                                                         \
    const unsigned long long m_ = __ballot(1);                                              \
    const unsigned long long v_ = (unsigned long long)(value);                              \
    if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_))                               \
      atomicAdd(&diag_ptr[4 + (slot)], v_);                                                  \
  } while (0)
#define DIAG_LANES(slot) DIAG(slot, __popcll(__ballot(1)))
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
  for (int off = 32; off > 0; off >>= 1)
    v = max(v, (uint32_t)__shfl_xor((int)v, off));
  return v;
}
#else
#define DIAG(slot, value) do { } while (0)
#define DIAG_LANES(slot) do { } while (0)
#endif

/* PT_PHASE builds (make variant NAME=phase DEFS="-DPT_PHASE"; tools/phase.py): where a wave's cycles go, phase by phase.
 * PHASE(k) charges the shader cycles since the wave's previous mark (s_memtime) to phase k; at the kernel's end the sums go
 * to stats[64 + k].  The marks cost a few instructions each (~3 % in all): a profile, not a benchmark.  Pooled kernels. */
#ifdef PT_PHASE
#define PT_PHASE_SLOTS 16
__shared__ unsigned long long pt_phase_acc[PT_BLOCK / 64][PT_PHASE_SLOTS];
__shared__ unsigned long long pt_phase_last[PT_BLOCK / 64];
#define PHASE(k)                                                                            \
  do                                                                                        \
  {                                                                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                             \
    if ((threadIdx.x & 63u) == 0u)                                                          \
    {                                                                                       \
      pt_phase_acc[threadIdx.x >> 6][k] += t_ - pt_phase_last[threadIdx.x >> 6];            \
      pt_phase_last[threadIdx.x >> 6] = t_;                                                 \
    }                                                                                       \
  } while (0)
#else
#define PHASE(k) do { } while (0)
#endif

constexpr double kEps = 1e-8;       /* raytracer.h:24 */
constexpr double kPi = 3.14159265359; /* raytracer.h:22 */

} // namespace

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
      filter_chunk<false, true>(filt, base, chunk, fr, cand_lo, cand_hi, big, &pruned);
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
#ifdef PT_MESH_BOUND /* round 4's one experiment on config 3: built, correct, NO GAIN -- not in the shipped kernel */
      if (FILT_LDS && !far_origin && mesh_bound != nullptr)
      { /* ---- the bounding sphere of ALL triangles first (the probe's own test and thresholds, bvh_probe /
         * mesh_bound_for): a triangle's filter entry is the sphere around its centroid through its farthest corner, and for
         * the right triangles of a box those reach far beyond the box (config 3's cube: 14.5 from its centre against a
         * bounding sphere of 10.4) -- a ray that misses the mesh's own ball drops every triangle candidate at once instead
         * of taking them through the pre-test.  Conservative like the probe (the PT_DIAG build re-checks every dropped
         * triangle with the exact test, as for the filter: 0 violations).  Measured (profiles/r04_c3_mesh_bound_ab.txt):
         * bounding-sphere candidates per ray 0.83 -> 0.67, but pre-test WAVE iterations per trip only 1.83 -> 1.74 -- the
         * lanes whose rays go near the cube set the wave's pace, and those pass the ball too --, 18.62 -> 18.66 ms. ---- */
        const MeshBound &mb = *mesh_bound;
        const float lx = mb.cx - ox, ly = mb.cy - oy, lz = mb.cz - oz;
        const float tca = __builtin_fmaf(lz, dz.x, __builtin_fmaf(ly, dy.x, lx * dx.x));
        const float ll = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
        const float d2 = __builtin_fmaf(-tca, tca, ll);
        if ((tca < mb.neg_tol) | (d2 > mb.r2_hi)) /* NaNs compare false: kept */
          tri_lo = tri_hi = 0u;
      }
#endif
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
  static_assert(!SPH_FILT || (GEOM_LDS && !FILT_LDS), "SPH_FILT: the sphere pairs only, for the hierarchy kernels");
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
  if (SPH_FILT)
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
 * after 2^31 pieces (the launcher keeps samples x 2^(max_depth + 2) below 2^30).  Range: 2^-64 (bits below are dropped: 5e-20
 * absolute per term) to 2^128, all a float32 pixel can hold; a term at or above that flags the pixel like a NaN.  (Six words: a
 * seventh would cost the kernel its fourth workgroup per CU.) */
#define PT_WIN_N 6
#define PT_WIN_E0 (-64)
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
/* the sum: words combined from the top (each conversion and product is exact up to 2^-53 relative of its own word: the result is
 * within a few ulps of the exact sum, which is more than the reference's own left-to-right fp64 summation guarantees) */
__device__ __forceinline__ double win_value(const unsigned long long *w)
{
  double v = 0.0;
#pragma unroll
  for (int k = PT_WIN_N - 1; k >= 0; k--)
    v += ldexp((double)(long long)w[k], PT_WIN_E0 + 32 * k);
  return v;
}

/* ids of the pending-ray stacks of the pooled refraction kernel: 128 per wave (a wave never holds more than 64 paths in its lanes
 * and 64 on its waiting list), handed out lazily -- at a path's first M_REFRACTION hit -- from a 128-bit free mask in LDS, by
 * compare-and-swap: lanes of one wave contend in lock step, one wins per round, and few ask in the same trip */
__device__ __forceinline__ uint32_t pend_id_take(unsigned long long *free_mask)
{
  for (;;)
  {
    const unsigned long long m0 = free_mask[0];
    unsigned long long *word = m0 ? &free_mask[0] : &free_mask[1];
    const unsigned long long m = m0 ? m0 : free_mask[1];
    if (m == 0ull)
      return 0xFFu; /* (cannot happen: 128 ids for at most 128 paths) */
    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
    if (atomicCAS(word, m, m & ~(1ull << bit)) == m)
      return bit + (m0 ? 0u : 64u);
  }
}
__device__ __forceinline__ void pend_id_give(unsigned long long *free_mask, uint32_t id)
{
  atomicOr(&free_mask[id >> 6], 1ull << (id & 63u));
}
/* the pooled refraction kernel's view of a path's stack: the id is taken at the FIRST push (most paths never meet an
 * M_REFRACTION surface and never ask), records are [id][entry][field], 80 contiguous bytes */
struct PoolStack
{
  double *wave_base;             /* the wave's 128 stacks in the workgroup's pool slot */
  unsigned long long *free_mask; /* LDS: the wave's free ids */
  uint32_t *id;                  /* the path's id (a register of the calling lane), 0xFF: none yet */
  int capacity;                  /* entries per stack */
  /* (min: an id of 0xFF -- "none free", which 128 ids for at most 128 paths rule out -- must not address another wave's stacks) */
  __device__ __forceinline__ double *rec(int e) const { return wave_base + ((size_t)min(*id, 127u) * (uint32_t)capacity + (uint32_t)e) * PT_PEND_FIELDS; }
  __device__ __forceinline__ void push(int e, const V3 &o, const V3 &d, const V3 &T, int depth) const
  {
    if (*id == 0xFFu)
      *id = pend_id_take(free_mask);
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

/* a slot of a pool of `per` slots per XCD with one in-use flag each (zero between launches) */
__device__ __forceinline__ uint32_t pt_pool_acquire(uint32_t *flags_base, uint32_t per)
{
  if (flags_base == nullptr || per == 0u)
    return 0xFFFFFFFFu;
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

/* ---- one trace_path() call (raytracer.c:482-554).  Returns true when the path ended; P.Ls
 * then holds the finished sample's radiance. -------------------------------------------- */
/* VARIANT: 0 literal scan / 1 filtered scan.  REFRACT: scene has M_REFRACTION materials.
 * CHECKER: scene has M_CHECKERED materials (atan2 / fmod code; its polynomial constants
 * would otherwise be hoisted into -- and spilled from -- registers of every scene).
 * TRIS: scene has triangles.  FILT_LDS: the filter table is staged in LDS (small scenes). */
/* One round of random_in_unit_sphere (raytracer.c:231-241): x, y, z drawn in that order.
 * Returns true when the round must be repeated.  Reference: while (sqrt(len2) > 1); with a
 * correctly rounded sqrt, sqrt(x) > 1  <=>  x > 1 + 2^-52 (x = 1 + 2^-52 still rounds to 1.0),
 * so the square root is taken once, after the last round (tests/test_host.py checks the
 * equivalence around the boundary). */
__device__ __forceinline__ bool rejection_round(uint64_t &rng, V3 &q, double &len2)
{
  q.x = rnd_pm1(rng);
  q.y = rnd_pm1(rng);
  q.z = rnd_pm1(rng);
  len2 = v_dot(q, q);
  return len2 > 1.0000000000000002;
}

/* random_on_hemisphere's tail (:242-253) and the cos_theta of :549 for an accepted sample q */
__device__ __forceinline__ V3 hemisphere_from_sample(const V3 &q, double len2, const V3 &n, double &weight)
{
  /* len2 is 0 or >= 2^-60 (coordinates are multiples of 2^-30): sqrt_unscaled's domain.
   * len2 == 0 needs three draws of exactly 2^30 (probability 2^-93); the reference aborts
   * there (assert in vec3_normalize, vector.h:56) */
  const double len = sqrt_unscaled(len2);
  V3 nd = v_scale(q, rcp_unscaled(len));
  /* :250-252 flip into the normal's hemisphere, :549 cos_theta = dot(flipped, n): negating a
   * vector negates its dot product exactly, so the second dot product is the first with the
   * sign the flip gave it */
  const double side = v_dot(nd, n);
  /* (the conditional negations as one sign mask XORed into the four high words: the same doubles -- negation is the
   * sign bit -- in five instructions instead of four negations and four selects) */
  const uint32_t flip = side < 0 ? 0x80000000u : 0u;
  auto signed_by = [flip](double x) { return __hiloint2double((int)((uint32_t)__double2hiint(x) ^ flip), __double2loint(x)); };
  nd = {signed_by(nd.x), signed_by(nd.y), signed_by(nd.z)};
  weight = signed_by(side);
  return nd;
}

/* The closest hit of one intersect() call, kept between the two halves of trace_step when
 * the pooled kernel postpones the walk of the triangle hierarchy. */
struct HitRec
{
  double min_t, bary_u, bary_v;
  int best;
  bool depth_ok; /* the call ran the scan at all (:487) */
  /* DEFER_DIR: a diffuse hit whose new direction is still to be sampled.  P.d holds the
   * NORMAL meanwhile, P.T lacks the factor albedo * cos; dir_slot / dir_scale say which
   * albedo (material slot, checker factor). */
  bool need_dir;
  uint32_t dir_slot; /* bit 31: the hit is on a hull facet whose stored normal points outward (PT_HULL_PLUS) */
  double dir_scale;
  /* the ray this call sends on starts on a hull facet and leaves on its outer side by more than the launch's
   * margin: it cannot meet a triangle (pt_build_hull_flags); kernels with parked walks skip the probe for it */
  bool leaving;
  TriLast last; /* kernels with M_CHECKERED code and triangles only */
};

/* MODE 0: the whole call.  MODE 1: the first half only -- depth test, the flat scan WITHOUT the
 * hierarchy walk; the result so far goes to *rec and nothing else changes.  MODE 2: the second
 * half only, from *rec (which the caller may have completed with bvh_traverse).
 * DEFER_DIR: a diffuse hit does not sample its direction here; the caller does (HitRec). */
template <int VARIANT, bool REFRACT, bool CHECKER, bool TRIS, bool FILT_LDS, int MODE = 0, bool DEFER_DIR = false,
          bool SPH_LDS = false, bool FILT_MEM = false, class STK = PendStack>
__device__ __forceinline__ bool trace_step(const SceneCtx &S, Path &P, uint32_t &n_casts,
                                           unsigned long long *diag_ptr, const STK &stack, int &stack_n,
                                           HitRec *rec = nullptr, const uint32_t *prim_pairs = nullptr)
{
  V3 add = {S.bg, S.bg, S.bg}; /* what this call contributes if the path ends here */
  bool path_ends = true;
  const V3 o = P.o, d = P.d;
  /* DEFER_DIR callers (the pooled and parked-walk kernels) flush a path's radiance to the pixel sums every trip: P.Ls is
   * zero on entry, and this call's one term -- the hit's emission whether the path goes on or dies in the roulette,
   * BACKGROUND if it found nothing or ran out of depth -- is the throughput AT ENTRY times `add`.  Formed once (below,
   * before the throughput changes) instead of accumulated into P.Ls in two places (three products, three "+ 0" the compiler may not fold, three more
   * additions and six selects per trip).  The static body (DEFER_DIR = false: M_REFRACTION beyond the pooled kernel, cast_ray) keeps the general form. */
  constexpr bool ONE_TERM = DEFER_DIR; /* (also with REFRACT -- the pooled refraction kernel: a refractive hit contributes its one emission term like any
                                        * other hit, the two children only divide the throughput between them) */
  HitRec local;
  HitRec &H = (MODE == 0 && !DEFER_DIR) ? local : *rec;

  if (MODE != 2)
  {
    H.depth_ok = P.depth <= S.max_depth;
    H.min_t = S.t_start; /* DBL_MAX */
    H.best = -1;
    H.bary_u = 0;
    H.bary_v = 0;
    H.last.idx = -1;
    H.last.u = 0;
    H.last.v = 0;
    if (H.depth_ok)
    {
      n_casts++;
      /* ---- intersect(): closest hit, strict <, index order (:393-464) ---- */
      if (VARIANT == 0)
      {
        /* the literal scan: spheres, then triangles, every lane on the same primitive */
        for (uint32_t i = 0; i < S.n_sph; i++)
          exact_sphere(S.geom + PT_GEOM_STRIDE * i, i, o, d, H.min_t, H.best);
        for (uint32_t i = 0; i < S.n_tri; i++)
          exact_triangle<false, CHECKER && TRIS>(S.tri + 9 * (size_t)i, S.n_sph + i, o, d, H.min_t, H.best, H.bary_u, H.bary_v,
                                                 &H.last);
      }
      else
        scan_filtered<TRIS, TRIS && !FILT_LDS, FILT_LDS, MODE == 0, CHECKER && TRIS, SPH_LDS, FILT_MEM>(
            S.geom, S.tri, (FILT_LDS || SPH_LDS) ? S.filt_lds : S.filt, S.near_R2, S.n_sph, S.n_sph + S.n_tri, o, d, H.min_t, H.best,
            H.bary_u, H.bary_v, diag_ptr, S.bvh_nodes, S.n_bvh_nodes, S.bvh_tri, S.filt_shift, &H.last, S.stale_uv, S.tri32, prim_pairs, S.big,
            (TRIS && FILT_LDS && !(CHECKER && TRIS)) ? &S.mesh_bound : nullptr); /* (not where hit.u / hit.v follow EVERY passing triangle: same thing,
                                                                               * a triangle the ray passes lies inside the ball -- but keep that path as it was) */
    }
    if (MODE == 1)
      return false;
  }
  const double min_t = H.min_t;
  const int best = H.best;
  if (TRIS && (MODE != 0 || DEFER_DIR))
    H.leaving = false;
  /* ONE_TERM: BACKGROUND for every lane here, the hit's emission over it inside the hit's own branch below -- both before
   * anything touches the throughput, so no second copy of it has to live to the end of the call (nine 64-bit register
   * moves per trip), and written under the branches' lane masks, so no selects either */
  if (ONE_TERM)
    P.Ls = v_scale(P.T, S.bg);

  if (H.depth_ok)
  {
    if (best >= 0)
    {
      DIAG(8, 1);
      DIAG_LANES(9);
      /* ---- the winner's hit record (:406-411 / :428-431) ---- */
      V3 p = v_add(o, v_scale(d, min_t)); /* point_at :257 */
      /* (ONE_TERM kernels: the hit point becomes the path's origin here, for every lane that hit -- a path that ends below
       * never reads it again -- so that it is formed in the origin's own registers instead of copied there at the end) */
      if (ONE_TERM)
        P.o = p;
      V3 n;
      uint32_t slot, hull = 0u;
      double tex_u = 0, tex_v = 0;
      const bool is_tri = TRIS && (uint32_t)best >= S.n_sph;
      if (!is_tri)
      {
        const double *g = S.geom + PT_GEOM_STRIDE * best;
        /* |p - c| is r to within a few ulps of the coordinates, r in [1e-100, 1e100]
         * (rt_hip_scene_create): inside the domains of sqrt_unscaled and rcp_unscaled.  (Were
         * p to round onto c exactly, IEEE gives 0 * inf and this 0 * NaN: NaN both.) */
        const V3 pc = v_sub(p, ld3(g));
        n = v_scale(pc, rcp_unscaled(sqrt_unscaled(v_dot(pc, pc))));
        slot = (uint32_t)best;
      }
      else
      {
        const uint32_t ti = (uint32_t)best - S.n_sph;
        n = ld3(S.tri_normal + 3 * (size_t)ti);
        slot = S.tri_object[ti];
        hull = slot & (PT_HULL_PLUS | PT_HULL_MINUS);
        slot &= ~(PT_HULL_PLUS | PT_HULL_MINUS);
        /* the margin's error bound (hull_margin_for) assumes the ray that found this facet travelled no farther
         * than 2 near_R (its origin is then within 1.0001 x that + the facet's size of v0): a hit from farther away
         * -- a bounce off a far point of a wall-sized sphere -- lands less precisely on the plane.  (near_R^2 is
         * in SGPRs already; this kernel has none to spare for a constant of its own.) */
        if (TRIS && (MODE != 0 || DEFER_DIR) && !(min_t * min_t <= 4.0 * S.near_R2))
          hull = 0u;
      }
      const double *m = S.mat + PT_MAT_STRIDE * slot;
      const double prob = m[0];
      V3 albedo = ld3(m + 1);
      const V3 emission = ld3(m + 4);
      const uint32_t flags = (uint32_t)__double_as_longlong(m[7]);

      add = emission; /* a path that dies in the roulette returns emission (:502) */
      if (ONE_TERM)
        P.Ls = v_mul(P.T, emission);
      /* russian roulette :497-502: the draw is always consumed */
      if (rnd(P.rng) < prob)
      {
        path_ends = false;
        double checker_scale = 1.0;
        bool dir_deferred = false;
        if (CHECKER && (flags & PT_FLAG_CHECKER))
        {
          /* hit.u / hit.v as the scan leaves them: the LAST passing triangle's if the ray passes
           * any triangle (TriLast; a winning triangle passes, so it is covered), else the closest
           * sphere's (:410-411) */
          if (!(TRIS && H.last.idx >= 0))
          {
            tex_u = atan2_tab(n.x, n.z, S.atan_tab) / (2 * kPi) + 0.5; /* :410-411 */
            tex_v = n.y * 0.5 + 0.5;
          }
          else
          {
            /* :154-167 barycentric blend of the texture coordinates */
            const double *tx = S.tri_tex + 6 * (size_t)((uint32_t)H.last.idx - S.n_sph);
            const double lu = H.last.u, lv = H.last.v;
            double w0 = 1 - lu - lv;
            tex_u = (tx[0] * w0 + tx[2] * lu) + tx[4] * lv;
            tex_v = (tx[1] * w0 + tx[3] * lu) + tx[5] * lv;
          }
          /* checkered_texture :386-391, M = 100000 (:508) */
          double on = (double)((frac1(tex_u * 100000.0) > 0.5) ^ (frac1(tex_v * 100000.0) < 0.5)); /* fmod(., 1): frac1 */
          double c = 0.3 * (1 - on) + 0.7 * on;
          albedo = v_scale(albedo, c);
          checker_scale = c;
        }
        V3 nd;
        double weight = 1.0;
        bool split = false;
        if (REFRACT && (flags & PT_FLAG_REFRACT))
        {
          /* :514-529.  fresnel = mix(pow(1 - facing, 3), 1, 0.1); refract() with the
           * CLAMP_BETWEEN quirk (raytracer.h:30: cosi == 1 always) and iot = 1:
           *   eta = 1, k = 1 - eta*eta*(1 - cosi*cosi) = 1, n' = -N,
           *   refract(I) = I*eta + n'*(eta*cosi - sqrtf(k)) = I*1 + (-N)*0      (:354-373)
           * i.e. child A goes back along the incoming ray; child B is the mirror direction.
           * Both are normalised (:523, :526).  B waits on the stack with its share kr. */
          const double facing = -v_dot(d, n);
          const double fresnel = 1 * 0.1 + cube(1 - facing) * (1 - 0.1); /* pow(x, 3): a weight, see cube() */
          const double kr = fresnel, kt = (1 - fresnel) * 1.0;
          const V3 in = v_scale(d, -1);
          const V3 nn = v_scale(n, -1);
          const double coef = 1.0 * 1.0 - (double)sqrtf(1.0f);
          const V3 refr = v_add(v_scale(in, 1.0), v_scale(nn, coef));
          nd = v_normalize_fast(refr);
          const V3 refl = v_normalize_fast(v_sub(v_scale(d, 1), v_scale(n, 2 * v_dot(v_scale(d, 1), n))));
          const V3 base = v_mul(P.T, albedo);
          if (stack_n < stack.capacity)
            stack.push(stack_n++, p, refl, v_scale(base, kr), P.depth + 1);
          if (!ONE_TERM)
            P.Ls = v_add(P.Ls, v_mul(P.T, emission));
          P.T = v_scale(base, kt);
          split = true;
        }
        else if (flags & PT_FLAG_MIRROR)
        {
          /* reflect :349-352; direction left un-normalised (:542) */
          const double dn = v_dot(d, n);
          nd = v_sub(d, v_scale(n, 2 * dn));
          if (ONE_TERM)
            P.T = v_mul(P.T, albedo); /* here, under the branch's own lane mask, instead of three products and six selects below */
          /* nd . n = -(d . n) up to rounding far below the margin */
          if (TRIS && (MODE != 0 || DEFER_DIR))
            H.leaving = (hull & PT_HULL_PLUS) ? (-dn > S.hull_margin) : ((hull & PT_HULL_MINUS) ? (dn > S.hull_margin) : false);
        }
        else
        {
          /* random_on_hemisphere :231-253 */
          if (DEFER_DIR)
          {
            H.need_dir = true;
            H.dir_slot = slot | (hull & PT_HULL_PLUS); /* the bounce goes into the stored normal's hemisphere */
            H.dir_scale = checker_scale;
            dir_deferred = true;
            nd = n; /* P.d carries the normal until the caller has the sample */
          }
          else
          {
            V3 q;
            double len2;
            int tries = 0;
            bool again;
            do
            {
              DIAG(10, 1);
              DIAG_LANES(11);
              again = rejection_round(P.rng, q, len2);
            } while (again && ++tries < 100);
            nd = hemisphere_from_sample(q, len2, n, weight);
          }
        }
        /* L = e + albedo (.) (L_next * cos)  ==>  forward form */
        if (!split)
        {
          if (!ONE_TERM)
            P.Ls = v_add(P.Ls, v_mul(P.T, emission));
          if (!ONE_TERM && !(DEFER_DIR && dir_deferred))
            P.T = v_mul(P.T, (flags & PT_FLAG_MIRROR) ? albedo : v_scale(albedo, weight));
        }
        if (!ONE_TERM)
          P.o = p;
        P.d = nd;
        P.depth++;
      }
    }
  }
  if (path_ends)
  {
    if (!ONE_TERM)
      P.Ls = v_add(P.Ls, v_mul(P.T, add));
    if (REFRACT && stack_n > 0)
    {
      /* this branch of the tree is done: resume the most recent pending child; the RNG
       * stream simply continues, as it does across the reference's two recursive calls */
      stack.pop(--stack_n, P.o, P.d, P.T, P.depth);
      path_ends = false;
    }
  }
  return path_ends;
}

/* ---- one cast_ray() call (raytracer.c:556-641), the Whitted integrator on the other side of
 * render()'s `#if 1` (:207-211).  Same contract as trace_step: returns true when the sample is
 * finished.  One fixed point light (:567-568), Phong terms in the LIGHT's colour (1,1,1)
 * times the object colour (:586-603), a shadow ray with no distance limit (:570-572:
 * intersect(.., NULL) reports any hit in front of the point), a normalised mirror child
 * (:609-615) and a "refracted" child that, with the CLAMP_BETWEEN quirk, goes straight on
 * (:617-628).  Children are weighted by scalars, so the forward form carries a scalar weight
 * in P.T; a hit with both M_REFLECTION and M_REFRACTION traces the mirror child first and
 * parks the other on the pending-ray stack.  No random draws after the camera jitter. */
template <bool TRIS, bool FILT_LDS, bool STACK>
__device__ __forceinline__ bool whitted_step(const SceneCtx &S, Path &P, uint32_t &n_casts,
                                             unsigned long long *diag_ptr, const PendStack &stack, int &stack_n)
{
  V3 add = {S.bg, S.bg, S.bg}; /* depth limit or no hit: BACKGROUND (:561-564) */
  bool path_ends = true;
  const V3 o = P.o, d = P.d;

  if (P.depth <= S.max_depth)
  {
    n_casts++;
    double min_t = S.t_start;
    int best = -1;
    double bary_u = 0, bary_v = 0;
    TriLast last = {-1, 0, 0}; /* hit.u / hit.v of the scan: the last passing triangle's (TriLast) */
    scan_filtered<TRIS, TRIS && !FILT_LDS, FILT_LDS, true, TRIS>(S.geom, S.tri, FILT_LDS ? S.filt_lds : S.filt, S.near_R2,
                                                                 S.n_sph, S.n_sph + S.n_tri, o, d, min_t, best, bary_u,
                                                                 bary_v, diag_ptr, S.bvh_nodes, S.n_bvh_nodes, S.bvh_tri,
                                                                 S.filt_shift, &last, S.stale_uv, S.tri32, nullptr, S.big);
    if (best >= 0)
    {
      const V3 p = v_add(o, v_scale(d, min_t));
      V3 n;
      uint32_t slot;
      const bool is_tri = (uint32_t)best >= S.n_sph;
      if (!is_tri)
      {
        const V3 pc = v_sub(p, ld3(S.geom + PT_GEOM_STRIDE * best));
        n = v_scale(pc, 1.0 / sqrt_unscaled(v_dot(pc, pc)));
        slot = (uint32_t)best;
      }
      else
      {
        const uint32_t ti = (uint32_t)best - S.n_sph;
        n = ld3(S.tri_normal + 3 * (size_t)ti);
        slot = S.tri_object[ti] & ~(PT_HULL_PLUS | PT_HULL_MINUS);
      }
      const uint32_t flags = (uint32_t)__double_as_longlong(S.mat[PT_MAT_STRIDE * slot + 7]);
      V3 color = ld3(S.color_raw + 3 * (size_t)slot);

      /* shadow ray :570-572 */
      const V3 light_pos = {2, 7, 2};
      const V3 ldir = v_normalize(v_sub(light_pos, p));
      n_casts++;
      double shadow_t = S.t_start, su = 0, sv = 0;
      int blocker = -1;
      scan_filtered<TRIS, TRIS && !FILT_LDS, FILT_LDS>(S.geom, S.tri, FILT_LDS ? S.filt_lds : S.filt, S.near_R2, S.n_sph,
                                                       S.n_sph + S.n_tri, p, ldir, shadow_t, blocker, su, sv, diag_ptr,
                                                       S.bvh_nodes, S.n_bvh_nodes, S.bvh_tri, S.filt_shift, nullptr, false,
                                                       S.tri32, nullptr, S.big); /* (a shadow ray asks "any hit?": pruning walls that cannot be the
                                                                                  * CLOSEST hit never removes the closest one, so a hit stays a hit) */
      const double lit = blocker >= 0 ? 0.0 : 1.0;

      if (flags & PT_FLAG_CHECKER)
      {
        double tex_u, tex_v;
        if (!(TRIS && last.idx >= 0))
        {
          tex_u = atan2_tab(n.x, n.z, S.atan_tab) / (2 * kPi) + 0.5; /* :410-411 */
          tex_v = n.y * 0.5 + 0.5;
        }
        else
        {
          const double *tx = S.tri_tex + 6 * (size_t)((uint32_t)last.idx - S.n_sph);
          const double w0 = 1 - last.u - last.v;
          tex_u = (tx[0] * w0 + tx[2] * last.u) + tx[4] * last.v;
          tex_v = (tx[1] * w0 + tx[3] * last.u) + tx[5] * last.v;
        }
        /* checkered_texture :386-391 with M = 10 (:583) */
        const double on = (double)((frac1(tex_u * 10.0) > 0.5) ^ (frac1(tex_v * 10.0) < 0.5)); /* fmod(., 1): frac1 */
        color = v_scale(color, 0.3 * (1 - on) + 0.7 * on);
      }

      /* :586-603; light_color = (1,1,1), so each term is the same scalar in all channels */
      const double ka = 0.25, kd = 0.5, ks = 0.8, alpha = 10.0;
      const double n_dot_l = v_dot(n, ldir);
      const double diffuse = 1.0 * (kd * (0.0 > n_dot_l ? 0.0 : n_dot_l)); /* MAX(0.0, x) */
      const V3 reflected = v_sub(ldir, v_scale(n, 2 * v_dot(ldir, n)));
      const V3 view = v_normalize(v_sub(p, o));
      const double v_dot_r = v_dot(view, reflected);
      (void)alpha;
      const double specular = 1.0 * (ks * pow10(v_dot_r > 0.0 ? v_dot_r : 0.0)); /* pow(MAX(x, 0.0), alpha = 10): pow10() */
      const double shade = 1.0 * ka + (specular + diffuse) * lit;
      const V3 surface = v_scale(color, shade);
      add = surface;

      const bool mirror = (flags & PT_FLAG_MIRROR) != 0, glass = (flags & PT_FLAG_REFRACT) != 0;
      if (mirror || glass)
      {
        double kr = 1.0, kt = 0.0;
        V3 through = d;
        if (glass)
        {
          const double facing = -v_dot(d, n);
          const double fresnel = 1 * 0.1 + cube(1 - facing) * (1 - 0.1); /* mix() :255; pow(x, 3): cube() */
          kr = fresnel; /* :622 -- also the weight of an M_REFLECTION child of the same hit */
          kt = (1 - fresnel) * 0.5;
          /* refract(I, N, 1.0) :354-373 with cosi == 1: I*1 + (-N)*(1*1 - sqrtf(1)) */
          const double coef = 1.0 * 1.0 - (double)sqrtf(1.0f);
          through = v_normalize(v_add(v_scale(d, 1.0), v_scale(v_scale(n, -1), coef)));
        }
        P.Ls = v_add(P.Ls, v_mul(P.T, surface));
        const V3 weight = P.T;
        if (mirror)
        {
          const V3 refl = v_normalize(v_sub(d, v_scale(n, 2 * v_dot(d, n))));
          /* STACK = false: the launcher has checked that no material carries both flags */
          if (STACK && glass && stack_n < stack.capacity)
            stack.push(stack_n++, p, through, v_scale(weight, kt), P.depth + 1);
          P.d = refl;
          P.T = v_scale(weight, kr);
        }
        else
        {
          P.d = through;
          P.T = v_scale(weight, kt);
        }
        P.o = p;
        P.depth++;
        path_ends = false;
      }
    }
  }
  if (path_ends)
  {
    P.Ls = v_add(P.Ls, v_mul(P.T, add));
    if (STACK && stack_n > 0)
    {
      stack.pop(--stack_n, P.o, P.d, P.T, P.depth);
      path_ends = false;
    }
  }
  return path_ends;
}

/* ---- epilogue shared by both kernels: coalesced tile store + counters ------------------- */

/* per-pixel mean (raytracer.c:215) and gamma-5 tonemap (:218-220) of one tile from its
 * fixed-point sums; thread 3 t + c handles channel c of pixel t */
/* nan_mask[c]: bit t set = channel c of pixel t received a NaN sample.  The reference's fp64 sum
 * carries a NaN to the pixel (raytracer.c:212-215) and CLAMP(NaN) = 1 stores byte 255 (:218); an
 * integer sum cannot, so the pooled kernels flag such samples apart and the pixel becomes NaN
 * here.  (Samples are otherwise finite and within the scale's bound: emission is finite and the
 * throughput at most 1, rt_hip_render_tiles_chunked.) */
__device__ __forceinline__ void finish_pixels(const PtLaunch &L, const unsigned long long *sums,
                                              const unsigned long long *nan_mask, uint32_t tile, float *out_f,
                                              uint8_t *out_b)
{
  /* thread = (pixel, channel): 192 threads, one pow each (a pixel per thread kept three waves waiting on the first) */
  if (threadIdx.x < PT_TILE_PIXELS * 3)
  {
    const uint32_t t = threadIdx.x / 3u, c = threadIdx.x - 3u * t;
    const bool inside = (tile % L.tiles_x) * PT_TILE + (t & 7u) < (uint32_t)L.width &&
                        (tile / L.tiles_x) * PT_TILE + (t >> 3) < (uint32_t)L.height;
    const double inv_s = 1.0 / (double)L.samples;
    double mean = ((double)(long long)sums[threadIdx.x] * L.acc_inv_scale) * inv_s;
    const double quiet_nan = __longlong_as_double(0x7FF8000000000000ll);
    mean = ((nan_mask[c] >> t) & 1ull) ? quiet_nan : mean;
    out_f[threadIdx.x] = inside ? (float)mean : 0.f;
    out_b[threadIdx.x] = inside ? tonemap(mean) : 0;
  }
}

/* slot = index of the tile in the compact output; with_pixels = false when this workgroup
 * only contributed a sample chunk (pt_resolve_tiles writes the pixels then) */
__device__ __forceinline__ void store_tile(const PtLaunch &L, const float *out_f, const uint8_t *out_b,
                                           const unsigned long long *wg_stats, uint32_t tile, uint32_t slot,
                                           uint32_t n_prims, bool with_pixels, bool count_samples)
{
  /* 192 floats = 768 contiguous bytes per tile */
  if (with_pixels && threadIdx.x < PT_TILE_PIXELS * 3)
    L.tiles_rgb[(size_t)slot * (PT_TILE_PIXELS * 3) + threadIdx.x] = out_f[threadIdx.x];
  if (with_pixels && L.tiles_rgb8 && threadIdx.x < PT_TILE_PIXELS * 3 / 4)
    reinterpret_cast<uint32_t *>(L.tiles_rgb8)[(size_t)slot * (PT_TILE_PIXELS * 3 / 4) + threadIdx.x] =
        reinterpret_cast<const uint32_t *>(out_b)[threadIdx.x];
  if (L.stats && threadIdx.x == 0)
  {
    const unsigned long long rays = wg_stats[0], casts = wg_stats[1];
    atomicAdd(&L.stats[0], rays);
    atomicAdd(&L.stats[1], casts);
    atomicAdd(&L.stats[2], casts * (unsigned long long)n_prims);
  }
  if (L.stats && count_samples && threadIdx.x == 64)
  {
    const uint32_t tx0 = (tile % L.tiles_x) * PT_TILE, ty0 = (tile / L.tiles_x) * PT_TILE;
    const uint32_t cw = min((uint32_t)PT_TILE, (uint32_t)L.width - tx0);
    const uint32_t ch = min((uint32_t)PT_TILE, (uint32_t)L.height - ty0);
    atomicAdd(&L.stats[3], (unsigned long long)cw * ch * (unsigned long long)L.samples);
  }
}

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
      pend_slot_lds = pt_pool_acquire(L.pend_flags, L.pend_slots_per_xcd);
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
   * pool, never seen): the tile comes out NaN, as in the static body */
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
    /* thread = (pixel, channel), as finish_pixels: the windowed sum -> mean -> float + tonemapped byte */
    if (threadIdx.x < PT_TILE_PIXELS * 3)
    {
      const uint32_t t = threadIdx.x / 3u, c = threadIdx.x - 3u * t;
      const bool inside = (tile % L.tiles_x) * PT_TILE + (t & 7u) < (uint32_t)L.width && (tile / L.tiles_x) * PT_TILE + (t >> 3) < (uint32_t)L.height;
      double mean = win_value(&pix_win[threadIdx.x * PT_WIN_N]) * (1.0 / (double)L.samples);
      mean = ((pix_nan[c] >> t) & 1ull) ? __longlong_as_double(0x7FF8000000000000ll) : mean;
      out_f[threadIdx.x] = inside ? (float)mean : 0.f;
      out_b[threadIdx.x] = inside ? tonemap(mean) : 0;
    }
    __syncthreads();
    store_tile(L, out_f, out_b, wg_stats, tile, slot, S.n_sph + S.n_tri, true, true);
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

/* ---- hierarchy scenes: pooled samples + PARKED walks (pt_render_tiles_tri_big[_chk]) ---------
 *
 * Scenes with a triangle hierarchy (more than PT_FILT_LDS_MAX primitives).  The pooled body above
 * makes a ray that can reach the mesh WAIT in its lane until enough lanes wait, then walks the
 * hierarchy with the waiting lanes only: the lanes in between idle (loop occupancy 55 % on config
 * 5) and a walk batch holds ~32 rays whose lengths range from 2 to 25 visits (12 % of the lanes
 * busy inside walks; profiles/r02a_c5_pmc.txt: 30 % VALU lane utilisation overall).  Here such a
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
static_assert(PT_PARK_WAVE_BYTES >= PT_PARK_Q * 128u + PT_TILE_PIXELS * 8u && PT_PARK_F64_FIELDS * 8u + PT_PARK_U32_FIELDS * 4u <= 128u, "ring bytes per wave");
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
#ifdef PT_PARK_ONE_RECORD /* round 2's layout, for A/B */
__device__ __forceinline__ uint32_t ring_fi(uint32_t field, uint32_t e) { return e * 16u + field; }
__device__ __forceinline__ uint32_t ring_ui(uint32_t field, uint32_t e) { return e * 32u + 2u * PT_PARK_F64_FIELDS + field; }
#else
__device__ __forceinline__ uint32_t ring_fi(uint32_t field, uint32_t e)
{ /* fields: 0-2 o, 3-5 d, 6-8 T, 9 rng, 10 min_t, 11-12 last u, v */
  return field < 6u ? e * 8u + field
                    : (field == 10u ? e * 8u + 6u
                                    : (field < 10u ? PT_PARK_Q * 8u + e * 4u + (field - 6u) : PT_PARK_Q * 12u + e * 4u + (field - 11u)));
}
__device__ __forceinline__ uint32_t ring_ui(uint32_t field, uint32_t e)
{ /* fields: 0 best, 1 depth << 6 | pixel slot (+ PT_DIAG flags), 2 last index */
  return field < 2u ? e * 16u + 14u + field : PT_PARK_Q * 24u + e * 8u + 4u;
}
#endif

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
 * pt_launch_render takes the lane-waiting _tri_big kernels), or -- a sizing bug of the pool, never seen: it has
 * PT_PARK_SLOTS_PER_XCD = 192 slots for at most 160 resident workgroups -- every slot of this XCD taken after a bounded
 * search.  The waves of such a workgroup render nothing and say so: every pixel of their tiles comes out NaN (bytes 255;
 * render_tiles_queued), rather than spin for ever or walk rays from registers the kernel does not have.  Thread 0 only. */
__device__ __forceinline__ uint32_t lane_of_thread() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t pt_park_acquire(const PtLaunch &L)
{
  return pt_pool_acquire(L.park_ws == nullptr ? nullptr : L.park_flags, L.park_slots_per_xcd);
}

/* The per-lane traversal stacks of the parked-walk kernels: 24-bit entries (a 16-bit and an 8-bit array, entry-major,
 * one entry per tree level and lane), because at four workgroups per CU every kilobyte of LDS counts there.  A reference
 * fits 24 bits while node indices stay below 2^23 and meshes below 2^(23 - PT_BVH_COUNT_BITS) triangles
 * (checked on the host, pt_pick_kernel: other meshes take the lane-waiting kernels). */
struct WalkStack
{
  uint16_t *lo; /* [levels][PT_BLOCK] */
  uint8_t *hi;  /* [levels][PT_BLOCK] */
#ifdef PT_BVH_WIDE
  /* the four-wide walk can hold three entries per level: those beyond the LDS array's `cap` levels (rare) go to an
   * overflow area behind the wave's ring in the workspace, [entry][lane], read and written by the owning lane only */
  uint32_t cap;
  uint32_t *ovf;
#endif
};
#define PT_WALK_LEAF_FLAG24 0x800000u
__device__ __forceinline__ uint32_t walk_ref24(uint32_t ref) /* PT_BVH_LEAF_FLAG (bit 31) moves to bit 23 */
{
  return (ref & 0x7FFFFFu) | ((ref >> 8) & PT_WALK_LEAF_FLAG24);
}
__device__ __forceinline__ uint32_t walk_ref32(uint32_t r24) { return (r24 & 0x7FFFFFu) | ((r24 & PT_WALK_LEAF_FLAG24) << 8); }
__device__ __forceinline__ void walk_push(const WalkStack &st, uint32_t sp, uint32_t ref)
{
#ifdef PT_BVH_WIDE
  if (sp >= st.cap)
  { /* (L1-bypassing both ways, like every access to the workspace) */
    __hip_atomic_store(st.ovf + (size_t)min(sp - st.cap, 31u) * 64u + (threadIdx.x & 63u), ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
#endif
  const uint32_t r = walk_ref24(ref);
  st.lo[sp * PT_BLOCK + threadIdx.x] = (uint16_t)r;
  st.hi[sp * PT_BLOCK + threadIdx.x] = (uint8_t)(r >> 16);
}
__device__ __forceinline__ uint32_t walk_pop(const WalkStack &st, uint32_t sp)
{
#ifdef PT_BVH_WIDE
  if (sp >= st.cap)
    return __hip_atomic_load(st.ovf + (size_t)min(sp - st.cap, 31u) * 64u + (threadIdx.x & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  return walk_ref32((uint32_t)st.lo[sp * PT_BLOCK + threadIdx.x] | ((uint32_t)st.hi[sp * PT_BLOCK + threadIdx.x] << 16));
}

#ifdef PT_BVH_WIDE
/* One visit of the four-wide walk: the boxes of node `ref`'s (up to) four children against the ray -- the binary visit's
 * slab test, bounds and NaN rules (bvh_test_children), two children per packed instruction.  Leaves the nearest hit child in
 * `ref` and pushes the others; -> false when no child is hit (the caller pops or finishes). */
__device__ __forceinline__ bool bvhw_visit(const float *__restrict__ wnodes, uint32_t &ref, const BvhRay &R, bool far_origin, float tmax,
                                           const WalkStack &stack, uint32_t &sp)
{
  const float widen = 6.0f * 5.9604644775390625e-08f;
  const float4 *node = reinterpret_cast<const float4 *>(wnodes + PT_BVHW_NODE_WORDS * (size_t)ref);
  const float4 xl = node[0], xh = node[1], yl = node[2], yh = node[3], zl = node[4], zh = node[5], rr = node[6];
  /* (lo, hi) planes of children (0, 1) and (2, 3) */
  const f32x2 ax1 = (f32x2{xl.x, xl.y} - R.ox) * R.ix, ax2 = (f32x2{xh.x, xh.y} - R.ox) * R.ix;
  const f32x2 bx1 = (f32x2{xl.z, xl.w} - R.ox) * R.ix, bx2 = (f32x2{xh.z, xh.w} - R.ox) * R.ix;
  const f32x2 ay1 = (f32x2{yl.x, yl.y} - R.oy) * R.iy, ay2 = (f32x2{yh.x, yh.y} - R.oy) * R.iy;
  const f32x2 by1 = (f32x2{yl.z, yl.w} - R.oy) * R.iy, by2 = (f32x2{yh.z, yh.w} - R.oy) * R.iy;
  const f32x2 az1 = (f32x2{zl.x, zl.y} - R.oz) * R.iz, az2 = (f32x2{zh.x, zh.y} - R.oz) * R.iz;
  const f32x2 bz1 = (f32x2{zl.z, zl.w} - R.oz) * R.iz, bz2 = (f32x2{zh.z, zh.w} - R.oz) * R.iz;
  float tn[4], tf[4];
  tn[0] = hw_max3(hw_min(ax1.x, ax2.x), hw_min(ay1.x, ay2.x), hw_min(az1.x, az2.x));
  tf[0] = hw_min3(hw_max(ax1.x, ax2.x), hw_max(ay1.x, ay2.x), hw_max(az1.x, az2.x));
  tn[1] = hw_max3(hw_min(ax1.y, ax2.y), hw_min(ay1.y, ay2.y), hw_min(az1.y, az2.y));
  tf[1] = hw_min3(hw_max(ax1.y, ax2.y), hw_max(ay1.y, ay2.y), hw_max(az1.y, az2.y));
  tn[2] = hw_max3(hw_min(bx1.x, bx2.x), hw_min(by1.x, by2.x), hw_min(bz1.x, bz2.x));
  tf[2] = hw_min3(hw_max(bx1.x, bx2.x), hw_max(by1.x, by2.x), hw_max(bz1.x, bz2.x));
  tn[3] = hw_max3(hw_min(bx1.y, bx2.y), hw_min(by1.y, by2.y), hw_min(bz1.y, bz2.y));
  tf[3] = hw_min3(hw_max(bx1.y, bx2.y), hw_max(by1.y, by2.y), hw_max(bz1.y, bz2.y));
  const uint32_t r[4] = {__float_as_uint(rr.x), __float_as_uint(rr.y), __float_as_uint(rr.z), __float_as_uint(rr.w)};
  bool hit[4];
  float near_d = __builtin_inff();
  int near_c = -1;
#pragma unroll
  for (int c = 0; c < 4; c++)
  {
    const float n_ = tn[c] - fabsf(tn[c]) * widen, f_ = tf[c] + fabsf(tf[c]) * widen;
    /* (a missing child has an inverted box; far origins keep every REAL child) */
    hit[c] = r[c] != PT_BVHW_EMPTY && (far_origin || (f_ >= n_ && f_ >= 0.0f && n_ <= tmax));
    if (hit[c] && (near_c < 0 || n_ < near_d))
    {
      near_d = n_;
      near_c = c;
    }
  }
  if (near_c < 0)
    return false;
#pragma unroll
  for (int c = 3; c >= 0; c--) /* (the others wait, in reverse child order) */
    if (hit[c] && c != near_c)
    {
      walk_push(stack, sp, r[c]);
      sp++;
    }
  ref = r[near_c];
  return true;
}
#endif

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
#ifdef PT_BVH_WIDE
        if (!bvhw_visit(S.bvh_nodes + pt_bvhw_offset_words(S.n_bvh_nodes), ref, R, far_origin, wtmax, stack, sp))
        {
          if (sp == 0)
            finished = true;
          else
          {
            sp--;
            ref = walk_pop(stack, sp);
          }
        }
#else
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
#endif
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

template <bool CHECKER, bool SPHERE_PROBE = false>
__device__ __forceinline__ void render_tiles_queued(const PtLaunch &L)
{
  constexpr bool TRIS = true, FILT_LDS = false;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  /* ONE WAVE = ONE TILE here (a workgroup = four tiles, its waves independent of each other between the
   * barrier after staging and the one before the slot goes back): a wave's pool is its tile's 64 pixels
   * x samples, four times the 16-pixel strips of round 2's pooled body, so the tail in which the last paths of a
   * pool run on with most lanes idle -- each walk costs a park / walk / resume cycle, so the tail is long in
   * these kernels -- weighs a quarter as much.  (The image is 4K-sized or the scene's cost per ray is high
   * wherever these kernels run, so a quarter as many workgroups still fill the chip many times over.) */
  __shared__ unsigned long long pix_sum_all[PT_BLOCK / 64][PT_TILE_PIXELS * 3];
  __shared__ unsigned long long pix_nan_all[PT_BLOCK / 64][3];
  __shared__ uint32_t park_slot_lds;
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

#ifdef PT_PHASE
  if ((threadIdx.x & 63u) == 0u)
  {
    for (int k = 0; k < PT_PHASE_SLOTS; k++)
      pt_phase_acc[threadIdx.x >> 6][k] = 0;
    pt_phase_last[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime();
  }
#endif
  SceneCtx S_init = stage_scene<true, FILT_LDS, true>(L, lds);
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
    stack.lo = reinterpret_cast<uint16_t *>(lds + (PT_GEOM_STRIDE * (size_t)S.n_sph + PT_MAT_STRIDE * (size_t)(L.scene.n_spheres + L.scene.n_meshes) +
                                                   pt_filt_pair_slots(S.n_sph)));
    stack.hi = reinterpret_cast<uint8_t *>(stack.lo + (size_t)levels * PT_BLOCK);
#ifdef PT_BVH_WIDE
    stack.cap = levels; /* the LDS array keeps the binary walk's size; deeper entries overflow (WalkStack) */
    stack.ovf = nullptr; /* set below, once the wave's ring is known */
#endif
  }
  {
    unsigned long long *z = &pix_sum_all[0][0];
    for (uint32_t k = threadIdx.x; k < (PT_BLOCK / 64) * PT_TILE_PIXELS * 3; k += PT_BLOCK)
      z[k] = 0;
    if (threadIdx.x < (PT_BLOCK / 64) * 3)
      (&pix_nan_all[0][0])[threadIdx.x] = 0;
  }
  if (threadIdx.x == 0)
    park_slot_lds = pt_park_acquire(L);
  camera_to_lds(L, cam_lds);
  PHASE(9);
  __syncthreads();
  PHASE(10);

  const uint32_t wave = threadIdx.x >> 6;
  /* work units = tile_count x sample_chunks, chunk-major (consecutive units are different tiles); wave w of
   * workgroup b takes unit 4 b + w; the last workgroup may have waves without a unit (pool = 0) */
  const uint32_t unit = blockIdx.x * (PT_BLOCK / 64) + wave;
  const bool has_unit = unit < L.tile_count * L.sample_chunks;
  const uint32_t slot = has_unit ? unit % L.tile_count : 0u, chunk = has_unit ? unit / L.tile_count : 0u;
  const uint32_t tile = L.tile_first + slot * L.tile_stride;
  const uint32_t tx0 = (tile % L.tiles_x) * PT_TILE, ty0 = (tile / L.tiles_x) * PT_TILE;
  const uint32_t vcols = min((uint32_t)PT_TILE, (uint32_t)L.width - tx0);
  const uint32_t vrows = min((uint32_t)PT_TILE, (uint32_t)L.height - ty0);
  const uint32_t n_valid = vcols * vrows;
  const uint32_t spp = (uint32_t)L.samples;
  const uint32_t s_begin = (uint32_t)(((uint64_t)chunk * spp) / L.sample_chunks);
  const uint32_t s_end = (uint32_t)(((uint64_t)(chunk + 1u) * spp) / L.sample_chunks);
  const uint32_t park_slot = park_slot_lds;
  const bool ring_ok = park_slot != 0xFFFFFFFFu;
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
#ifdef PT_BVH_WIDE
  stack.ovf = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ring.f) + PT_PARK_Q * 128u + 512u);
#endif
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
  int stack_n = 0;
  const PendStack no_stack = {nullptr, 0, 0u, 0u};
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
          wu[lane] = (dp & 63u) | META_RESUMED | ((dp >> 6) << META_DEPTH_SHIFT);
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
        P.depth = (int)(meta >> META_DEPTH_SHIFT);
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
      (void)trace_step<1, false, CHECKER, TRIS, FILT_LDS, 1, true, true>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit);
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
      step_done = trace_step<1, false, CHECKER, TRIS, FILT_LDS, 2, true, true>(S, P, n_casts, diag_ptr, no_stack, stack_n, &hit);
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
        P.Ls = {0, 0, 0};
      }
      if (step_done)
        busy = false;
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
    const uint32_t slot = unit_e % L.tile_count, chunk = unit_e / L.tile_count;
    const uint32_t tile_e = L.tile_first + slot * L.tile_stride;
    const uint32_t vcols = min((uint32_t)PT_TILE, (uint32_t)L.width - (tile_e % L.tiles_x) * PT_TILE);
    const uint32_t vrows = min((uint32_t)PT_TILE, (uint32_t)L.height - (tile_e / L.tiles_x) * PT_TILE);
    const uint32_t n_valid = vcols * vrows;
    unsigned long long *const pix_sum = pix_sum_all[wave_now()];
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
    __builtin_amdgcn_wave_barrier();
    if (L.sample_chunks == 1)
    {
      const uint32_t t = lane;
      const bool inside = (t & 7u) < vcols && (t >> 3) < vrows;
      const double inv_s = 1.0 / (double)L.samples;
      const double quiet_nan = __longlong_as_double(0x7FF8000000000000ll);
      V3 mean;
      mean.x = ((double)(long long)pix_sum[3 * t + 0] * L.acc_inv_scale) * inv_s;
      mean.y = ((double)(long long)pix_sum[3 * t + 1] * L.acc_inv_scale) * inv_s;
      mean.z = ((double)(long long)pix_sum[3 * t + 2] * L.acc_inv_scale) * inv_s;
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
  if (threadIdx.x == 0 && ring_ok)
    atomicExch(&L.park_flags[park_slot], 0u); /* every wave is past its last ring access */
}

/* Kernel family pt_render_tiles[_tri][_big][_chk|_refr], picked by scene content
 * (pt_pick_kernel): "_tri" = scene has triangles; "_big" = the filter table is not in LDS
 * (more than PT_FILT_LDS_MAX primitives, or centres / radii beyond fp32's comfortable range):
 * table by scalar loads, NaN-safe compares, triangles through the hierarchy; "_chk" = scene
 * has M_CHECKERED materials (atan2_tab / frac1); "_refr" = scene has M_REFRACTION materials
 * (static body + pending second children in the pool; also covers M_CHECKERED) -- small staged sphere scenes take
 * pt_render_tiles_refr_pool, the pooled body, instead.
 * pt_render_tiles itself is the headline configuration: diffuse / mirror / emissive spheres,
 * small scene. */
#define PT_KERNEL_G(name, bounds, CHECKER, TRIS, FILT_LDS, GEOM_LDS)                         \
  extern "C" __global__ bounds void name(const PtLaunch L)                                  \
  {                                                                                         \
    render_tiles_pooled<CHECKER, TRIS, FILT_LDS, GEOM_LDS>(L);                              \
  }
#define PT_KERNEL(name, bounds, CHECKER, TRIS, FILT_LDS) PT_KERNEL_G(name, bounds, CHECKER, TRIS, FILT_LDS, true)
PT_KERNEL(pt_render_tiles, __launch_bounds__(PT_BLOCK, PT_MIN_WAVES), false, false, true)
PT_KERNEL(pt_render_tiles_big, __launch_bounds__(PT_BLOCK, PT_MIN_WAVES), false, false, false)
PT_KERNEL(pt_render_tiles_tri, __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_TRI), false, true, true)
PT_KERNEL(pt_render_tiles_tri_big, __launch_bounds__(PT_BLOCK, 4), false, true, false) /* 24 KB of traversal stacks: 4 workgroups per CU */
/* the hierarchy kernels with parked walks; the pooled ones above stay selectable (RT_HIP_KERNEL_VARIANT=2) for A/B */
#ifndef PT_MIN_WAVES_QUEUED
#define PT_MIN_WAVES_QUEUED 4
#endif
extern "C" __global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_QUEUED) void pt_render_tiles_tri_queued(const PtLaunch L)
{
  render_tiles_queued<false>(L);
}
extern "C" __global__ __launch_bounds__(PT_BLOCK) void pt_render_tiles_tri_queued_chk(const PtLaunch L)
{
  render_tiles_queued<true>(L);
}
/* round meshes: the probe is the triangles' bounding sphere alone (bvh_probe) */
extern "C" __global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_QUEUED) void pt_render_tiles_tri_queued_sph(const PtLaunch L)
{
  render_tiles_queued<false, true>(L);
}
/* small sphere scenes with M_REFRACTION on the pooled body (render_tiles_pooled, REFR) */
#ifndef PT_MIN_WAVES_REFR_POOL
#define PT_MIN_WAVES_REFR_POOL 4
#endif
extern "C" __global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_REFR_POOL) void pt_render_tiles_refr_pool(const PtLaunch L)
{
  render_tiles_pooled<true, false, true, true, true>(L);
}
/* ... and the same for sphere scenes that are streamed (beyond the staging budget, or beyond ~85 spheres by preference): geometry and
 * materials from memory, the sign-form table by scalar loads, as pt_render_tiles_pool_mem_s -- the reference's own generator gives a
 * fifth of its spheres M_REFRACTION (main.c:107-115), so a large packed room is exactly this scene class */
extern "C" __global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_REFR_POOL) void pt_render_tiles_refr_pool_mem(const PtLaunch L)
{
  render_tiles_pooled<true, false, true, false, true>(L);
}
/* ... and for small scenes with a mesh (the flat filter + fp32 pre-test kernels' scene class) */
extern "C" __global__ __launch_bounds__(PT_BLOCK, 3) void pt_render_tiles_tri_refr_pool(const PtLaunch L)
{
  render_tiles_pooled<true, true, true, true, true>(L);
}
PT_KERNEL(pt_render_tiles_chk, __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_CHK), true, false, true)
PT_KERNEL(pt_render_tiles_big_chk, __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_CHK), true, false, false)
PT_KERNEL(pt_render_tiles_tri_chk, __launch_bounds__(PT_BLOCK), true, true, true)
PT_KERNEL(pt_render_tiles_tri_big_chk, __launch_bounds__(PT_BLOCK), true, true, false)
/* Scenes whose sphere geometry + materials exceed the LDS staging budget (pt_geom_in_lds: more than 256 spheres): the SAME
 * pooled body -- job pool, swap, fixed-point sums, four rejection rounds per trip, sample chunks -- with geometry and
 * materials gathered from memory (PtSceneView.geom4 / material: 32 + 64 bytes per sphere, L2-resident up to tens of
 * thousands of spheres) and the filter table streamed through scalar loads as in the _big kernels.  Until round 4 the
 * 257th sphere dropped a scene onto pt_render_tiles_mem, the static body with every material's code and a 2.7 KB
 * private stack per lane (still the kernel of such scenes WITH M_REFRACTION, whose throughput is unbounded). */
PT_KERNEL_G(pt_render_tiles_pool_mem, __launch_bounds__(PT_BLOCK, PT_MIN_WAVES), false, false, false, false)
PT_KERNEL_G(pt_render_tiles_pool_mem_chk, __launch_bounds__(PT_BLOCK), true, false, false, false)
PT_KERNEL_G(pt_render_tiles_pool_mem_tri, __launch_bounds__(PT_BLOCK, 4), false, true, false, false)
PT_KERNEL_G(pt_render_tiles_pool_mem_tri_chk, __launch_bounds__(PT_BLOCK), true, true, false, false)
/* ... and, for sphere-only scenes within fp32's comfortable range (no centre or radius beyond 1e17), with the small scenes' FORM
 * of the filter -- sign tests, per-tile culling of the primary trips, the walls pruned among themselves -- read from memory
 * (stage_scene, FILT_FROM_MEMORY).  Measured on rooms packed as main.c:65-138 would (tools/many_spheres.py, profiles/r04_many_spheres.txt). */
PT_KERNEL_G(pt_render_tiles_pool_mem_s, __launch_bounds__(PT_BLOCK, PT_MIN_WAVES), false, false, true, false)
PT_KERNEL_G(pt_render_tiles_pool_mem_s_chk, __launch_bounds__(PT_BLOCK), true, false, true, false)
#undef PT_KERNEL
#undef PT_KERNEL_G

/* ---- static body: lane = (pixel, sample slice), fp64 partial sums ------------------------
 * Lane l of wave w: pixel (l >> 2) of the wave's 16, sample slice (l & 3): samples s = slice,
 * slice + 4, ...; the four slice sums of a pixel are combined by xor-shuffles in a fixed
 * order.  Floating-point sums have no range limit, which is what scenes with M_REFRACTION
 * need (see render_tiles_pooled); VARIANT 0 of it is the plain reference kernel
 * (RT_HIP_KERNEL_VARIANT=0). */
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
      pend_slot_lds = pt_pool_acquire(L.pend_flags, L.pend_slots_per_xcd);
    __syncthreads();
  }
  const uint32_t pend_slot = STACKED ? pend_slot_lds : 0u;
  /* (no slot: a sizing bug of the pool, never seen -- the launcher refuses to launch without a pool.  The tile then comes
   * out NaN, bytes 255, rather than wrong: see the epilogue) */
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

/* launch bounds of the M_REFRACTION kernels (waves per SIMD).  Until round 4 they had none: 203-232 VGPRs and a 2.7 KB private
 * stack, two waves per SIMD.  With the pending rays in the pool (PendStack) and the material code's library calls gone
 * (atan2_tab, cube, frac1) the sphere kernels need 127 VGPRs and the mesh kernels ~150, without scratch */
#ifndef PT_MIN_WAVES_REFR
#define PT_MIN_WAVES_REFR 4
#endif
#ifndef PT_MIN_WAVES_REFR_TRI
#define PT_MIN_WAVES_REFR_TRI 3
#endif
#define PT_KERNEL_STATIC(name, WAVES, VARIANT, REFRACT, CHECKER, TRIS, FILT_LDS)             \
  extern "C" __global__ __launch_bounds__(PT_BLOCK, WAVES) void name(const PtLaunch L)      \
  {                                                                                         \
    render_tiles_static<VARIANT, REFRACT, CHECKER, TRIS, FILT_LDS, 0, true>(L);             \
  }
PT_KERNEL_STATIC(pt_render_tiles_v0, 1, 0, false, true, true, false)
PT_KERNEL_STATIC(pt_render_tiles_refr, PT_MIN_WAVES_REFR, 1, true, true, false, true)
PT_KERNEL_STATIC(pt_render_tiles_big_refr, PT_MIN_WAVES_REFR, 1, true, true, false, false)
PT_KERNEL_STATIC(pt_render_tiles_tri_refr, PT_MIN_WAVES_REFR_TRI, 1, true, true, true, true)
PT_KERNEL_STATIC(pt_render_tiles_tri_big_refr, PT_MIN_WAVES_REFR_TRI, 1, true, true, true, false)
#undef PT_KERNEL_STATIC

/* cast_ray kernels: the static body with whitted_step, without a pending-ray stack (scenes with
 * a material that has both M_REFLECTION and M_REFRACTION take pt_whitted_tiles_mem) */
/* Launch bounds measured on the MI355X (1920x1080 x 64 spp, ms at 4 / 3 / 2 waves per SIMD): spheres
 * (config 4) 8.2 / 8.4 / 9.4; small mesh (config 3) 6.2 / 5.2 / 6.5; hierarchy (config 5, 4K x 8 spp)
 * 8.6 / 8.1 / 8.9.  None of the three is free of scratch below 2 waves (228-308 B at 4, 44-156 B at 3). */
#ifndef PT_MIN_WAVES_WHITTED
#define PT_MIN_WAVES_WHITTED 4
#endif
#ifndef PT_MIN_WAVES_WHITTED_TRI
#define PT_MIN_WAVES_WHITTED_TRI 3
#endif
#define PT_KERNEL_WHITTED(name, WAVES, TRIS, FILT_LDS)                                       \
  extern "C" __global__ __launch_bounds__(PT_BLOCK, WAVES) void name(const PtLaunch L)      \
  {                                                                                         \
    render_tiles_static<1, false, true, TRIS, FILT_LDS, 1, true>(L);                        \
  }
PT_KERNEL_WHITTED(pt_whitted_tiles, PT_MIN_WAVES_WHITTED, false, true)
PT_KERNEL_WHITTED(pt_whitted_tiles_big, PT_MIN_WAVES_WHITTED, false, false)
PT_KERNEL_WHITTED(pt_whitted_tiles_tri, PT_MIN_WAVES_WHITTED_TRI, true, true)
PT_KERNEL_WHITTED(pt_whitted_tiles_tri_big, PT_MIN_WAVES_WHITTED_TRI, true, false)
#undef PT_KERNEL_WHITTED

/* Scenes whose sphere geometry + materials exceed the LDS staging budget (pt_geom_in_lds: more
 * than ~256 spheres, or thousands of meshes): the most general static body -- every material,
 * triangles through the hierarchy -- reading geometry and materials from memory.  The O(n)
 * sphere scan dominates such scenes whatever the kernel around it does. */
extern "C" __global__ __launch_bounds__(PT_BLOCK) void pt_render_tiles_mem(const PtLaunch L)
{
  render_tiles_static<1, true, true, true, false, 0, false>(L);
}
extern "C" __global__ __launch_bounds__(PT_BLOCK) void pt_whitted_tiles_mem(const PtLaunch L)
{
  render_tiles_static<1, false, true, true, false, 2, false>(L);
}

/* Second pass of a chunked render: per-tile fixed-point sums -> float3 + tonemapped bytes. */
extern "C" __global__ __launch_bounds__(PT_BLOCK) void pt_resolve_tiles(const PtLaunch L)
{
  __shared__ float out_f[PT_TILE_PIXELS * 3];
  __shared__ uint8_t out_b[PT_TILE_PIXELS * 3 + 64];
  const uint32_t slot = blockIdx.x;
  const uint32_t tile = L.tile_first + slot * L.tile_stride;
  finish_pixels(L, L.acc_ws + (size_t)slot * (PT_TILE_PIXELS * 3),
                L.acc_ws + (size_t)L.tile_count * (PT_TILE_PIXELS * 3) + (size_t)slot * 3, tile, out_f, out_b);
  __syncthreads();
  if (threadIdx.x < PT_TILE_PIXELS * 3)
    L.tiles_rgb[(size_t)slot * (PT_TILE_PIXELS * 3) + threadIdx.x] = out_f[threadIdx.x];
  if (L.tiles_rgb8 && threadIdx.x < PT_TILE_PIXELS * 3 / 4)
    reinterpret_cast<uint32_t *>(L.tiles_rgb8)[(size_t)slot * (PT_TILE_PIXELS * 3 / 4) + threadIdx.x] =
        reinterpret_cast<const uint32_t *>(out_b)[threadIdx.x];
}

/* Self-test hook (rt_hip_selftest_math): evaluates the kernel's exact-arithmetic shortcuts
 * on caller data so a test can compare them bit for bit with the host's IEEE results.
 * op 0: sqrt_unscaled(a[i]);  op 1: div_small_int(a[i], b[i], 1/b[i]);  op 2: the library
 * sqrt(a[i]);  op 3: a[i] / b[i];  op 4: rnd_pm1-style fused r * 2^-30 - 1 with r = a[i];  op 5: rcp_unscaled(a[i]);
 * op 6: atan2_tab(a[i], b[i]);  op 7: frac1(a[i]) (= fmod(a[i], 1.0));  op 8: win_add of every a[i] into one accumulator (out[0..6]). */
extern "C" __global__ __launch_bounds__(256) void pt_selftest_math(int op, const double *a, const double *b,
                                                                  double *out, size_t n)
{
  __shared__ double tab[PT_ATAN_TAB];
  atan_table_to_lds(tab);
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    double r = 0;
    if (op == 0)
      r = sqrt_unscaled(a[i]);
    else if (op == 1)
      r = div_small_int(a[i], b[i], 1.0 / b[i]);
    else if (op == 2)
      r = sqrt(a[i]);
    else if (op == 3)
      r = a[i] / b[i];
    else if (op == 4)
      r = __builtin_fma(a[i], 1.0 / 1073741824.0, -1.0);
    else if (op == 5)
      r = rcp_unscaled(a[i]);
    else if (op == 6)
      r = atan2_tab(a[i], b[i], tab);
    else if (op == 7)
      r = frac1(a[i]);
    else if (op == 8)
    { /* the windowed pixel sums of pt_render_tiles_refr_pool: every a[i] into ONE accumulator, out[0 .. PT_WIN_N) (zeroed by the
       * caller); out[PT_WIN_N] counts the values win_add refused (non-finite, or at least 2^128) */
      if (!win_add(reinterpret_cast<unsigned long long *>(out), a[i]))
        atomicAdd(reinterpret_cast<unsigned long long *>(out) + PT_WIN_N, 1ull);
      continue;
    }
    out[i] = r;
  }
}

/* Self-test hook (rt_hip_selftest_intersect): the kernel's own exact primitive tests and its
 * phase-1 filter on caller data, one lane per case, so that known-answer vectors generated by
 * the compiled reference (tests/golden/primitives.npz) reach exact_sphere / exact_triangle /
 * filter_chunk themselves and not only their host twins.
 *   kind 0: prims = n x 4 (cx cy cz r*r), the record exact_sphere reads in the render kernels;
 *   kind 1: prims = n x 9 (v0, e1, e2), the record exact_triangle reads.
 * Case i = ray i against primitive i: hit[i], tuv[3i..] = t, and for triangles the barycentric
 * u, v the render kernels blend texture coordinates with.
 * Filter: one workgroup (one wavefront) per block of 64 cases; `filt` is the table
 * pt_build_filter made of all n primitives for near_R, so block b's pairs are those of the 64
 * primitives of the block.  Every lane runs phase 1 over them with its own ray, in the three
 * forms the render kernels use, and stores the 64-bit keep masks:
 *   keep[3i + 0]  spheres: sign-test form from LDS (pt_render_tiles); triangles: the per-lane fp32
 *                 Moeller-Trumbore pre-test (tri_may_hit32) instead
 *   keep[3i + 1]  compare form from LDS, push_keep_bit (pt_render_tiles_tri)
 *   keep[3i + 2]  compare form, table by scalar loads (the _big kernels)
 * bit j = ray i keeps primitive 64 b + j.  A test can so check 64 n (ray, primitive) pairs:
 * the filter must keep every pair the exact test accepts. */
extern "C" __global__ __launch_bounds__(64) void pt_selftest_intersect(int kind, const double *rays, const double *prims,
                                                                      const f32x2 *filt, const float4 *tri32, uint32_t n,
                                                                      double near_R2, double filt_shift, double t_start,
                                                                      uint8_t *hit, double *tuv, unsigned long long *keep)
{
  __shared__ f32x2 filt_lds[PT_FILT_STRIDE * 33]; /* 32 pairs + the look-ahead pair */
  const uint32_t base = blockIdx.x * 64u;
  const uint32_t chunk = min(64u, n - base);
  for (uint32_t k = threadIdx.x; k < PT_FILT_STRIDE * 33; k += 64)
    filt_lds[k] = filt[PT_FILT_STRIDE * (size_t)(base >> 1) + k];
  __syncthreads();
  const uint32_t i = base + threadIdx.x;
  const bool live = i < n;
  const uint32_t src = live ? i : base; /* idle lanes of the last block shadow its first case */
  const V3 o = ld3(rays + 6 * (size_t)src), d = ld3(rays + 6 * (size_t)src + 3);

  double min_t = t_start, bu = 0, bv = 0;
  int best = -1;
  if (kind == 0)
    exact_sphere(prims + 4 * (size_t)src, 0u, o, d, min_t, best);
  else
    exact_triangle(prims + 9 * (size_t)src, 0u, o, d, min_t, best, bu, bv);

  uint32_t lo, hi;
  unsigned long long m0 = ~0ull, m1, m2;
  if (kind == 0)
  {
    const FiltRay fs = filter_ray<true>(o, d, filt_shift, near_R2);
    filter_chunk<false, true>(filt_lds, 0u, chunk, fs, lo, hi);
    m0 = ((unsigned long long)hi << 32) | lo;
  }
  const FiltRay fr = filter_ray<false>(o, d, filt_shift, near_R2);
  if (kind == 1)
  { /* the per-lane fp32 pre-test (tri_may_hit32) of this ray against every triangle of the block */
    m0 = 0;
    for (uint32_t j = 0; j < chunk; j++)
      if (fr.far_origin || tri_may_hit32(tri32 + (PT_TRI32_STRIDE / 4) * (size_t)(base + j), fr.ox, fr.oy, fr.oz, fr.dx.x, fr.dy.x, fr.dz.x))
        m0 |= 1ull << j;
  }
  filter_chunk<true, true>(filt_lds, 0u, chunk, fr, lo, hi);
  m1 = ((unsigned long long)hi << 32) | lo;
  filter_chunk<false, false>(filt, base, chunk, fr, lo, hi);
  m2 = ((unsigned long long)hi << 32) | lo;
  if (live)
  {
    hit[i] = best >= 0 ? 1 : 0;
    tuv[3 * (size_t)i + 0] = min_t;
    tuv[3 * (size_t)i + 1] = bu;
    tuv[3 * (size_t)i + 2] = bv;
    keep[3 * (size_t)i + 0] = m0;
    keep[3 * (size_t)i + 1] = m1;
    keep[3 * (size_t)i + 2] = m2;
  }
}

/* Self-test hook (rt_hip_selftest_xcc): which XCD each workgroup of a launch ran on, as the parked-walk kernels read it
 * (pt_park_acquire): counts[x] = workgroups that saw HW_REG_XCC_ID == x. */
extern "C" __global__ __launch_bounds__(64) void pt_selftest_xcc(unsigned int *counts)
{
  if (threadIdx.x == 0)
    atomicAdd(&counts[(uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u], 1u);
}

hipError_t pt_launch_selftest_xcc(unsigned int *counts, uint32_t n_workgroups, hipStream_t stream)
{
  hipLaunchKernelGGL(pt_selftest_xcc, dim3(n_workgroups), dim3(64), 0, stream, counts);
  return hipGetLastError();
}

hipError_t pt_launch_selftest(int op, const double *a, const double *b, double *out, size_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(pt_selftest_math, dim3(256), dim3(256), 0, stream, op, a, b, out, n);
  return hipGetLastError();
}

/* Builds the packed-fp32 phase-1 filter table for one launch (the thresholds depend on
 * near_R, i.e. on the camera).  Per pair: cx cy cz r2_hi neg_tol, two primitives per f32x2.
 * Bound (e = 2^-24, fp32 unit roundoff; a = |c| + |o| <= A := |c| + near_R; |d| <= 1.0001):
 *   c, o, d are rounded to fp32 (relative e each), L = c - o adds one rounding, so
 *   |L32 - L| <= 2.01 e a per component; each 3-term fused dot product adds <= 3 e of its
 *   magnitude.  Hence  |tca32 - tca| <= 6.2 e A   and   |d2_32 - d2| <= 20.5 e A^2,  where tca,
 *   d2 are the real-number values; the reference's own fp64 rounding of them (~1e-16
 *   relative) is absorbed by the 1.5x slack:
 *     drop  <=>  tca32 < -(Rb + 10 e A)   or   d2_32 > R2 + 32 e A^2       (never a false drop)
 *   sphere: R2 = r*r, Rb = 0 (intersect_sphere rejects tca < 0, raytracer.c:84);
 *   triangle: R2 = Rb^2 of its bounding sphere, Rb = that radius (the hit point is inside the
 *   bounding sphere, so the centre is at most Rb behind the origin).
 * Thresholds are rounded away from the accept region when stored as fp32. */
/* fp32 hierarchy nodes for one launch: the planes of a node's two children as (child 0,
 * child 1) pairs, boxes widened by 4 e (near_R + |b|) and rounded outward (see bvh_traverse);
 * then the two child references. */
extern "C" __global__ __launch_bounds__(256) void pt_build_bvh(const double *bvh_src, uint32_t n_nodes, double near_R,
                                                              float *nodes)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += gridDim.x * blockDim.x)
  {
    const double *src = bvh_src + PT_BVH_SRC_DOUBLES * (size_t)i;
    float *dst = nodes + PT_BVH_NODE_WORDS * (size_t)i;
    const double e = 5.9604644775390625e-08;
    for (int c = 0; c < 2; c++)
      for (int k = 0; k < 3; k++)
      {
        const double lo = src[6 * c + k], hi = src[6 * c + 3 + k];
        dst[4 * k + c] = __double2float_rd(lo - 4.0 * e * (near_R + fabs(lo)));
        dst[4 * k + 2 + c] = __double2float_ru(hi + 4.0 * e * (near_R + fabs(hi)));
      }
    const uint32_t *refs = reinterpret_cast<const uint32_t *>(src + 12);
    dst[12] = __uint_as_float(refs[0]);
    dst[13] = __uint_as_float(refs[1]);
    dst[14] = 0.f;
    dst[15] = 0.f;
  }
}

/* PT_BVH_WIDE builds: the four-wide device nodes for one near_R -- the same widening and outward rounding as pt_build_bvh,
 * planes grouped per axis: x lo of children 0..3, x hi, y lo, y hi, z lo, z hi, then the four references. */
extern "C" __global__ __launch_bounds__(256) void pt_build_bvh_wide(const double *src_nodes, uint32_t n_nodes, double near_R, float *nodes)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += gridDim.x * blockDim.x)
  {
    const double *src = src_nodes + PT_BVHW_SRC_DOUBLES * (size_t)i;
    float *dst = nodes + PT_BVHW_NODE_WORDS * (size_t)i;
    const double e = 5.9604644775390625e-08;
    for (int c = 0; c < 4; c++)
      for (int k = 0; k < 3; k++)
      {
        const double lo = src[6 * c + k], hi = src[6 * c + 3 + k];
        const bool none = lo > hi; /* no child: the box stays inverted */
        dst[8 * k + c] = none ? 3.0e38f : __double2float_rd(lo - 4.0 * e * (near_R + fabs(lo)));
        dst[8 * k + 4 + c] = none ? -3.0e38f : __double2float_ru(hi + 4.0 * e * (near_R + fabs(hi)));
      }
    const uint32_t *refs = reinterpret_cast<const uint32_t *>(src + 24);
    for (int c = 0; c < 4; c++)
      dst[24 + c] = __uint_as_float(refs[c]);
    dst[28] = dst[29] = dst[30] = dst[31] = 0.f;
  }
}

/* HULL FACETS (scene creation, once): triangle F is one if every corner p of every triangle of the scene has
 * m . (p - v0_F) <= tau for m = +n_F (PT_HULL_PLUS: the stored normal points outward) or m = -n_F (PT_HULL_MINUS).
 * What it buys (render_tiles_queued): a ray that starts at a hit point on F -- within delta of F's plane -- with
 * m . d > mu has m . (o + t d - v0) >= t mu - delta, so it can meet a triangle point only at t <= (tau + delta) / mu;
 * the launch picks mu so that this is below EPSILON / 4 (rt_hip_shim.hip, hull_margin_for), where intersect_triangle
 * rejects the hit (t > EPSILON, raytracer.c:150): the ray cannot hit any triangle, whatever the mesh looks like
 * elsewhere.  Every facet of a convex mesh is one; of config 5's bounces off the mesh 42 % of all hierarchy walks
 * were such rays, each ~15 node visits to find nothing (PT_DIAG counters, profiles/).  One thread per triangle over
 * all 3 n corners: quadratic, so only up to PT_HULL_MAX_TRIS triangles (3 x 10^8 plane tests for config 5: ~1 ms). */
extern "C" __global__ __launch_bounds__(256) void pt_build_hull_flags(const double *__restrict__ tri_geom,
                                                                      const double *__restrict__ tri_normal, uint32_t n_tri,
                                                                      double tau, uint32_t *tri_object)
{
  __shared__ double corner[3 * 256][3];
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = f < n_tri;
  const double *g = tri_geom + 9 * (size_t)(live ? f : 0u);
  const double *nn = tri_normal + 3 * (size_t)(live ? f : 0u);
  const double nx = nn[0], ny = nn[1], nz = nn[2], vx = g[0], vy = g[1], vz = g[2];
  double smax = -1.7976931348623157e308, smin = 1.7976931348623157e308;
  bool bad = !(nx == nx) || !(ny == ny) || !(nz == nz); /* a degenerate triangle has no plane */
  {
    /* shape: the bound on how far a computed hit point lies from F's plane grows with |e1||e2| / |e1 x e2| (the
     * rounding of intersect_triangle's t; derivation at hull_margin_for): facets sharper than 1/16 go without a flag */
    const double cx = g[4] * g[8] - g[5] * g[7], cy = g[5] * g[6] - g[3] * g[8], cz = g[3] * g[7] - g[4] * g[6];
    const double e1 = g[3] * g[3] + g[4] * g[4] + g[5] * g[5], e2 = g[6] * g[6] + g[7] * g[7] + g[8] * g[8];
    bad |= !((cx * cx + cy * cy + cz * cz) * 256.0 >= e1 * e2) || !(e1 * e2 > 0.0);
  }
  for (uint32_t base = 0; base < n_tri; base += 256)
  {
    __syncthreads();
    const uint32_t t = base + threadIdx.x;
    if (t < n_tri)
    {
      const double *q = tri_geom + 9 * (size_t)t;
      for (int k = 0; k < 3; k++)
        for (int a = 0; a < 3; a++)
          corner[3 * threadIdx.x + k][a] = k == 0 ? q[a] : q[a] + q[3 * k + a]; /* v0, v0 + e1, v0 + e2: as the kernels see it */
    }
    __syncthreads();
    const uint32_t n = 3u * min(256u, n_tri - base);
    for (uint32_t c = 0; c < n; c++)
    {
      const double s = (nx * (corner[c][0] - vx) + ny * (corner[c][1] - vy)) + nz * (corner[c][2] - vz);
      smax = fmax(smax, s);
      smin = fmin(smin, s);
      bad |= !(s == s);
    }
  }
  if (live)
  {
    uint32_t bits = 0u;
    if (!bad && smax <= tau)
      bits = PT_HULL_PLUS;
    else if (!bad && smin >= -tau)
      bits = PT_HULL_MINUS;
    tri_object[f] = (tri_object[f] & ~(PT_HULL_PLUS | PT_HULL_MINUS)) | bits;
  }
}

/* The fp32 triangle table of tri_may_hit32 for one near_R: v0, e1, e2 rounded to nearest, the four
 * thresholds formed in fp64 and rounded up. */
extern "C" __global__ __launch_bounds__(256) void pt_build_tri32(const double *tri_geom, uint32_t n_tri, double near_R,
                                                                float *out)
{
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tri; t += gridDim.x * blockDim.x)
  {
    const double *g = tri_geom + 9 * (size_t)t;
    float *f = out + PT_TRI32_STRIDE * (size_t)t;
    for (int k = 0; k < 9; k++)
      f[k] = (float)g[k];
    const double e = 5.9604644775390625e-08;
    const double l0 = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), l1 = sqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5]),
                 l2 = sqrt(g[6] * g[6] + g[7] * g[7] + g[8] * g[8]);
    const double S = (near_R + l0) * 1.0001;
    const double up = 1.0 + 4.0 * e;
    double Ea = 16.0 * e * l1 * l2 * up, KU = 20.0 * e * S * l2 * up, KV = 20.0 * e * S * l1 * up, KT = 20.0 * e * S * l1 * l2 * up;
    /* products near fp32's range (or non-finite input): the pre-test keeps the triangle whatever it computes */
    if (!(S * l1 * l2 < 1e30) || !(l1 * l2 < 1e30))
      Ea = __longlong_as_double(0x7FF0000000000000ll);
    f[9] = __double2float_ru(Ea);
    f[10] = __double2float_ru(KU);
    f[11] = __double2float_ru(KV);
    f[12] = __double2float_ru(KT);
    f[13] = f[14] = f[15] = 0.f;
  }
}

extern "C" __global__ __launch_bounds__(256) void pt_build_filter(const double *entry_src, uint32_t n_entries,
                                                                 double near_R, float *filt)
{
  const uint32_t n_slots = (n_entries + 1u) & ~1u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += gridDim.x * blockDim.x)
  {
    float *f = filt + 2 * PT_FILT_STRIDE * (size_t)(i >> 1) + (i & 1u);
    if (i < n_entries)
    {
      const double *src = entry_src + PT_ENTRY_SRC_STRIDE * (size_t)i; /* cx cy cz R2 |c| Rb */
      const double e = 5.9604644775390625e-08;                          /* 2^-24 */
      const double A = src[4] + near_R;
      f[0] = (float)src[0];
      f[2] = (float)src[1];
      f[4] = (float)src[2];
      /* (1 + 8 e): one e for this conversion, the rest for the roundings of the chain that
       * carries -r2_hi as its addend (scan_filtered, SHIFT form: 3 e max(r2_hi, A^2)) */
      f[6] = (float)((src[3] + 32.0 * e * A * A) * (1.0 + 8.0 * e));
      f[8] = -(float)((src[5] + 10.0 * e * A) * (1.0 + 4.0 * e));
      /* sign-test form (spheres, small scenes): kq = |c|^2 - r2_hi', formed in fp64 and rounded DOWN, with its
       * own widening r2_hi' = R2 + 40 e A^2 + 8 e | |c|^2 - R2 | (pt_sign_widen_r2).  Bound behind it (e = 2^-24, every input
       * rounded to fp32, fused 3-term chains, o' the pulled-back origin, |o'| <= near_R + tol_max; the bound is stated with
       * A' = |c| + near_R + tol_max while the code forms A = |c| + near_R: tol_max = 12 e (max |c| + near_R) <= 7.2e-7 A, so
       * A'^2 <= (1 + 1.5e-6) A^2 -- inside the 40 over 28 slack of the widening by five orders of magnitude):
       *   tca32 = fma(cz,dz, fma(cy,dy, fma(cx,dx, -o'.d))):  |tca32 - tca'| <= 8.2 e A   (2 e |c| inputs, 3 e A chain,
       *           5.1 e |o'| for o'.d);
       *   ll32  = fma(cz,-2oz, fma(cy,-2oy, fma(cx,-2ox, kq + |o'|^2))):
       *           |ll32 - (|c-o'|^2 - r2_hi')| <= e (5 |kq| + 9.1 |o'|^2 + 10 |c||o'|) <= 10 e A^2 + 5 e |kq|;
       *   q32   = fma(tca32, tca32, -ll32):  |q32 - (r2_hi' - d2)| <= 2 A 8.2 e A + e A^2 + 10 e A^2 + 5 e |kq|
       *           <= 28 e A^2 + 5 e |kq|  <  the widening (|kq| <= | |c|^2 - R2 | + 40 e A^2),
       * so d2 <= R2 in exact arithmetic implies q32 >= 0: never a false drop (the PT_DIAG build re-checks every
       * dropped sphere with the exact test: 0 violations). */
      {
        const double cc = src[4] * src[4]; /* |c| was rounded up by 1e-12: inside the slack */
        const double g = fabs(cc - src[3]);
        /* pt_device.h: the widening shared with the host's big_prune_for (= (40 e A^2 + 8 e g)(1 + 8 e), then 4 e g) */
        f[10] = __double2float_rd((cc - (src[3] + pt_sign_widen_r2(A, g))) - pt_sign_widen_kq(g));
      }
    }
    else
    { /* padding slot of an odd count: masked out by valid_lo / valid_hi in the scan */
      f[0] = f[2] = f[4] = 0.f;
      f[6] = -1.f;
      f[8] = 0.f;
      f[10] = 0.f;
    }
  }
}

/* Scatter compact tile-major buffers to row-major images: one thread per
 * (pixel-in-tile, tile); consecutive threads read consecutive floats. */
extern "C" __global__ __launch_bounds__(256) void pt_untile(const float *tiles_rgb, const uint8_t *tiles_rgb8,
                                                          int width, int height, uint32_t tiles_x,
                                                          uint32_t tile_first, uint32_t tile_stride,
                                                          uint32_t tile_count, float *image_rgb,
                                                          uint8_t *image_rgb8)
{
  const size_t total = (size_t)tile_count * PT_TILE_PIXELS;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x)
  {
    const uint32_t k = (uint32_t)(idx / PT_TILE_PIXELS), pit = (uint32_t)(idx % PT_TILE_PIXELS);
    const uint32_t tile = tile_first + k * tile_stride;
    const uint32_t x = (tile % tiles_x) * PT_TILE + (pit & 7u);
    const uint32_t y = (tile / tiles_x) * PT_TILE + (pit >> 3);
    if (x >= (uint32_t)width || y >= (uint32_t)height)
      continue;
    const size_t dst = ((size_t)y * width + x) * 3, src = idx * 3;
    if (image_rgb)
    {
      image_rgb[dst + 0] = tiles_rgb[src + 0];
      image_rgb[dst + 1] = tiles_rgb[src + 1];
      image_rgb[dst + 2] = tiles_rgb[src + 2];
    }
    if (image_rgb8)
    {
      image_rgb8[dst + 0] = tiles_rgb8[src + 0];
      image_rgb8[dst + 1] = tiles_rgb8[src + 1];
      image_rgb8[dst + 2] = tiles_rgb8[src + 2];
    }
  }
}

/* ---- launch wrappers (host side), declared in pt_device.h ---------------------- */

size_t pt_render_lds_bytes(const PtSceneView &sc)
{
  if (!pt_geom_in_lds(sc))
    return 0;
  size_t doubles = PT_GEOM_STRIDE * (size_t)sc.n_spheres + PT_MAT_STRIDE * (size_t)(sc.n_spheres + sc.n_meshes);
  const size_t n_entries = (size_t)sc.n_spheres + sc.n_triangles;
  if (pt_filter_in_lds(sc))
    doubles += pt_filt_pair_slots((uint32_t)n_entries) + (size_t)sc.n_triangles * (PT_TRI32_STRIDE / 2); /* f32x2 = one double-sized slot */
  return doubles * sizeof(double);
}

/* which member of the kernel family a launch of this scene takes (the selection of
 * pt_launch_render, also reported by rt_hip_kernel_name for profiles and bench lines) */
/* have_park_ws = false: the parked-walk kernels' workspace is missing (its allocation failed): the lane-waiting kernels */
static int pt_pick_kernel(const PtSceneView &scene, uint32_t integrator, int variant, const char **name, bool have_park_ws = true)
{
  static const char *const names[31] = {
      "pt_render_tiles",      "pt_render_tiles_big",      "pt_render_tiles_tri",      "pt_render_tiles_tri_big",
      "pt_render_tiles_chk",  "pt_render_tiles_big_chk",  "pt_render_tiles_tri_chk",  "pt_render_tiles_tri_big_chk",
      "pt_render_tiles_refr", "pt_render_tiles_big_refr", "pt_render_tiles_tri_refr", "pt_render_tiles_tri_big_refr",
      "pt_render_tiles_v0",   "pt_whitted_tiles",         "pt_whitted_tiles_big",     "pt_whitted_tiles_tri",
      "pt_whitted_tiles_tri_big", "pt_render_tiles_mem",  "pt_whitted_tiles_mem",
      "pt_render_tiles_tri_queued", "pt_render_tiles_tri_queued_chk", "pt_render_tiles_tri_queued_sph",
      "pt_render_tiles_pool_mem", "pt_render_tiles_pool_mem_chk", "pt_render_tiles_pool_mem_tri", "pt_render_tiles_pool_mem_tri_chk",
      "pt_render_tiles_pool_mem_s", "pt_render_tiles_pool_mem_s_chk", "pt_render_tiles_refr_pool", "pt_render_tiles_refr_pool_mem",
      "pt_render_tiles_tri_refr_pool"};
  const bool tris = scene.n_triangles != 0;
  const bool big = !pt_filter_in_lds(scene);
  const bool refr = scene.any_refract != 0, chk = scene.any_checker != 0;
  const bool cast_ray = integrator == 1;
  int which = cast_ray ? 13 + (tris ? 2 : 0) + (big ? 1 : 0) : (refr ? 8 : (chk ? 4 : 0)) + (tris ? 2 : 0) + (big ? 1 : 0);
  /* too large to stage, or cast_ray with two-child materials: the two general kernels */
  const bool in_memory = !pt_geom_in_lds(scene) || (cast_ray && scene.any_mirror_glass);
  if (in_memory)
  {
    which = cast_ray ? 18 : 17;
    /* trace_path without M_REFRACTION: the pooled body with geometry from memory (variant 3: the static kernel, for A/B) */
    if (!cast_ray && !refr && variant != 3)
      which = (!tris && !scene.wide_range && variant != 4) ? 26 + (chk ? 1 : 0) : 22 + (tris ? 2 : 0) + (chk ? 1 : 0); /* (variant 4: the compare-form kernel, for A/B) */
  }
  else if (variant == 0 && !refr && !cast_ray)
    which = 12;
  else if (!cast_ray && variant != 5 && pt_prefer_streaming(scene))
    which = 26 + (chk ? 1 : 0); /* a sphere scene that would fit the staging budget but is faster streamed (pt_device.h; variant 5: staged, for A/B) */
  else if ((which == 3 || which == 7) && variant != 2 && !scene.wide_range && scene.n_bvh_nodes < (1u << 23) && scene.n_triangles < (1u << (23 - PT_BVH_COUNT_BITS)))
    which = which == 3 ? (scene.mesh_round ? 21 : 19) : 20; /* hierarchy scenes: parked walks (variant 2, scenes beyond fp32's comfortable range,
                                   * whose filter needs the NaN-safe compares, and meshes whose references do not fit the walk's
                                   * 24-bit stack entries keep the lane-waiting pooled kernels) */
  /* refractive sphere scenes that are streamed (beyond the staging budget, or beyond ~85 spheres by preference: pt_stream_sized) */
  if (!cast_ray && refr && !tris && !scene.wide_range && variant != 7 && variant != 3 && (!pt_geom_in_lds(scene) || (variant != 5 && pt_stream_sized(scene))))
    which = 29;
  if (which == 8 && variant != 7)
    which = 28;
  if (which == 10 && variant != 7)
    which = 30; /* ... with a small mesh */ /* small staged sphere scenes with M_REFRACTION: the pooled body (variant 7: the static one, for A/B; the launcher also
                 * falls back to it for launches whose sample x depth product could overflow the windowed sums) */
  if (which >= 19 && which <= 21 && !have_park_ws)
    which = which == 20 ? 7 : 3; /* no ring workspace: the lane-waiting kernels need none */
  if (name)
    *name = names[which];
  return which;
}

bool pt_kernel_needs_pend_pool(const PtSceneView &scene, uint32_t integrator, int variant)
{
  const int which = pt_pick_kernel(scene, integrator, variant, nullptr);
  return (which >= 8 && which <= 11) || which == 17 || which == 18 || (which >= 28 && which <= 30); /* _refr, pt_render_tiles_mem, pt_whitted_tiles_mem, _refr_pool[_mem] */
}

const char *pt_kernel_name(const PtSceneView &scene, uint32_t integrator, int variant, bool have_park_ws)
{
  const char *name = nullptr;
  (void)pt_pick_kernel(scene, integrator, variant, &name, have_park_ws);
  return name;
}

/* The camera-dependent tables of a scene for one near_R (two ~2 us kernels): the packed-fp32
 * filter table and the fp32 hierarchy nodes.  The shim keeps them per (scene, near_R) and builds
 * them once (rt_hip_shim.hip, TableSet), never while a render that reads them can be in flight. */
hipError_t pt_launch_build_hull_flags(const double *tri_geom, const double *tri_normal, uint32_t n_tri, double tau,
                                      uint32_t *tri_object, hipStream_t stream)
{
  if (n_tri == 0 || n_tri > PT_HULL_MAX_TRIS)
    return hipSuccess;
  hipLaunchKernelGGL(pt_build_hull_flags, dim3((n_tri + 255u) / 256u), dim3(256), 0, stream, tri_geom, tri_normal, n_tri, tau,
                     tri_object);
  return hipGetLastError();
}

hipError_t pt_launch_build_tables(const PtSceneView &scene, double near_R, float *filt, float *bvh_nodes, hipStream_t stream)
{
  const uint32_t n_nodes = scene.n_bvh_nodes;
  /* every primitive gets a filter entry: small-scene kernels scan triangles through the flat
   * filter, the others read the sphere part only and walk the hierarchy for the triangles */
  const uint32_t n_entries = scene.n_spheres + scene.n_triangles;
  const uint32_t blocks = n_entries ? min(1024u, (n_entries + 255u) / 256u) : 0u;
  if (blocks)
    hipLaunchKernelGGL(pt_build_filter, dim3(blocks), dim3(256), 0, stream, scene.entry_src, n_entries, near_R, filt);
  if (n_nodes)
    hipLaunchKernelGGL(pt_build_bvh, dim3(min(1024u, (n_nodes + 255u) / 256u)), dim3(256), 0, stream, scene.bvh_src,
                       n_nodes, near_R, bvh_nodes);
  if (scene.n_bvhw_nodes) /* PT_BVH_WIDE builds */
    hipLaunchKernelGGL(pt_build_bvh_wide, dim3(min(1024u, (scene.n_bvhw_nodes + 255u) / 256u)), dim3(256), 0, stream, scene.bvhw_src,
                       scene.n_bvhw_nodes, near_R, bvh_nodes + pt_bvhw_offset_words(n_nodes));
  /* the pre-test table behind the pair table: in scan order for small scenes (staged in LDS with the pairs), in the
   * hierarchy's leaf order for large meshes (read from HBM at the leaves) */
  if (scene.n_triangles != 0 && (pt_filter_in_lds(scene) || n_nodes != 0))
    hipLaunchKernelGGL(pt_build_tri32, dim3(min(1024u, (scene.n_triangles + 255u) / 256u)), dim3(256), 0, stream,
                       pt_filter_in_lds(scene) ? scene.tri_geom : scene.tri_geom_leaf, scene.n_triangles, near_R,
                       filt + 2 * (size_t)pt_filt_pair_slots(n_entries));
  return hipGetLastError();
}

hipError_t pt_launch_render(const PtLaunch &launch, hipStream_t stream, int variant)
{
  /* development knob: RT_HIP_EXTRA_LDS=<bytes> of unused dynamic LDS per workgroup, to measure how a
   * kernel responds to fewer resident workgroups per CU */
  static const size_t extra_lds = [] {
    const char *e = getenv("RT_HIP_EXTRA_LDS");
    return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)0;
  }();
  size_t lds_bytes = pt_render_lds_bytes(launch.scene) + extra_lds;
  typedef void (*Kernel)(const PtLaunch);
  static const Kernel family[31] = {pt_render_tiles,      pt_render_tiles_big,      pt_render_tiles_tri,      pt_render_tiles_tri_big,
                                    pt_render_tiles_chk,  pt_render_tiles_big_chk,  pt_render_tiles_tri_chk,  pt_render_tiles_tri_big_chk,
                                    pt_render_tiles_refr, pt_render_tiles_big_refr, pt_render_tiles_tri_refr, pt_render_tiles_tri_big_refr,
                                    pt_render_tiles_v0,   pt_whitted_tiles,         pt_whitted_tiles_big,     pt_whitted_tiles_tri,
                                    pt_whitted_tiles_tri_big, pt_render_tiles_mem,  pt_whitted_tiles_mem,
                                    pt_render_tiles_tri_queued, pt_render_tiles_tri_queued_chk, pt_render_tiles_tri_queued_sph,
                                    pt_render_tiles_pool_mem, pt_render_tiles_pool_mem_chk, pt_render_tiles_pool_mem_tri,
                                    pt_render_tiles_pool_mem_tri_chk, pt_render_tiles_pool_mem_s, pt_render_tiles_pool_mem_s_chk,
                                    pt_render_tiles_refr_pool, pt_render_tiles_refr_pool_mem, pt_render_tiles_tri_refr_pool};
  int which = pt_pick_kernel(launch.scene, launch.integrator, variant, nullptr,
                             launch.park_ws != nullptr && launch.park_slots_per_xcd != 0u);
  /* the pooled refraction kernel's windowed sums hold 2^31 pieces per word: a sample of a refractive scene has at most
   * 2^(max_depth + 2) terms (a full binary tree of children), so keep samples x 2^(max_depth + 2) <= 2^30 -- any other launch
   * (4,097 spp at depth 16, say) takes the static kernel, whose fp64 sums have no such limit */
  if (which == 28 && !pt_refr_pool_fits(launch.samples, launch.max_depth))
    which = 8;
  if (which == 29 && !pt_refr_pool_fits(launch.samples, launch.max_depth))
    which = pt_geom_in_lds(launch.scene) ? 8 : 17;
  if (which == 30 && !pt_refr_pool_fits(launch.samples, launch.max_depth))
    which = 10;
  const Kernel kernel = family[which];
  if ((which >= 22 && which <= 27) || which == 29)
    lds_bytes = extra_lds; /* the in-memory pooled kernels stage nothing, whatever the scene's size */
  if (((which >= 8 && which <= 11) || which == 17 || which == 18 || (which >= 28 && which <= 30)) &&
      (launch.pend_ws == nullptr || launch.pend_entries < (uint32_t)launch.max_depth + 2u))
    return hipErrorInvalidValue; /* a kernel with a pending-ray stack needs its pool (rt_hip_shim.hip: pend_pool_for) */
  if (which >= 19 && which <= 21) /* the spheres' filter pairs, then per-lane traversal stacks (24-bit entries) sized by the tree, after the staged scene */
    lds_bytes += (size_t)pt_filt_pair_slots(launch.scene.n_spheres) * 8u +
                 (((size_t)max(launch.scene.bvh_depth, 1u) * PT_BLOCK * 3u + 15u) & ~(size_t)15u);
  if (lds_bytes > 64 * 1024)
  { /* the attribute belongs to the (kernel, current device) pair: set whenever it is needed -- a process-wide
     * "already raised" note would skip devices 1..N-1 of the multi-device path (round-2 advisor finding) */
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
      return e;
  }
  if (launch.sample_chunks > 1)
  {
    hipError_t e = hipMemsetAsync(launch.acc_ws, 0, (size_t)launch.tile_count * PT_ACC_WS_WORDS * sizeof(unsigned long long), stream);
    if (e != hipSuccess)
      return e;
  }
  /* the parked-walk kernels render a tile per wave, four work units per workgroup */
  const uint32_t n_units = launch.tile_count * launch.sample_chunks;
  hipLaunchKernelGGL(kernel, dim3((which >= 19 && which <= 21) ? (n_units + PT_BLOCK / 64 - 1) / (PT_BLOCK / 64) : n_units), dim3(PT_BLOCK), lds_bytes,
                     stream, launch);
  if (launch.sample_chunks > 1)
    hipLaunchKernelGGL(pt_resolve_tiles, dim3(launch.tile_count), dim3(PT_BLOCK), 0, stream, launch);
  return hipGetLastError();
}

hipError_t pt_launch_selftest_intersect(int kind, const double *rays, const double *prims, const double *entry_src,
                                        float *filt, float *tri32, uint32_t n, double near_R, double filt_shift,
                                        uint8_t *hit, double *tuv, unsigned long long *keep, hipStream_t stream)
{
  if (n == 0)
    return hipSuccess;
  hipLaunchKernelGGL(pt_build_filter, dim3(min(1024u, (n + 255u) / 256u)), dim3(256), 0, stream, entry_src, n, near_R, filt);
  if (kind == 1)
    hipLaunchKernelGGL(pt_build_tri32, dim3((n + 255u) / 256u), dim3(256), 0, stream, prims, n, near_R, tri32);
  hipLaunchKernelGGL(pt_selftest_intersect, dim3((n + 63u) / 64u), dim3(64), 0, stream, kind, rays, prims,
                     reinterpret_cast<const f32x2 *>(filt), reinterpret_cast<const float4 *>(tri32), n, near_R * near_R,
                     filt_shift, 1.7976931348623157e308, hit, tuv, keep);
  return hipGetLastError();
}

hipError_t pt_launch_untile(const float *tiles_rgb, const uint8_t *tiles_rgb8, int width, int height,
                            uint32_t tile_first, uint32_t tile_stride, uint32_t tile_count, float *image_rgb,
                            uint8_t *image_rgb8, hipStream_t stream)
{
  const uint32_t tiles_x = ((uint32_t)width + PT_TILE - 1) / PT_TILE;
  const size_t total = (size_t)tile_count * PT_TILE_PIXELS;
  uint32_t blocks = (uint32_t)((total + 255) / 256);
  if (blocks > 8192)
    blocks = 8192;
  hipLaunchKernelGGL(pt_untile, dim3(blocks), dim3(256), 0, stream, tiles_rgb, tiles_rgb8, width, height, tiles_x,
                     tile_first, tile_stride, tile_count, image_rgb, image_rgb8);
  return hipGetLastError();
}
