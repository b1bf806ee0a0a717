/* pt_math.h -- vectors, the RNG draws, fixed-point terms, the tonemap, the material code's math without library calls
 * (atan2_tab, cube, pow10, frac1), and the PT_DIAG / PT_PHASE instrumentation macros.
 * Part of the one translation unit pt_kernel.hip (included there, in this order: pt_math.h, pt_intersect.h, pt_filter.h,
 * pt_scene_ctx.h, pt_trace.h, pt_body_pooled.h, pt_body_queued.h, pt_body_static.h); device code for gfx950 only. */
#ifndef PT_MATH_H
#define PT_MATH_H

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

#endif /* PT_MATH_H */
