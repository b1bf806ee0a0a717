/* rt_rng.h -- the sampling RNG shared by every side of the parity contract.
 *
 * The reference draws from glibc rand() (raytracer.c:227, seeded once in
 * main.c:186).  That is one global sequential stream with data-dependent draw
 * counts, which no parallel device can reproduce.  BASELINE.json's north_star
 * therefore substitutes "the same xorshift RNG re-seeded per pixel" on BOTH
 * sides.  This header is that RNG; it is included verbatim by
 *   - oracle/ref_harness.c   (injected under the reference's compiled code
 *                             through `#define rand`),
 *   - oracle/pt_oracle.c     (the CPU restatement),
 *   - raytracer.c_amd/csrc/  (the HIP kernel, as __device__ code).
 *
 * Stream key = (seed, pixel index y*W+x, sample index s).  Re-seeding at every
 * (pixel, sample) -- a refinement of "per pixel" -- makes a sample's value
 * independent of which lane / tile / GPU computes it and of the samples drawn
 * before it, so samples of one pixel can be spread over lanes.
 *
 * Generator: xorshift64 (Marsaglia 13,7,17), state seeded through the
 * splitmix64 finaliser.  Output: the top 31 bits, i.e. a value in
 * [0, 2^31) exactly like glibc's rand() with RAND_MAX = 2^31-1, so the
 * reference's `rand() / (RAND_MAX + 1.0)` stays an exact power-of-two scale.
 */
#ifndef RT_RNG_H
#define RT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_RNG_FN __host__ __device__ static inline
#else
#define RT_RNG_FN static inline
#endif

RT_RNG_FN uint64_t rt_mix64(uint64_t z)
{
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}

/* The key is hashed in two steps so that the per-pixel half can be formed once per pixel. */
RT_RNG_FN uint64_t rt_rng_pixel_key(uint64_t seed, uint32_t pixel)
{
  return rt_mix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)pixel + 1u));
}

RT_RNG_FN uint64_t rt_rng_sample_state(uint64_t pixel_key, uint32_t sample)
{
  const uint64_t h = rt_mix64(pixel_key + 0xD1B54A32D192ED03ull * ((uint64_t)sample + 1u));
  return h ? h : 0x9E3779B97F4A7C15ull; /* xorshift state must be non-zero */
}

/* State for sample `sample` of pixel `pixel` under global seed `seed`. */
RT_RNG_FN uint64_t rt_rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample)
{
  return rt_rng_sample_state(rt_rng_pixel_key(seed, pixel), sample);
}

/* Next rand()-compatible draw: 31 uniform bits. */
RT_RNG_FN uint32_t rt_rng_next31(uint64_t *state)
{
  uint64_t x = *state;
  x ^= x << 13;
  x ^= x >> 7;
  x ^= x << 17;
  *state = x;
  return (uint32_t)(x >> 33);
}

/* random_double() of the reference (raytracer.c:227): r / 2^31, exact. */
RT_RNG_FN double rt_rng_double(uint64_t *state)
{
  return (double)rt_rng_next31(state) * (1.0 / 2147483648.0);
}

#endif /* RT_RNG_H */
