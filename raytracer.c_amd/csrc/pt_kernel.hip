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
 * travel with a path), pt_render_tiles_tri_queued* (hierarchy scenes: parked walks; _refr: with M_REFRACTION), pt_render_tiles[..]_refr and
 * pt_whitted_tiles[..] (static body: refraction's two-child tree where the pooled kernel does not apply, and cast_ray,
 * raytracer.c:556-641), see pt_pick_kernel.
 *
 * pt_render_tiles_v0 (development builds only, -DPT_DEV_KERNELS: kept for A/B and as the plainest statement of the algorithm): static
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
 *
 * Layout.  The device code is ONE translation unit -- every body is a template instantiated below, and the kernels share
 * their inlined pieces -- split by topic into headers that are included here, in this order, and nowhere else:
 *   pt_math.h         vectors, RNG draws, fixed-point terms, tonemap, atan2_tab / cube / frac1, PT_DIAG / PT_PHASE macros
 *   pt_intersect.h    exact_sphere / exact_triangle (fp64, the reference's operation order), the hierarchy in packed fp32
 *   pt_filter.h       the phase-1 filter (three forms), BigPrune, tile_cull, the fp32 triangle pre-test, scan_filtered
 *   pt_scene_ctx.h    SceneCtx / stage_scene, Path, pending-ray stacks, windowed sums, camera, start_sample
 *   pt_trace.h        trace_step (trace_path), whitted_step (cast_ray), finish_pixels / store_tile
 *   pt_body_pooled.h  render_tiles_pooled   (pt_render_tiles[_tri][_big][_chk], _pool_mem*, _refr_pool*)
 *   pt_body_queued.h  render_tiles_queued   (pt_render_tiles_tri_queued*: parked walks, also with M_REFRACTION)
 *   pt_body_static.h  render_tiles_static   (pt_render_tiles_v0, *_refr, pt_whitted_tiles*, *_mem)
 * This file keeps the kernel entry points (the family, by scene content), the table-building and self-test kernels, pt_untile,
 * and the host-side launchers (pt_pick_kernel, pt_launch_render) declared in pt_device.h.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#include "pt_device.h"
#include "rt_rng.h"

#include "pt_math.h"
#include "pt_intersect.h"
#include "pt_filter.h"
#include "pt_scene_ctx.h"
#include "pt_trace.h"
#include "pt_body_pooled.h"
#include "pt_body_queued.h"
#include "pt_body_static.h"

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
/* the hierarchy kernels with parked walks; the pooled ones above are the table's park = NO rows (no ring workspace, or a scene beyond
 * fp32's comfortable range), and RT_HIP_KERNEL_VARIANT=2 of the development build for A/B */
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
extern "C" __global__ __launch_bounds__(PT_BLOCK) void pt_render_tiles_tri_queued_chk_sph(const PtLaunch L)
{
  render_tiles_queued<true, true>(L);
}
/* hierarchy scenes with M_REFRACTION: parked walks + windowed sums + pending second children that travel with a path, also
 * through the ring (render_tiles_queued, REFR); windowed sums in the workspace (global atomics): LDS as the other forms */
#ifndef PT_MIN_WAVES_QUEUED_REFR
#define PT_MIN_WAVES_QUEUED_REFR 3
#endif
extern "C" __global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_QUEUED_REFR) void pt_render_tiles_tri_queued_refr(const PtLaunch L)
{
  render_tiles_queued<true, false, true>(L);
}
extern "C" __global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_QUEUED_REFR) void pt_render_tiles_tri_queued_refr_sph(const PtLaunch L)
{
  render_tiles_queued<true, true, true>(L);
}
/* ... and the two forms for scenes whose SPHERES exceed the LDS staging budget next to such a mesh (render_tiles_queued,
 * GEOM_LDS = false: sphere geometry, materials and the spheres' filter pairs from memory); the probe is the general one */
extern "C" __global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES_QUEUED) void pt_render_tiles_tri_queued_mem(const PtLaunch L)
{
  render_tiles_queued<false, false, false, false>(L);
}
extern "C" __global__ __launch_bounds__(PT_BLOCK) void pt_render_tiles_tri_queued_mem_chk(const PtLaunch L)
{
  render_tiles_queued<true, false, false, false>(L);
}
/* (no refraction form here: at three waves it spills two doubles inside the trip loop, and scenes with M_REFRACTION, more
 * spheres than the staging holds AND a large mesh are the rarest class there is -- they keep pt_render_tiles_mem) */
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
#ifdef PT_DEV_KERNELS
PT_KERNEL_STATIC(pt_render_tiles_v0, 1, 0, false, true, true, false)
#endif
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
  if (L.acc_windows)
  { /* the M_REFRACTION forms: windowed sums (win_add), merged chunk by chunk in carry-normalised form */
    if (threadIdx.x < PT_TILE_PIXELS * 3)
    {
      const uint32_t t = threadIdx.x / 3u, c = threadIdx.x - 3u * t;
      const bool inside = (tile % L.tiles_x) * PT_TILE + (t & 7u) < (uint32_t)L.width && (tile / L.tiles_x) * PT_TILE + (t >> 3) < (uint32_t)L.height;
      unsigned long long w[PT_WIN_N];
#pragma unroll
      for (int k = 0; k < PT_WIN_N; k++)
        w[k] = L.acc_ws[((size_t)slot * (PT_TILE_PIXELS * 3) + threadIdx.x) * PT_WIN_N + k];
      win_normalize(w);
      double mean = win_value(w) * (1.0 / (double)L.samples);
      const unsigned long long nan_mask = L.acc_ws[(size_t)L.tile_count * (PT_TILE_PIXELS * 3 * PT_WIN_N) + (size_t)slot * 3 + c];
      mean = ((nan_mask >> t) & 1ull) ? __longlong_as_double(0x7FF8000000000000ll) : mean;
      out_f[threadIdx.x] = inside ? (float)mean : 0.f;
      out_b[threadIdx.x] = inside ? tonemap(mean) : 0;
    }
  }
  else
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

/* The kernel family: one row per member. */
enum PtKernelId
{
  K_TILES = 0, K_BIG, K_TRI, K_TRI_BIG,
  K_CHK, K_BIG_CHK, K_TRI_CHK, K_TRI_BIG_CHK,
  K_REFR, K_BIG_REFR, K_TRI_REFR, K_TRI_BIG_REFR,
  K_WHITTED, K_WHITTED_BIG, K_WHITTED_TRI, K_WHITTED_TRI_BIG,
  K_MEM, K_WHITTED_MEM,
  K_TRI_QUEUED, K_TRI_QUEUED_CHK, K_TRI_QUEUED_SPH,
  K_POOL_MEM, K_POOL_MEM_CHK, K_POOL_MEM_TRI, K_POOL_MEM_TRI_CHK, K_POOL_MEM_S, K_POOL_MEM_S_CHK,
  K_REFR_POOL, K_REFR_POOL_MEM, K_TRI_REFR_POOL,
  K_TRI_QUEUED_REFR, K_TRI_QUEUED_REFR_SPH, K_TRI_QUEUED_CHK_SPH,
  K_TRI_QUEUED_MEM, K_TRI_QUEUED_MEM_CHK,
#ifdef PT_DEV_KERNELS
  K_V0, /* the literal single-phase scan: development builds only (RT_HIP_KERNEL_VARIANT=0) */
#endif
  K_COUNT
};
typedef void (*PtKernelFn)(const PtLaunch);
struct PtKernelInfo
{
  const char *name;
  PtKernelFn fn;
  bool pend_pool;    /* pushes pending second children: needs a slot of the pending-ray pool (rt_hip_shim.hip, pend_pool_for) */
  bool queued;       /* parked walks: a tile per WAVE (four work units per workgroup), filter pairs + traversal stacks in dynamic LDS */
  bool stages_none;  /* geometry and tables from memory: no staged scene in LDS, whatever the scene's size */
  bool wide_pend;    /* 4 x 512 stacks per pool slot (path ids that travel through the ring) */
  bool chunks;       /* takes sample_chunks > 1 (integer partial sums merged by pt_resolve_tiles) */
  bool windowed;     /* ... as windowed sums (win_add): PT_ACC_WS_WORDS_WIN words per tile of the chunk workspace */
};
#define PT_K(fn, pend, queued, none, wide, chunks) {#fn, fn, pend, queued, none, wide, chunks, false}
#define PT_KW(fn, pend, queued, none, wide) {#fn, fn, pend, queued, none, wide, true, true}
static const PtKernelInfo pt_kernels[K_COUNT] = {
    PT_K(pt_render_tiles, false, false, false, false, true),          PT_K(pt_render_tiles_big, false, false, false, false, true),
    PT_K(pt_render_tiles_tri, false, false, false, false, true),      PT_K(pt_render_tiles_tri_big, false, false, false, false, true),
    PT_K(pt_render_tiles_chk, false, false, false, false, true),      PT_K(pt_render_tiles_big_chk, false, false, false, false, true),
    PT_K(pt_render_tiles_tri_chk, false, false, false, false, true),  PT_K(pt_render_tiles_tri_big_chk, false, false, false, false, true),
    PT_K(pt_render_tiles_refr, true, false, false, false, false),     PT_K(pt_render_tiles_big_refr, true, false, false, false, false),
    PT_K(pt_render_tiles_tri_refr, true, false, false, false, false), PT_K(pt_render_tiles_tri_big_refr, true, false, false, false, false),
    PT_K(pt_whitted_tiles, false, false, false, false, false),        PT_K(pt_whitted_tiles_big, false, false, false, false, false),
    PT_K(pt_whitted_tiles_tri, false, false, false, false, false),    PT_K(pt_whitted_tiles_tri_big, false, false, false, false, false),
    PT_K(pt_render_tiles_mem, true, false, false, false, false),      PT_K(pt_whitted_tiles_mem, true, false, false, false, false),
    PT_K(pt_render_tiles_tri_queued, false, true, false, false, true), PT_K(pt_render_tiles_tri_queued_chk, false, true, false, false, true),
    PT_K(pt_render_tiles_tri_queued_sph, false, true, false, false, true),
    PT_K(pt_render_tiles_pool_mem, false, false, true, false, true),  PT_K(pt_render_tiles_pool_mem_chk, false, false, true, false, true),
    PT_K(pt_render_tiles_pool_mem_tri, false, false, true, false, true), PT_K(pt_render_tiles_pool_mem_tri_chk, false, false, true, false, true),
    PT_K(pt_render_tiles_pool_mem_s, false, false, true, false, true), PT_K(pt_render_tiles_pool_mem_s_chk, false, false, true, false, true),
    PT_KW(pt_render_tiles_refr_pool, true, false, false, false), PT_KW(pt_render_tiles_refr_pool_mem, true, false, true, false),
    PT_KW(pt_render_tiles_tri_refr_pool, true, false, false, false),
    PT_KW(pt_render_tiles_tri_queued_refr, true, true, false, true), PT_KW(pt_render_tiles_tri_queued_refr_sph, true, true, false, true),
    PT_K(pt_render_tiles_tri_queued_chk_sph, false, true, false, false, true),
    PT_K(pt_render_tiles_tri_queued_mem, false, true, true, false, true), PT_K(pt_render_tiles_tri_queued_mem_chk, false, true, true, false, true),
#ifdef PT_DEV_KERNELS
    PT_K(pt_render_tiles_v0, false, false, false, false, false),
#endif
};
#undef PT_K
#undef PT_KW

/* ---- which member a launch takes: ONE table ------------------------------------------------------------------------------
 * A launch is classified by the six things the family is split by, and the first row of pt_pick_table that matches names the
 * kernel.  The fallbacks (no ring workspace, windowed sums that do not fit, a pending-ray pool that could not be had at
 * 4 x 512 stacks, scenes beyond fp32's comfortable range) are rows like any other.  A row field of ANY matches everything.
 *
 *   integ   PATH trace_path (raytracer.c:482-554) | CAST cast_ray (:556-641)
 *   stage   STAGED   sphere geometry + materials fit the LDS staging budget and are staged
 *           STREAM   they would fit, but the scene is a sphere scene of more than ~85 spheres: faster streamed (pt_stream_sized)
 *           LARGE    beyond the staging budget (pt_geom_in_lds false)
 *   mesh    NONE | FLAT triangles scanned through the flat filter staged in LDS (pt_filter_in_lds) | HIER triangles through the
 *           hierarchy | HIERBIG the same with more than PT_FILT_LDS_MAX triangles (what a scene beyond the staging budget needs
 *           for parked walks: HIERBIG rows also match where HIER is asked for -- see pt_row_matches)
 *   mat     PLAIN | CHK any M_CHECKERED | REFR any M_REFRACTION (wins over CHK: the _refr bodies carry the checker code) |
 *           for CAST: GLASS2 a material with M_REFLECTION and M_REFRACTION (two children per hit), else PLAIN
 *   range   NORMAL | WIDE a centre or radius beyond 1e17 (NaN-safe compare filter, no sign tests, no parked walks)
 *   park    YES the parked-walk body may run: its workspace exists and references fit 24-bit stack entries | NO
 *   round   YES the mesh's bounding sphere shows a ray no more than its box (probe = the sphere alone) | NO
 *   fit     YES the windowed sums of the pooled refraction kernels hold this launch (pt_refr_pool_fits per chunk) and, for the
 *           parked-walk refraction kernels, the pending-ray pool has 4 x 512 stacks per slot | NO
 */
enum { ANY = -1 };
enum PtInteg { PATH = 0, CAST = 1 };
enum PtStage { STAGED = 0, STREAM = 1, LARGE = 2 };
enum PtMesh { NONE = 0, FLAT = 1, HIER = 2, HIERBIG = 3 };
enum PtMat { PLAIN = 0, CHK = 1, REFR = 2, GLASS2 = 3 };
enum PtYesNo { NO = 0, YES = 1 };
enum PtRange { NORMAL = 0, WIDE = 1 };
struct PtPickKey
{
  int integ, stage, mesh, mat, range, park, round, fit;
};
struct PtPickRow
{
  int integ, stage, mesh, mat, range, park, round, fit;
  int kernel;
};
static const PtPickRow pt_pick_table[] = {
    /* integ stage   mesh     mat     range   park round fit   kernel */
    /* ---- cast_ray: the static body; two-child materials or a scene beyond the staging budget: the general in-memory kernel */
    {CAST, LARGE,  ANY,     ANY,    ANY,    ANY, ANY, ANY, K_WHITTED_MEM},
    {CAST, ANY,    ANY,     GLASS2, ANY,    ANY, ANY, ANY, K_WHITTED_MEM},
    {CAST, ANY,    NONE,    ANY,    NORMAL, ANY, ANY, ANY, K_WHITTED},
    {CAST, ANY,    NONE,    ANY,    WIDE,   ANY, ANY, ANY, K_WHITTED_BIG},
    {CAST, ANY,    FLAT,    ANY,    ANY,    ANY, ANY, ANY, K_WHITTED_TRI},
    {CAST, ANY,    HIER,    ANY,    ANY,    ANY, ANY, ANY, K_WHITTED_TRI_BIG},
    /* ---- trace_path, scenes beyond the staging budget */
    {PATH, LARGE,  NONE,    REFR,   NORMAL, ANY, ANY, YES, K_REFR_POOL_MEM},
    {PATH, LARGE,  ANY,     REFR,   ANY,    ANY, ANY, ANY, K_MEM},              /* ... with a mesh, out of range, or sums that do not fit: the static body */
    {PATH, LARGE,  HIERBIG, PLAIN,  NORMAL, YES, ANY, ANY, K_TRI_QUEUED_MEM},
    {PATH, LARGE,  HIERBIG, CHK,    NORMAL, YES, ANY, ANY, K_TRI_QUEUED_MEM_CHK},
    {PATH, LARGE,  NONE,    PLAIN,  NORMAL, ANY, ANY, ANY, K_POOL_MEM_S},
    {PATH, LARGE,  NONE,    CHK,    NORMAL, ANY, ANY, ANY, K_POOL_MEM_S_CHK},
    {PATH, LARGE,  NONE,    PLAIN,  WIDE,   ANY, ANY, ANY, K_POOL_MEM},
    {PATH, LARGE,  NONE,    CHK,    WIDE,   ANY, ANY, ANY, K_POOL_MEM_CHK},
    {PATH, LARGE,  ANY,     PLAIN,  ANY,    ANY, ANY, ANY, K_POOL_MEM_TRI},     /* a small mesh, no ring workspace, or out of range */
    {PATH, LARGE,  ANY,     CHK,    ANY,    ANY, ANY, ANY, K_POOL_MEM_TRI_CHK},
    /* ---- trace_path, sphere scenes that fit but stream by preference */
    {PATH, STREAM, NONE,    PLAIN,  NORMAL, ANY, ANY, ANY, K_POOL_MEM_S},
    {PATH, STREAM, NONE,    CHK,    NORMAL, ANY, ANY, ANY, K_POOL_MEM_S_CHK},
    {PATH, STREAM, NONE,    REFR,   NORMAL, ANY, ANY, YES, K_REFR_POOL_MEM},
    {PATH, STREAM, NONE,    REFR,   NORMAL, ANY, ANY, NO,  K_REFR},
    /* ---- trace_path, staged scenes: spheres only */
    {PATH, STAGED, NONE,    PLAIN,  NORMAL, ANY, ANY, ANY, K_TILES},            /* the headline: BASELINE configs 1, 2, 4 */
    {PATH, STAGED, NONE,    CHK,    NORMAL, ANY, ANY, ANY, K_CHK},
    {PATH, STAGED, NONE,    REFR,   NORMAL, ANY, ANY, YES, K_REFR_POOL},
    {PATH, STAGED, NONE,    REFR,   NORMAL, ANY, ANY, NO,  K_REFR},
    {PATH, STAGED, NONE,    PLAIN,  WIDE,   ANY, ANY, ANY, K_BIG},
    {PATH, STAGED, NONE,    CHK,    WIDE,   ANY, ANY, ANY, K_BIG_CHK},
    {PATH, STAGED, NONE,    REFR,   WIDE,   ANY, ANY, ANY, K_BIG_REFR},
    /* ---- ... with a small mesh (flat filter + fp32 pre-test): BASELINE config 3 */
    {PATH, STAGED, FLAT,    PLAIN,  ANY,    ANY, ANY, ANY, K_TRI},
    {PATH, STAGED, FLAT,    CHK,    ANY,    ANY, ANY, ANY, K_TRI_CHK},
    {PATH, STAGED, FLAT,    REFR,   ANY,    ANY, ANY, YES, K_TRI_REFR_POOL},
    {PATH, STAGED, FLAT,    REFR,   ANY,    ANY, ANY, NO,  K_TRI_REFR},
    /* ---- ... with a mesh through the hierarchy: parked walks (BASELINE config 5), else the lane-waiting kernels */
    {PATH, STAGED, HIER,    PLAIN,  NORMAL, YES, YES, ANY, K_TRI_QUEUED_SPH},
    {PATH, STAGED, HIER,    PLAIN,  NORMAL, YES, NO,  ANY, K_TRI_QUEUED},
    {PATH, STAGED, HIER,    CHK,    NORMAL, YES, YES, ANY, K_TRI_QUEUED_CHK_SPH},
    {PATH, STAGED, HIER,    CHK,    NORMAL, YES, NO,  ANY, K_TRI_QUEUED_CHK},
    {PATH, STAGED, HIER,    REFR,   NORMAL, YES, YES, YES, K_TRI_QUEUED_REFR_SPH},
    {PATH, STAGED, HIER,    REFR,   NORMAL, YES, NO,  YES, K_TRI_QUEUED_REFR},
    {PATH, STAGED, HIER,    PLAIN,  ANY,    ANY, ANY, ANY, K_TRI_BIG},          /* no ring workspace, or out of range */
    {PATH, STAGED, HIER,    CHK,    ANY,    ANY, ANY, ANY, K_TRI_BIG_CHK},
    {PATH, STAGED, HIER,    REFR,   ANY,    ANY, ANY, ANY, K_TRI_BIG_REFR},     /* ... or sums / pool that do not fit */
};

static bool pt_row_matches(const PtPickRow &r, const PtPickKey &k)
{
  auto ok = [](int row, int key) { return row == ANY || row == key; };
  /* mesh: a row that asks for HIER takes HIERBIG scenes too (a big mesh is a hierarchy mesh); one that asks for HIERBIG only those */
  const bool mesh_ok = r.mesh == ANY || r.mesh == k.mesh || (r.mesh == HIER && k.mesh == HIERBIG);
  return ok(r.integ, k.integ) && ok(r.stage, k.stage) && mesh_ok && ok(r.mat, k.mat) && ok(r.range, k.range) && ok(r.park, k.park) &&
         ok(r.round, k.round) && ok(r.fit, k.fit);
}

PtPickKey pt_classify(const PtSceneView &scene, const PtPickFacts &f)
{
  PtPickKey k;
  const bool cast = f.integrator == 1u;
  k.integ = cast ? CAST : PATH;
  k.stage = !pt_geom_in_lds(scene) ? LARGE : ((!cast && pt_stream_sized(scene)) ? STREAM : STAGED);
  k.range = scene.wide_range ? WIDE : NORMAL;
  if (scene.n_triangles == 0u)
    k.mesh = NONE;
  else if (pt_filter_in_lds(scene))
    k.mesh = FLAT;
  else
    k.mesh = (scene.n_triangles > PT_FILT_LDS_MAX && scene.n_bvh_nodes != 0u) ? HIERBIG : HIER;
  if (cast)
    k.mat = scene.any_mirror_glass ? GLASS2 : PLAIN;
  else
    k.mat = scene.any_refract ? REFR : (scene.any_checker ? CHK : PLAIN);
  /* the parked-walk body's conditions: a ring workspace, references that fit the walk's 24-bit stack entries (range: the rows) */
  k.park = (f.have_park_ws && scene.n_bvh_nodes < (1u << 23) && scene.n_triangles < (1u << (23 - PT_BVH_COUNT_BITS))) ? YES : NO;
  k.round = scene.mesh_round ? YES : NO;
  /* the pooled refraction kernels' windowed sums hold 2^31 pieces per word: a sample of a refractive scene has at most
   * 2^(max_depth + 2) terms (a full binary tree of children), so a launch needs samples x 2^(max_depth + 2) <= 2^30 */
  k.fit = (pt_refr_pool_fits(f.samples, f.max_depth) && (k.mesh < HIER || f.wide_pend_ok)) ? YES : NO;
  return k;
}

#ifdef PT_DEV_KERNELS
/* development builds (-DPT_DEV_KERNELS: `make shim-dev` -> librt_hip_dev.so): RT_HIP_KERNEL_VARIANT,
 * read once per process, rewrites the key so that a scene takes another arm of the table (A/B), or names the literal kernel:
 *   0 pt_render_tiles_v0 (the plainest statement of the algorithm)   2 hierarchy scenes on the lane-waiting kernels (park = NO)
 *   3 scenes beyond the staging budget on the static in-memory kernel  4 ... on the compare-form pooled kernel
 *   5 sphere scenes that would stream by preference are staged          7 refractive scenes on the static kernels (fit = NO) */
int pt_dev_variant()
{
  static const int v = [] {
    const char *e = getenv("RT_HIP_KERNEL_VARIANT");
    return (e && e[0] >= '0' && e[0] <= '7' && e[0] != '1' && e[0] != '6') ? e[0] - '0' : 1;
  }();
  return v;
}
static int pt_dev_pick(PtPickKey &k)
{
  const int v = pt_dev_variant();
  if (v == 0 && k.integ == PATH && k.stage != LARGE && k.mat != REFR)
    return K_V0;
  if (v == 2)
    k.park = NO;
  if (v == 3 && k.integ == PATH && k.stage == LARGE)
    return K_MEM;
  if (v == 4 && k.integ == PATH && k.stage == LARGE && k.mesh == NONE && k.mat != REFR)
    return k.mat == CHK ? K_POOL_MEM_CHK : K_POOL_MEM;
  if (v == 5 && k.stage == STREAM)
    k.stage = STAGED;
  if (v == 7)
    k.fit = NO;
  return -1;
}
#endif

int pt_pick_kernel(const PtSceneView &scene, const PtPickFacts &f)
{
  PtPickKey k = pt_classify(scene, f);
#ifdef PT_DEV_KERNELS
  {
    const int dev = pt_dev_pick(k);
    if (dev >= 0)
      return dev;
  }
#endif
  for (const PtPickRow &r : pt_pick_table)
    if (pt_row_matches(r, k))
      return r.kernel;
  return -1; /* unreachable: the table's last rows of every (integ, stage, mesh) block match ANY of the rest -- pinned by tests/test_pick_table.py */
}

const char *pt_kernel_name_of(int which) { return which >= 0 && which < K_COUNT ? pt_kernels[which].name : ""; }
int pt_kernel_count(void) { return K_COUNT; }
bool pt_kernel_uses_pend_pool(int which) { return pt_kernels[which].pend_pool; }
bool pt_kernel_is_queued(int which) { return pt_kernels[which].queued; }
bool pt_kernel_takes_chunks(int which) { return pt_kernels[which].chunks; }
bool pt_kernel_is_windowed(int which) { return pt_kernels[which].windowed; }
uint32_t pt_kernel_pend_columns_of(int which) { return pt_kernels[which].wide_pend ? 4u * 512u : PT_PEND_COLUMNS; }

/* launches per family member in this process (rt_hip_kernel_launches): what a test run actually exercised */
static std::atomic<unsigned long long> pt_launch_counts[K_COUNT];
unsigned long long pt_kernel_launches(int which) { return which >= 0 && which < K_COUNT ? pt_launch_counts[which].load() : 0ull; }

/* slots per XCD a pool must offer so that every resident workgroup of the kernels that take one finds a slot: CUs per XCD x
 * the most workgroups of any such kernel a CU holds (occupancy without dynamic LDS: an upper bound), + 25 %.  Until round 5
 * these were constants sized for 32 CUs x 4 workgroups with no slack (round-4 advisor finding). */
uint32_t pt_pool_slots_per_xcd(bool park_pool)
{
#ifdef PT_DEV_KERNELS
  { /* development builds: RT_HIP_POOL_SLOTS=n fixes both pools at n slots per XCD (n = 1: acquisition fails for all but eight workgroups) */
    const char *e = getenv("RT_HIP_POOL_SLOTS");
    const unsigned long n = e ? strtoul(e, nullptr, 10) : 0ul;
    if (n >= 1ul && n <= 4096ul)
      return (uint32_t)n;
  }
#endif
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int most = 1;
  for (int k = 0; k < K_COUNT; k++)
  {
    if (!(park_pool ? pt_kernels[k].queued : pt_kernels[k].pend_pool))
      continue;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(pt_kernels[k].fn), PT_BLOCK, 0) == hipSuccess)
      most = max(most, n);
    else
      (void)hipGetLastError();
  }
  if (most > 8)
    most = 8; /* 2,048 threads per CU */
  const uint32_t per_xcd = ((uint32_t)cus + PT_PARK_XCDS - 1u) / PT_PARK_XCDS;
  const uint32_t want = per_xcd * (uint32_t)most;
  return ((want + want / 4u) + 31u) & ~31u;
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
  /* the pre-test table behind the pair table: in scan order for small scenes (staged in LDS with the pairs), in the
   * hierarchy's leaf order for large meshes (read from HBM at the leaves) */
  if (scene.n_triangles != 0 && (pt_filter_in_lds(scene) || n_nodes != 0))
    hipLaunchKernelGGL(pt_build_tri32, dim3(min(1024u, (scene.n_triangles + 255u) / 256u)), dim3(256), 0, stream,
                       pt_filter_in_lds(scene) ? scene.tri_geom : scene.tri_geom_leaf, scene.n_triangles, near_R,
                       filt + 2 * (size_t)pt_filt_pair_slots(n_entries));
  return hipGetLastError();
}

hipError_t pt_launch_render(const PtLaunch &launch, hipStream_t stream, int which)
{
  if (which < 0 || which >= K_COUNT)
    return hipErrorInvalidValue;
  size_t extra_lds = 0;
#ifdef PT_DEV_KERNELS
  /* development knob: RT_HIP_EXTRA_LDS=<bytes> of unused dynamic LDS per workgroup, to measure how a
   * kernel responds to fewer resident workgroups per CU */
  static const size_t extra_lds_env = [] {
    const char *e = getenv("RT_HIP_EXTRA_LDS");
    return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)0;
  }();
  extra_lds = extra_lds_env;
#endif
  size_t lds_bytes = pt_render_lds_bytes(launch.scene) + extra_lds;
  const PtKernelInfo &k = pt_kernels[which];
  const PtKernelFn kernel = k.fn;
  if (k.stages_none)
    lds_bytes = extra_lds; /* the in-memory pooled kernels stage nothing, whatever the scene's size */
  if (k.pend_pool && (launch.pend_ws == nullptr || launch.pend_entries < (uint32_t)launch.max_depth + 2u ||
                      launch.pend_slot_doubles < (uint64_t)launch.pend_entries * PT_PEND_FIELDS_HOST * pt_kernel_pend_columns_of(which)))
    return hipErrorInvalidValue; /* a kernel with a pending-ray stack needs its pool, wide enough (rt_hip_shim.hip: pend_pool_for) */
  const bool queued = k.queued;
  if (queued && (launch.park_ws == nullptr || launch.park_slots_per_xcd == 0u))
    return hipErrorInvalidValue; /* the parked-walk kernels never run without their workspace (pt_pick_kernel: park) */
  if (launch.sample_chunks > 1 && !k.chunks)
    return hipErrorInvalidValue;
  if (queued) /* the spheres' filter pairs (staged forms), then per-lane traversal stacks (24-bit entries) sized by the tree, after the staged scene */
    lds_bytes += (k.stages_none ? (size_t)0 : (size_t)pt_filt_pair_slots(launch.scene.n_spheres) * 8u) +
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
    if ((launch.acc_windows != 0u) != k.windowed || launch.acc_ws == nullptr)
      return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(launch.acc_ws, 0, (size_t)launch.tile_count * (k.windowed ? PT_ACC_WS_WORDS_WIN : PT_ACC_WS_WORDS) * sizeof(unsigned long long), stream);
    if (e != hipSuccess)
      return e;
  }
  /* the parked-walk kernels render a tile per wave, four work units per workgroup */
  const uint32_t n_units = launch.tile_count * launch.sample_chunks;
  hipLaunchKernelGGL(kernel, dim3(queued ? (n_units + PT_BLOCK / 64 - 1) / (PT_BLOCK / 64) : n_units), dim3(PT_BLOCK), lds_bytes,
                     stream, launch);
  if (launch.sample_chunks > 1)
    hipLaunchKernelGGL(pt_resolve_tiles, dim3(launch.tile_count), dim3(PT_BLOCK), 0, stream, launch);
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    pt_launch_counts[which].fetch_add(1ull);
  return e;
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
