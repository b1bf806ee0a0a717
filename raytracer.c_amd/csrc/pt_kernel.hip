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
 * Mapping (one workgroup = one 8x8 pixel tile):
 *   256 threads = 4 wavefronts.  Wavefront w owns tile rows 2w, 2w+1 (16
 *   pixels); lane l of it works on pixel (l >> 2) of those 16 and on sample
 *   slice (l & 3): samples s = slice, slice+4, ... of that pixel.  Each lane
 *   runs a FLATTENED loop -- one iteration = one trace_path() call of the
 *   reference; a lane whose path ends starts its next sample in the same
 *   iteration slot -- so all 64 lanes stay busy in the scene scan, which is
 *   wave-uniform (every lane tests the same primitive, read from LDS as a
 *   broadcast).  The four slice partial sums of a pixel are combined with two
 *   xor-shuffles in a fixed order, the tile is staged in LDS and leaves as one
 *   fully coalesced 768-byte float3 store (plus 192 tonemapped bytes).
 *
 * Numerics: everything on the decision path (hit / miss, closest index,
 * Russian roulette, rejection sampling, hemisphere flip) is fp64 in exactly
 * the reference's operation order, compiled with -ffp-contract=off, IEEE
 * sqrt and division -- so every branch decision, hence every RNG draw and
 * the ray / test counters, equals the CPU reference's bit for bit.  Only the
 * radiance VALUE is accumulated differently (forward: L += T*e; T *= albedo*cos
 * instead of the recursive nesting), a ~1e-16 relative difference.
 *
 * No MFMA: this is branchy fp64 scalar-per-lane math with no dense
 * contraction.  The bounding roof is the fp64 VALU issue rate.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

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

/* raytracer.c:227 */
__device__ __forceinline__ double rnd(uint64_t &state) { return rt_rng_double(&state); }

/* raytracer.c:218-220 */
__device__ __forceinline__ uint8_t tonemap(double x)
{
  double g = pow(x, 1 / 5.0);
  double lo = (g < 1) ? g : 1.0; /* MIN(x, 1): NaN -> 1 */
  double cl = (0 > lo) ? 0.0 : lo; /* MAX(0, .) */
  return (uint8_t)(255.0 * cl);
}

constexpr double kEps = 1e-8;       /* raytracer.h:24 */
constexpr double kBg = 10 / 255.0;  /* raytracer.h:46 BACKGROUND */
constexpr double kPi = 3.14159265359; /* raytracer.h:22 */

} // namespace

extern "C" __global__ __launch_bounds__(PT_BLOCK) void pt_render_tiles(const PtLaunch L)
{
  extern __shared__ double lds[];
  __shared__ float out_f[PT_TILE_PIXELS * 3];
  __shared__ uint8_t out_b[PT_TILE_PIXELS * 3 + 64];
  __shared__ unsigned long long wg_stats[2];

  const PtSceneView &sc = L.scene;
  const uint32_t n_sph = sc.n_spheres;
  const uint32_t n_mat = sc.n_spheres + sc.n_meshes;
  const uint32_t n_tri = sc.n_triangles;
  const bool tris_in_lds = n_tri <= PT_MAX_LDS_TRIS;

  double *geom = lds;                               /* n_sph x 4 */
  double *mat = geom + 4 * (size_t)n_sph;           /* n_mat x 8 */
  double *tri = mat + PT_MAT_STRIDE * (size_t)n_mat; /* n_tri x 9 when staged */

  /* ---- stage the scene in LDS (once per workgroup) ---- */
  for (uint32_t k = threadIdx.x; k < 4 * n_sph; k += PT_BLOCK)
    geom[k] = sc.sphere_geom[k];
  for (uint32_t k = threadIdx.x; k < PT_MAT_STRIDE * n_mat; k += PT_BLOCK)
    mat[k] = sc.material[k];
  if (tris_in_lds)
    for (uint32_t k = threadIdx.x; k < 9 * n_tri; k += PT_BLOCK)
      tri[k] = sc.tri_geom[k];
  if (threadIdx.x < 2)
    wg_stats[threadIdx.x] = 0;
  __syncthreads();
  const double *tri_src = tris_in_lds ? tri : sc.tri_geom;

  /* ---- which pixel / slice am I ---- */
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t pix_in_tile = wave * 16u + (lane >> 2);
  const uint32_t slice = lane & (PT_SLICES - 1);
  const uint32_t tile = L.tile_first + blockIdx.x * L.tile_stride;
  const uint32_t px = (tile % L.tiles_x) * PT_TILE + (pix_in_tile & 7u);
  const uint32_t py = (tile / L.tiles_x) * PT_TILE + (pix_in_tile >> 3);
  const bool inside = px < (uint32_t)L.width && py < (uint32_t)L.height;
  const uint32_t pixel = py * (uint32_t)L.width + px;
  const uint32_t S = (uint32_t)L.samples;

  const V3 cam_pos = ld3(L.cam.pos), cam_h = ld3(L.cam.horizontal);
  const V3 cam_v = ld3(L.cam.vertical), cam_llc = ld3(L.cam.llc);
  const double inv_w = (double)L.width - 1.0, inv_h = (double)L.height - 1.0; /* divisors */

  V3 acc = {0, 0, 0};   /* sum of finished samples of this lane's slice */
  V3 Ls = {0, 0, 0};    /* radiance of the sample in flight */
  V3 T = {1, 1, 1};     /* its path throughput */
  V3 o = {0, 0, 0}, d = {0, 0, 1};
  uint64_t rng = 1;
  int depth = 0;
  uint32_t n_rays = 0, n_casts = 0;
  uint32_t s = inside ? slice : S;
  bool fresh = true;

  while (s < S)
  {
    if (fresh)
    {
      /* raytracer.c:203-206 + get_camera_ray :375-384 */
      rng = rt_rng_seed(L.seed, pixel, s);
      double u = ((double)px + rnd(rng)) / inv_w;
      double v = ((double)py + rnd(rng)) / inv_h;
      V3 on_plane = v_add(cam_llc, v_add(v_scale(cam_h, u), v_scale(cam_v, v)));
      o = cam_pos;
      d = v_normalize(v_sub(cam_pos, on_plane));
      T = {1, 1, 1};
      Ls = {0, 0, 0};
      depth = 0;
      fresh = false;
    }

    /* ---- one trace_path() call (raytracer.c:482) ---- */
    n_rays++;
    bool path_ends = true;
    V3 add = {kBg, kBg, kBg}; /* what this call contributes if the path ends here */

    if (depth <= L.max_depth)
    {
      n_casts++;
      /* ---- intersect(): closest hit, strict <, index order (:393-464) ---- */
      double min_t = 1.7976931348623157e308; /* DBL_MAX */
      int best = -1;
      for (uint32_t i = 0; i < n_sph; i++)
      {
        /* intersect_sphere :82-117 */
        const double *g = geom + 4 * i;
        V3 Lv = {g[0] - o.x, g[1] - o.y, g[2] - o.z};
        double tca = v_dot(Lv, d);
        double d2 = v_dot(Lv, Lv) - tca * tca;
        double r2 = g[3];
        if (!(tca < 0) && !(d2 > r2))
        {
          double thc = sqrt(r2 - d2);
          double t0 = tca - thc, t1 = tca + thc;
          if (t0 > t1)
          {
            double tmp = t0;
            t0 = t1;
            t1 = tmp;
          }
          if (t0 < 0)
            t0 = t1;
          if (t0 > kEps && t0 < min_t)
          {
            min_t = t0;
            best = (int)i;
          }
        }
      }
      double bary_u = 0, bary_v = 0;
      for (uint32_t i = 0; i < n_tri; i++)
      {
        /* intersect_triangle :132-150 (Moeller-Trumbore, two-sided) */
        const double *g = tri_src + 9 * (size_t)i;
        V3 v0 = ld3(g), e1 = ld3(g + 3), e2 = ld3(g + 6);
        V3 h = v_cross(d, e2);
        double a = v_dot(e1, h);
        if (!(a > -kEps && a < kEps))
        {
          double f = 1.0 / a;
          V3 sv = v_sub(o, v0);
          double u = f * v_dot(sv, h);
          if (!(u < 0.0 || u > 1.0))
          {
            V3 q = v_cross(sv, e1);
            double v = f * v_dot(d, q);
            if (!(v < 0.0 || u + v > 1.0))
            {
              double t = f * v_dot(e2, q);
              if (t > kEps && t < min_t)
              {
                min_t = t;
                best = (int)(n_sph + i);
                bary_u = u;
                bary_v = v;
              }
            }
          }
        }
      }

      if (best >= 0)
      {
        /* ---- the winner's hit record (:406-411 / :428-431) ---- */
        V3 p = v_add(o, v_scale(d, min_t)); /* point_at :257 */
        V3 n;
        uint32_t slot;
        double tex_u = 0, tex_v = 0;
        const bool is_tri = (uint32_t)best >= n_sph;
        if (!is_tri)
        {
          const double *g = geom + 4 * best;
          n = v_normalize(v_sub(p, ld3(g)));
          slot = (uint32_t)best;
        }
        else
        {
          const uint32_t ti = (uint32_t)best - n_sph;
          n = ld3(sc.tri_normal + 3 * (size_t)ti);
          slot = sc.tri_object[ti];
        }
        const double *m = mat + PT_MAT_STRIDE * slot;
        const double prob = m[0];
        V3 albedo = ld3(m + 1);
        const V3 emission = ld3(m + 4);
        const uint32_t flags = (uint32_t)__double_as_longlong(m[7]);

        add = emission; /* a path that dies in the roulette returns emission (:502) */
        /* russian roulette :497-502: the draw is always consumed */
        if (rnd(rng) < prob)
        {
          path_ends = false;
          if (flags & PT_FLAG_CHECKER)
          {
            if (!is_tri)
            {
              tex_u = atan2(n.x, n.z) / (2 * kPi) + 0.5; /* :410-411 */
              tex_v = n.y * 0.5 + 0.5;
            }
            else
            {
              /* :154-167 barycentric blend of the texture coordinates */
              const double *tx = sc.tri_tex + 6 * (size_t)((uint32_t)best - n_sph);
              double w0 = 1 - bary_u - bary_v;
              tex_u = (tx[0] * w0 + tx[2] * bary_u) + tx[4] * bary_v;
              tex_v = (tx[1] * w0 + tx[3] * bary_u) + tx[5] * bary_v;
            }
            /* checkered_texture :386-391, M = 100000 (:508) */
            double on = (double)((fmod(tex_u * 100000.0, 1.0) > 0.5) ^ (fmod(tex_v * 100000.0, 1.0) < 0.5));
            double c = 0.3 * (1 - on) + 0.7 * on;
            albedo = v_scale(albedo, c);
          }
          V3 nd;
          double weight = 1.0;
          if (flags & PT_FLAG_MIRROR)
          {
            /* reflect :349-352; direction left un-normalised (:542) */
            nd = v_sub(d, v_scale(n, 2 * v_dot(d, n)));
          }
          else
          {
            /* random_on_hemisphere :231-253: x, y, z drawn in that order */
            V3 q;
            double len;
            int tries = 0;
            do
            {
              q.x = rnd(rng) * 2.0 + -1.0;
              q.y = rnd(rng) * 2.0 + -1.0;
              q.z = rnd(rng) * 2.0 + -1.0;
              len = sqrt(v_dot(q, q));
            } while (len > 1 && ++tries < 100);
            nd = v_scale(q, 1.0 / len);
            if (v_dot(nd, n) < 0)
              nd = v_scale(nd, -1);
            weight = v_dot(nd, n); /* cos_theta :549 */
          }
          /* L = e + albedo (.) (L_next * cos)  ==>  forward form */
          Ls = v_add(Ls, v_mul(T, emission));
          T = v_mul(T, (flags & PT_FLAG_MIRROR) ? albedo : v_scale(albedo, weight));
          o = p;
          d = nd;
          depth++;
        }
      }
    }

    if (path_ends)
    {
      Ls = v_add(Ls, v_mul(T, add));
      acc = v_add(acc, Ls);
      s += PT_SLICES;
      fresh = true;
    }
  }

  /* ---- per-pixel mean: fixed-order reduction over the 4 slice lanes ---- */
  acc.x += __shfl_xor(acc.x, 1);
  acc.y += __shfl_xor(acc.y, 1);
  acc.z += __shfl_xor(acc.z, 1);
  acc.x += __shfl_xor(acc.x, 2);
  acc.y += __shfl_xor(acc.y, 2);
  acc.z += __shfl_xor(acc.z, 2);
  const V3 mean = v_scale(acc, 1.0 / (double)S); /* :215 */

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

  /* ---- coalesced tile store: 192 floats = 768 contiguous bytes ---- */
  if (threadIdx.x < PT_TILE_PIXELS * 3)
    L.tiles_rgb[(size_t)blockIdx.x * (PT_TILE_PIXELS * 3) + threadIdx.x] = out_f[threadIdx.x];
  if (L.tiles_rgb8 && threadIdx.x < PT_TILE_PIXELS * 3 / 4)
    reinterpret_cast<uint32_t *>(L.tiles_rgb8)[(size_t)blockIdx.x * (PT_TILE_PIXELS * 3 / 4) + threadIdx.x] =
        reinterpret_cast<const uint32_t *>(out_b)[threadIdx.x];
  if (L.stats && threadIdx.x == 0)
  {
    const unsigned long long rays = wg_stats[0], casts = wg_stats[1];
    atomicAdd(&L.stats[0], rays);
    atomicAdd(&L.stats[1], casts);
    atomicAdd(&L.stats[2], casts * (unsigned long long)(n_sph + n_tri));
  }
  if (L.stats && threadIdx.x == 64)
  {
    const uint32_t tx0 = (tile % L.tiles_x) * PT_TILE, ty0 = (tile / L.tiles_x) * PT_TILE;
    const uint32_t cw = min((uint32_t)PT_TILE, (uint32_t)L.width - tx0);
    const uint32_t ch = min((uint32_t)PT_TILE, (uint32_t)L.height - ty0);
    atomicAdd(&L.stats[3], (unsigned long long)cw * ch * S);
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
  size_t doubles = 4 * (size_t)sc.n_spheres + PT_MAT_STRIDE * (size_t)(sc.n_spheres + sc.n_meshes);
  if (sc.n_triangles <= PT_MAX_LDS_TRIS)
    doubles += 9 * (size_t)sc.n_triangles;
  return doubles * sizeof(double);
}

hipError_t pt_launch_render(const PtLaunch &launch, hipStream_t stream)
{
  const size_t lds_bytes = pt_render_lds_bytes(launch.scene);
  static size_t lds_allowed = 0; /* raised once per process if a scene needs > 64 KiB */
  if (lds_bytes > 64 * 1024 && lds_bytes > lds_allowed)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pt_render_tiles),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
      return e;
    lds_allowed = lds_bytes;
  }
  hipLaunchKernelGGL(pt_render_tiles, dim3(launch.tile_count), dim3(PT_BLOCK), lds_bytes, stream, launch);
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
