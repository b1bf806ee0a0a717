/* oracle/ref_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Builds the reference's own hot path (gue-ni/raytracer.c) from the sources
 * where they lie under /root/reference -- nothing is copied into this repo --
 * into oracle/_ref/libref_oracle_d<DEPTH>.so.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load the result.
 *
 * Technique: this TU includes the unmodified
 * raytracer.h, optionally re-defines MAX_DEPTH (a bare macro, raytracer.h:25),
 * then includes the unmodified raytracer.c, with `rand` macro-replaced by a
 * thread-local stream so that the reference's random_double() (raytracer.c:227)
 * draws from include/rt_rng.h.  Every tracing / intersection / shading
 * instruction executed is the reference's compiled code; the only restated
 * lines are the 12-line pixel loop of render() (raytracer.c:197-221), because
 * render() has no per-pixel hook through which to re-seed.  That restatement
 * is itself checked: with libc rand() left in place it reproduces render()'s
 * framebuffer byte for byte (ref_render_loop_libc vs ref_render_as_shipped,
 * tests/test_oracle_ref.py).
 *
 * Build: oracle/Makefile (gcc --std=c99 -fopenmp -O3, the reference's flags,
 * Makefile:2 there; -O3 is mandatory because vector.h uses plain C99 `inline`).
 */
#ifndef _DEFAULT_SOURCE
#define _DEFAULT_SOURCE
#endif
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include "../include/rt_rng.h" /* by path: -I must point ONLY at the reference */

/* libc rand captured before the macro below hides it */
static int (*const harness_libc_rand)(void) = rand;

static __thread uint64_t harness_state;
static __thread long long harness_draws;
static int harness_use_libc = 0;

static int harness_rand(void)
{
  harness_draws++;
  if (harness_use_libc)
    return harness_libc_rand();
  return (int)rt_rng_next31(&harness_state);
}

#define rand harness_rand
#include "raytracer.h" /* the reference's, via -I/root/reference */
#ifdef ORACLE_MAX_DEPTH
#undef MAX_DEPTH
#define MAX_DEPTH ORACLE_MAX_DEPTH
#endif

#ifdef ORACLE_MESH_HOOK
/* ---- mesh-capable build (oracle/_ref/libref_mesh_d<N>.so) ---------------------------------
 *
 * The reference's live intersect() scans spheres only; its triangle-mesh branch survives as a
 * comment block against a retired Object layout (raytracer.c:414-455).  This build REVIVES that
 * block around the reference's own COMPILED primitives -- intersect_sphere (:77-118),
 * intersect_triangle (:120-174), calculate_surface_normal (:42-45), point_at (:257) -- and makes
 * the reference's compiled trace_path() / cast_ray() call it, so whole samples and frames of
 * mesh scenes come out of reference code in composition, not only primitive by primitive.
 *
 * How the calls are redirected without touching the reference's files: `intersect` appears as
 * a token exactly five times in raytracer.c once comments are stripped -- the static prototype
 * (:30), the definition (:393), and the calls in trace_path (:487), cast_ray (:561) and cast_ray's
 * shadow ray (:572).  The macro below numbers the occurrences with __COUNTER__, so the definition
 * becomes harness_isect_2 (still compiled, still callable: ref_intersect_scene uses it to check
 * the revived sphere branch against it) and the three calls become harness_isect_3/4/5, which
 * are defined here.  If the reference ever gains or loses an occurrence the build breaks (an
 * undefined harness_isect_6, or an unused-static error below), it cannot silently mis-bind.
 *
 * The revived loop, literally: the dead code switches per object on its type; the Object layout
 * of today has no type field, so the harness keeps the scene as n_spheres Objects followed by one
 * Object per mesh (flags / color / emission live there, which is where trace_path :493-504 looks
 * them up by hit.object_id) and a parallel array of the meshes' vertex lists.  Everything
 * between the braces of the two `if`s is the reference's text (:404-411 live, :426-432 dead),
 * with `objects[i].geometry.mesh` spelled as the harness's array.
 *
 * One consequence of taking the block literally: intersect_triangle() writes the hit's texture
 * coordinates into `local` (:165-166) whenever it returns true, BEFORE the caller's
 * `local.t < min_t` test, and the block does not restore them.  So after the scan hit.u / hit.v
 * are those of the LAST triangle in scan order that the ray passes at all (t > EPSILON), not the
 * closest one's -- and not the closest sphere's either when a triangle lies behind it.  (hit.t is
 * stale in the same way already in the live sphere scan.)  Only M_CHECKERED materials read
 * u / v.  ref_intersect_mesh_scene reports both the literal values and the winner's own. */
#if __COUNTER__ != 0
#error "the occurrence numbering of `intersect` needs __COUNTER__ to start at 0 here"
#endif
typedef struct
{
  uint flags;
  vec3 color, emission;
  TriangleMesh mesh;
} HarnessMesh; /* = MeshObject of include/raytracer.h (layout checked in tests/test_oracle_ref.py) */

static size_t harness_n_spheres = 0;      /* objects[0 .. n_spheres) are spheres, the rest meshes */
static const HarnessMesh *harness_meshes = NULL;
/* the winner's own texture coordinates, captured at its update (see above) */
static __thread double harness_win_u, harness_win_v, harness_min_t;

static bool harness_mesh_intersect(const Ray *ray, Object *objects, size_t n, Hit *hit)
{
  /* raytracer.c:395-399 */
  double old_t = hit != NULL ? hit->t : DBL_MAX;
  double min_t = old_t;

  Hit local = {.t = DBL_MAX};

  for (uint i = 0; i < n; i++)
  {
    if (i < harness_n_spheres)
    {
      /* case GEOMETRY_SPHERE -- the live text, raytracer.c:404-412 */
      if (intersect_sphere(ray, objects[i].center, objects[i].radius, &local) && local.t < min_t)
      {
        min_t = local.t;
        local.object_id = i;
        local.point = point_at(ray, local.t);
        local.normal = vec3_normalize(vec3_sub(local.point, objects[i].center));
        local.u = atan2(local.normal.x, local.normal.z) / (2 * PI) + 0.5;
        local.v = local.normal.y * 0.5 + 0.5;
        harness_win_u = local.u;
        harness_win_v = local.v;
      }
    }
    else
    {
      /* case GEOMETRY_MESH -- the comment block, raytracer.c:419-436 */
      const TriangleMesh *mesh = &harness_meshes[i - harness_n_spheres].mesh;
      for (uint ti = 0; ti < mesh->num_triangles; ti++)
      {
        Vertex v0 = mesh->vertices[(ti * 3) + 0];
        Vertex v1 = mesh->vertices[(ti * 3) + 1];
        Vertex v2 = mesh->vertices[(ti * 3) + 2];

        if (intersect_triangle(ray, v0, v1, v2, &local) && local.t < min_t)
        {
          min_t = local.t;
          local.object_id = i;
          local.point = point_at(ray, local.t);
          local.normal = calculate_surface_normal(v0.pos, v1.pos, v2.pos);
          harness_win_u = local.u;
          harness_win_v = local.v;
        }
      }
    }
  }
  harness_min_t = min_t;

  /* raytracer.c:458-463 */
  if (hit != NULL)
  {
    memcpy(hit, &local, sizeof(*hit));
  }

  return min_t < old_t;
}

/* occurrences 3, 4, 5: the calls in trace_path :487, cast_ray :561 and :572 */
static bool harness_isect_3(const Ray *r, Object *o, size_t n, Hit *h) { return harness_mesh_intersect(r, o, n, h); }
static bool harness_isect_4(const Ray *r, Object *o, size_t n, Hit *h) { return harness_mesh_intersect(r, o, n, h); }
static bool harness_isect_5(const Ray *r, Object *o, size_t n, Hit *h) { return harness_mesh_intersect(r, o, n, h); }
#define HARNESS_CAT2(a, b) a##b
#define HARNESS_CAT(a, b) HARNESS_CAT2(a, b)
#define intersect HARNESS_CAT(harness_isect_, __COUNTER__)
#endif /* ORACLE_MESH_HOOK */

#include "raytracer.c" /* the reference's, unmodified */
#undef rand
#ifdef ORACLE_MESH_HOOK
#undef intersect
#if __COUNTER__ != 6
#error "raytracer.c no longer has exactly five occurrences of `intersect`: re-derive the hook numbering"
#endif
#define intersect harness_isect_2 /* the reference's compiled live (sphere-only) scan, raytracer.c:393-464 */
#endif

/* vector.h's plain-inline functions need one external definition each in case
 * the optimiser declines to inline a call (C99 6.7.4p7). */
extern inline vec3 vec3_mult(vec3 a, vec3 b);
extern inline vec3 vec3_sub(vec3 a, vec3 b);
extern inline vec3 vec3_add(vec3 a, vec3 b);
extern inline REAL vec3_dot(vec3 a, vec3 b);
extern inline REAL vec3_length(vec3 v);
extern inline vec3 vec3_scalar_mult(vec3 v, REAL s);
extern inline vec3 vec3_scalar_div(vec3 v, REAL s);
extern inline vec2 vec2_scalar_mult(vec2 v, REAL s);
extern inline vec2 vec2_add(vec2 a, vec2 b);
extern inline vec3 vec3_cross(vec3 a, vec3 b);
extern inline int vec3_equal(vec3 a, vec3 b);
extern inline vec3 vec3_normalize(vec3 v);
extern inline vec3 mat4_vector_mult(mat4 A, vec3 v);
extern inline void mat4_mult(mat4 A, mat4 B, mat4 C);

/* ---- introspection -------------------------------------------------------- */

int ref_max_depth(void) { return MAX_DEPTH; }

/* OpenMP team size of the as-shipped render() (raytracer.c:184); 1 = deterministic */
void ref_set_threads(int n) { omp_set_num_threads(n); }

/* sizeof / offsetof of the boundary structs, for the layout tests */
void ref_layout(uint64_t out[16])
{
  out[0] = sizeof(Object);
  out[1] = offsetof(Object, flags);
  out[2] = offsetof(Object, radius);
  out[3] = offsetof(Object, center);
  out[4] = offsetof(Object, color);
  out[5] = offsetof(Object, emission);
  out[6] = sizeof(Camera);
  out[7] = sizeof(Options);
  out[8] = sizeof(Ray);
  out[9] = sizeof(Hit);
  out[10] = sizeof(Vertex);
  out[11] = sizeof(vec3);
  out[12] = offsetof(Options, width);
  out[13] = offsetof(Options, height);
  out[14] = offsetof(Options, samples);
  out[15] = sizeof(TriangleMesh);
}

/* ---- primitive known-answer wrappers (flat double arrays in / out) -------- */

static vec3 v3(const double *p) { return (vec3){p[0], p[1], p[2]}; }
static void put3(double *o, vec3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

void ref_init_camera(Camera *cam, const double pos[3], const double target[3], int w, int h)
{
  Options o;
  memset(&o, 0, sizeof o);
  o.width = w;
  o.height = h;
  init_camera(cam, v3(pos), v3(target), &o);
}

void ref_camera_ray(const Camera *cam, double u, double v, double out[6])
{
  Ray r = get_camera_ray(cam, u, v);
  put3(out, r.origin);
  put3(out + 3, r.direction);
}

int ref_intersect_sphere(const double ray[6], const double center[3], double radius, double *t)
{
  Ray r = {v3(ray), v3(ray + 3)};
  Hit h = {.t = DBL_MAX};
  int ok = intersect_sphere(&r, v3(center), radius, &h);
  *t = h.t;
  return ok;
}

/* verts: 3 x (pos xyz, tex st) = 15 doubles; out: t, u, v as written to Hit */
int ref_intersect_triangle(const double ray[6], const double verts[15], double out[3])
{
  Ray r = {v3(ray), v3(ray + 3)};
  Vertex a = {v3(verts), {verts[3], verts[4]}};
  Vertex b = {v3(verts + 5), {verts[8], verts[9]}};
  Vertex c = {v3(verts + 10), {verts[13], verts[14]}};
  Hit h = {.t = DBL_MAX, .u = 0, .v = 0};
  int ok = intersect_triangle(&r, a, b, c, &h);
  out[0] = h.t;
  out[1] = h.u;
  out[2] = h.v;
  return ok;
}

void ref_surface_normal(const double v[9], double out[3])
{
  put3(out, calculate_surface_normal(v3(v), v3(v + 3), v3(v + 6)));
}

void ref_cross(const double a[3], const double b[3], double out[3])
{
  put3(out, vec3_cross(v3(a), v3(b)));
}

void ref_reflect(const double in[3], const double n[3], double out[3])
{
  put3(out, reflect(v3(in), v3(n)));
}

void ref_refract(const double in[3], const double n[3], double iot, double out[3])
{
  put3(out, refract(v3(in), v3(n), iot));
}

void ref_checkered(const double color[3], double u, double v, double m, double out[3])
{
  put3(out, checkered_texture(v3(color), u, v, m));
}

/* closest hit of the live sphere scan (raytracer.c:393-464) */
int ref_intersect_scene(const double ray[6], Object *objs, size_t n, double out_pn[6],
                        double out_tuv[3], uint32_t *id)
{
  Ray r = {v3(ray), v3(ray + 3)};
  Hit h = {.t = DBL_MAX};
  int ok = intersect(&r, objs, n, &h);
  put3(out_pn, h.point);
  put3(out_pn + 3, h.normal);
  out_tuv[0] = h.t;
  out_tuv[1] = h.u;
  out_tuv[2] = h.v;
  *id = h.object_id;
  return ok;
}

/* first `count` draws of the (seed, pixel, sample) stream as random_double() */
void ref_random_doubles(uint64_t seed, uint32_t pixel, uint32_t sample, int count, double *out)
{
  harness_use_libc = 0;
  harness_state = rt_rng_seed(seed, pixel, sample);
  for (int k = 0; k < count; k++)
    out[k] = random_double();
}

/* ---- one sample: raytracer.c:203-208 with the stream re-seeded ------------ */

/* Which side of render()'s `#if 1` (raytracer.c:207-211) the harness takes: 0 = trace_path
 * (as shipped), 1 = cast_ray, the Whitted integrator the reference keeps compiled but
 * unreferenced (raytracer.c:556-641). */
static int harness_integrator = 0;
void ref_set_integrator(int which) { harness_integrator = which; }

static vec3 harness_sample(Object *objs, size_t n, Camera *cam, int w, int h, uint32_t x,
                           uint32_t y, uint32_t s, uint64_t seed)
{
  harness_state = rt_rng_seed(seed, y * (uint32_t)w + x, s);
  double u = (double)(x + random_double()) / ((double)w - 1.0);
  double v = (double)(y + random_double()) / ((double)h - 1.0);
  Ray ray = get_camera_ray(cam, u, v);
  if (harness_integrator == 1)
    return cast_ray(&ray, objs, n, 0);
  return trace_path(&ray, objs, n, 0);
}

/* out: rgb[3]; stats: rays, tests, draws */
void ref_trace_sample(Object *objs, size_t n, Camera *cam, int w, int h, uint32_t x, uint32_t y,
                      uint32_t s, uint64_t seed, double out_rgb[3], long long stats[3])
{
  harness_use_libc = 0;
  ray_count = 0;
  intersection_test_count = 0;
  harness_draws = 0;
  put3(out_rgb, harness_sample(objs, n, cam, w, h, x, y, s, seed));
  stats[0] = ray_count;
  stats[1] = intersection_test_count;
  stats[2] = harness_draws;
}

/* ---- pixel loop: raytracer.c:197-221 restated, per-(pixel,sample) streams - */

/* pixels: `npix` linear indices y*w+x (NULL = 0..npix-1).  out_mean: npix*3
 * linear fp64 means (may be NULL); out_rgb8: npix*3 tonemapped bytes (may be
 * NULL).  stats: rays (all trace_path calls), primitive tests. */
void ref_render_pixels(Object *objs, size_t n, Camera *cam, int w, int h, int spp, uint64_t seed,
                       const uint32_t *pixels, size_t npix, double *out_mean, uint8_t *out_rgb8,
                       long long stats[2])
{
  const double gamma = 5.0;
  harness_use_libc = 0;
  ray_count = 0;
  intersection_test_count = 0;
  for (size_t k = 0; k < npix; k++)
  {
    uint32_t p = pixels ? pixels[k] : (uint32_t)k;
    uint32_t x = p % (uint32_t)w, y = p / (uint32_t)w;
    vec3 pixel = {0, 0, 0};
    for (uint32_t s = 0; s < (uint32_t)spp; s++)
      pixel = vec3_add(pixel, harness_sample(objs, n, cam, w, h, x, y, s, seed));
    pixel = vec3_scalar_mult(pixel, 1.0 / (double)spp);
    if (out_mean)
      put3(out_mean + 3 * k, pixel);
    if (out_rgb8)
    {
      out_rgb8[3 * k + 0] = (uint8_t)(255.0 * CLAMP(pow(pixel.x, 1 / gamma)));
      out_rgb8[3 * k + 1] = (uint8_t)(255.0 * CLAMP(pow(pixel.y, 1 / gamma)));
      out_rgb8[3 * k + 2] = (uint8_t)(255.0 * CLAMP(pow(pixel.z, 1 / gamma)));
    }
  }
  stats[0] = ray_count;
  stats[1] = intersection_test_count;
}

/* ---- as-shipped paths, libc rand(): the reference's own render() ---------- */

/* CPU-A/CPU-B of BASELINE.md: render() exactly as shipped. Thread count is
 * whatever OMP_NUM_THREADS says (1 = its deterministic best case). */
void ref_render_as_shipped(uint8_t *fb, Object *objs, size_t n, Camera *cam, int w, int h, int spp,
                           unsigned libc_seed, long long stats[2])
{
  Options o;
  memset(&o, 0, sizeof o);
  o.width = w;
  o.height = h;
  o.samples = spp;
  harness_use_libc = 1;
  srand(libc_seed);
  ray_count = 0;
  intersection_test_count = 0;
  render(fb, objs, n, cam, &o);
  stats[0] = ray_count;
  stats[1] = intersection_test_count;
  harness_use_libc = 0;
}

/* The restated pixel loop driven by the same libc stream: must reproduce
 * ref_render_as_shipped byte for byte at 1 thread (SURVEY 8c "Check 1"). */
void ref_render_loop_libc(uint8_t *fb, Object *objs, size_t n, Camera *cam, int w, int h, int spp,
                          unsigned libc_seed, long long stats[2])
{
  const double gamma = 5.0;
  harness_use_libc = 1;
  srand(libc_seed);
  ray_count = 0;
  intersection_test_count = 0;
  for (uint32_t y = 0; y < (uint32_t)h; y++)
    for (uint32_t x = 0; x < (uint32_t)w; x++)
    {
      vec3 pixel = {0, 0, 0};
      for (uint32_t s = 0; s < (uint32_t)spp; s++)
      {
        double u = (double)(x + random_double()) / ((double)w - 1.0);
        double v = (double)(y + random_double()) / ((double)h - 1.0);
        Ray ray = get_camera_ray(cam, u, v);
        pixel = vec3_add(pixel, trace_path(&ray, objs, n, 0));
      }
      pixel = vec3_scalar_mult(pixel, 1.0 / (double)spp);
      uint32_t k = (y * (uint32_t)w + x) * 3;
      fb[k + 0] = (uint8_t)(255.0 * CLAMP(pow(pixel.x, 1 / gamma)));
      fb[k + 1] = (uint8_t)(255.0 * CLAMP(pow(pixel.y, 1 / gamma)));
      fb[k + 2] = (uint8_t)(255.0 * CLAMP(pow(pixel.z, 1 / gamma)));
    }
  stats[0] = ray_count;
  stats[1] = intersection_test_count;
  harness_use_libc = 0;
}

#ifdef ORACLE_MESH_HOOK
/* ---- mesh scenes through the reference's compiled trace_path() / cast_ray() -------------
 * (libref_mesh_d<N>.so only.)  The scene arrives as the boundary's own arrays: n_spheres
 * Objects and n_meshes MeshObjects (include/raytracer.h); the harness lays them out as the
 * revived scan expects (see the top of this file): one Object array, spheres first, then one
 * Object per mesh carrying its flags / color / emission. */

static Object *harness_bind_scene(const Object *spheres, size_t n_spheres, const HarnessMesh *meshes, size_t n_meshes)
{
  Object *all = calloc(n_spheres + n_meshes + 1, sizeof(Object));
  if (n_spheres)
    memcpy(all, spheres, n_spheres * sizeof(Object));
  for (size_t m = 0; m < n_meshes; m++)
  {
    all[n_spheres + m].flags = meshes[m].flags;
    all[n_spheres + m].color = meshes[m].color;
    all[n_spheres + m].emission = meshes[m].emission;
  }
  harness_n_spheres = n_spheres;
  harness_meshes = meshes;
  return all;
}

/* One call of the revived scan.  out_pn: point, normal; out_tuv: Hit.t / .u / .v exactly as the
 * literal block leaves them (stale, see the top of this file); out_win: min_t and the winner's
 * own u, v; id: Hit.object_id (sphere index, or n_spheres + mesh index); tests: primitive tests
 * counted by the compiled primitives (intersection_test_count). */
int ref_intersect_mesh_scene(const double ray[6], const Object *spheres, size_t n_spheres, const HarnessMesh *meshes,
                             size_t n_meshes, double out_pn[6], double out_tuv[3], double out_win[3], uint32_t *id,
                             long long *tests)
{
  Object *all = harness_bind_scene(spheres, n_spheres, meshes, n_meshes);
  Ray r = {v3(ray), v3(ray + 3)};
  Hit h = {.t = DBL_MAX};
  intersection_test_count = 0;
  harness_win_u = harness_win_v = 0;
  int ok = harness_mesh_intersect(&r, all, n_spheres + n_meshes, &h);
  put3(out_pn, h.point);
  put3(out_pn + 3, h.normal);
  out_tuv[0] = h.t;
  out_tuv[1] = h.u;
  out_tuv[2] = h.v;
  out_win[0] = harness_min_t;
  out_win[1] = harness_win_u;
  out_win[2] = harness_win_v;
  *id = h.object_id;
  *tests = intersection_test_count;
  free(all);
  return ok;
}

void ref_mesh_trace_sample(const Object *spheres, size_t n_spheres, const HarnessMesh *meshes, size_t n_meshes,
                           Camera *cam, int w, int h, uint32_t x, uint32_t y, uint32_t s, uint64_t seed,
                           double out_rgb[3], long long stats[3])
{
  Object *all = harness_bind_scene(spheres, n_spheres, meshes, n_meshes);
  ref_trace_sample(all, n_spheres + n_meshes, cam, w, h, x, y, s, seed, out_rgb, stats);
  free(all);
}

void ref_mesh_render_pixels(const Object *spheres, size_t n_spheres, const HarnessMesh *meshes, size_t n_meshes,
                            Camera *cam, int w, int h, int spp, uint64_t seed, const uint32_t *pixels, size_t npix,
                            double *out_mean, uint8_t *out_rgb8, long long stats[2])
{
  Object *all = harness_bind_scene(spheres, n_spheres, meshes, n_meshes);
  ref_render_pixels(all, n_spheres + n_meshes, cam, w, h, spp, seed, pixels, npix, out_mean, out_rgb8, stats);
  free(all);
}

int ref_mesh_hook(void) { return 1; }
uint64_t ref_mesh_layout(void) { return sizeof(HarnessMesh); }
#endif /* ORACLE_MESH_HOOK */
