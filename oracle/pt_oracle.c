/* oracle/pt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's path-tracing hot path
 * (gue-ni/raytracer.c), written from SURVEY.md appendix A with each function
 * citing the reference file:line it follows.  It exists to CHECK the HIP
 * path: only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke()
 * may link, load or call it; nothing under raytracer.c_amd/ does.
 *
 * Pinning (tests/test_oracle_ref.py, tests/test_golden.py): bit-identical --
 * linear fp64 means, tonemapped bytes, ray / test / draw counts -- to the
 * reference's own compiled trace_path()/intersect() (oracle/_ref, built by
 * oracle/Makefile from /root/reference) on every sphere scene of
 * BASELINE.json's configs, and to the golden fixtures under tests/golden/
 * that the compiled reference generated.  The triangle-mesh scene scan has no
 * live caller in the reference (its intersect() mesh branch is commented out,
 * raytracer.c:414-455); it is pinned IN COMPOSITION against oracle/_ref/
 * libref_mesh_d<N>.so, where that comment block is revived around the
 * reference's compiled intersect_triangle / calculate_surface_normal / point_at
 * and called by its compiled trace_path() / cast_ray() (ref_harness.c,
 * ORACLE_MESH_HOOK): closest hits on random rays, per-sample traces and whole
 * frames of mesh scenes are bit-identical (tests/test_oracle_ref.py).
 *
 * Compile without FMA contraction (oracle/Makefile: -ffp-contract=off), as
 * the reference is (gcc --std=c99, no -march).
 */
#include <math.h>
#include "pt_oracle.h"
#include "rt_rng.h"

typedef struct
{
  const Object *objs;
  size_t n_objs;
  const MeshObject *meshes;
  size_t n_meshes;
  int max_depth;
  int whitted; /* 1: cast_ray (raytracer.c:556-641) instead of trace_path */
  uint64_t rng;
  long long rays, tests, casts, draws;
} Ctx;

typedef struct
{
  double t, u, v;
  vec3 point, normal;
  int is_mesh;   /* texture coords come from the triangle, not from atan2 */
  size_t index;  /* sphere index, or mesh index */
} Closest;

/* raytracer.c:227  random_double = rand() / (RAND_MAX + 1.0), RAND_MAX = 2^31-1 */
static double draw(Ctx *c)
{
  c->draws++;
  return (double)rt_rng_next31(&c->rng) / ((double)2147483647 + 1);
}

/* raytracer.c:229 */
static double draw_range(Ctx *c, double lo, double hi) { return draw(c) * (hi - lo) + lo; }

/* raytracer.c:77-118: geometric ray/sphere solution.  Returns 1 and *t on an
 * accepted hit. */
static int sphere_test(const Ray *ray, vec3 center, double radius, double *t)
{
  vec3 L = vec3_sub(center, ray->origin);
  double tca = vec3_dot(L, ray->direction);
  if (tca < 0)
    return 0;
  double d2 = vec3_dot(L, L) - tca * tca;
  double radius2 = radius * radius;
  if (d2 > radius2)
    return 0;
  double thc = sqrt(radius2 - d2);
  double t0 = tca - thc, t1 = tca + thc;
  if (t0 > t1)
  {
    double tmp = t0;
    t0 = t1;
    t1 = tmp;
  }
  if (t0 < 0)
  {
    t0 = t1;
    if (t0 < 0)
      return 0;
  }
  if (t0 > EPSILON)
  {
    *t = t0;
    return 1;
  }
  return 0;
}

/* raytracer.c:120-174: Moeller-Trumbore, two-sided; u,v out are the
 * barycentric-interpolated TEXTURE coordinates (:154-167). */
static int triangle_test(const Ray *ray, const Vertex *a, const Vertex *b, const Vertex *c,
                         double *t_out, double *u_out, double *v_out)
{
  vec3 edge1 = vec3_sub(b->pos, a->pos);
  vec3 edge2 = vec3_sub(c->pos, a->pos);
  vec3 h = vec3_cross(ray->direction, edge2);
  double det = vec3_dot(edge1, h);
  if (det > -EPSILON && det < EPSILON)
    return 0;
  double f = 1.0 / det;
  vec3 s = vec3_sub(ray->origin, a->pos);
  double u = f * vec3_dot(s, h);
  if (u < 0.0 || u > 1.0)
    return 0;
  vec3 q = vec3_cross(s, edge1);
  double v = f * vec3_dot(ray->direction, q);
  if (v < 0.0 || u + v > 1.0)
    return 0;
  double t = f * vec3_dot(edge2, q);
  if (!(t > EPSILON))
    return 0;
  vec2 tex = vec2_add(vec2_add(vec2_scalar_mult(a->tex, 1 - u - v), vec2_scalar_mult(b->tex, u)),
                      vec2_scalar_mult(c->tex, v));
  *t_out = t;
  *u_out = tex.x;
  *v_out = tex.y;
  return 1;
}

/* raytracer.c:42-45 (winding as written there: cross(v2-v0, v1-v0)) */
static vec3 surface_normal(vec3 v0, vec3 v1, vec3 v2)
{
  return vec3_normalize(vec3_cross(vec3_sub(v2, v0), vec3_sub(v1, v0)));
}

/* raytracer.c:257 */
static vec3 at(const Ray *ray, double t)
{
  return vec3_add(ray->origin, vec3_scalar_mult(ray->direction, t));
}

/* raytracer.c:393-464: linear scan, strict `<` so the first index wins ties.
 * Spheres first (live code :401-412), then meshes (comment block :417-435). */
static int closest_hit(Ctx *c, const Ray *ray, Closest *best)
{
  double min_t = DBL_MAX;
  int found = 0;
  c->casts++;
  for (size_t i = 0; i < c->n_objs; i++)
  {
    double t;
    c->tests++;
    if (sphere_test(ray, c->objs[i].center, c->objs[i].radius, &t) && t < min_t)
    {
      min_t = t;
      found = 1;
      best->t = t;
      best->is_mesh = 0;
      best->index = i;
      best->point = at(ray, t);
      best->normal = vec3_normalize(vec3_sub(best->point, c->objs[i].center));
    }
  }
  /* intersect_triangle() stores the texture coordinates into the caller's Hit whenever it
   * returns true (:165-166), i.e. BEFORE the caller's `local.t < min_t` test (:426), and the
   * scan never restores them: after the loop hit.u / hit.v are those of the LAST triangle in
   * scan order that the ray passes at t > EPSILON -- closest or not, and even when the closest
   * hit is a sphere (spheres are scanned first, and their u / v are written only on a closer
   * hit, :410-411).  Reproduced here; only M_CHECKERED materials read u / v. */
  int any_triangle = 0;
  double last_u = 0, last_v = 0;
  for (size_t m = 0; m < c->n_meshes; m++)
  {
    const TriangleMesh *mesh = &c->meshes[m].mesh;
    for (size_t ti = 0; ti < mesh->num_triangles; ti++)
    {
      const Vertex *a = &mesh->vertices[3 * ti], *b = a + 1, *d = a + 2;
      double t, u, v;
      c->tests++;
      if (triangle_test(ray, a, b, d, &t, &u, &v))
      {
        any_triangle = 1;
        last_u = u;
        last_v = v;
        if (t < min_t)
        {
          min_t = t;
          found = 1;
          best->t = t;
          best->is_mesh = 1;
          best->index = m;
          best->point = at(ray, t);
          best->normal = surface_normal(a->pos, b->pos, d->pos);
        }
      }
    }
  }
  if (any_triangle)
  {
    best->u = last_u;
    best->v = last_v;
  }
  else if (found)
  {
    /* raytracer.c:410-411, evaluated once for the winner (same value as
     * evaluating it at every provisional winner) */
    best->u = atan2(best->normal.x, best->normal.z) / (2 * PI) + 0.5;
    best->v = best->normal.y * 0.5 + 0.5;
  }
  return found;
}

/* raytracer.c:349-352 */
static vec3 reflect_dir(vec3 in, vec3 n)
{
  return vec3_sub(in, vec3_scalar_mult(n, 2 * vec3_dot(in, n)));
}

/* raytracer.c:354-373 with the CLAMP_BETWEEN quirk (raytracer.h:30): cosi is
 * the constant MAX(-1, MIN(1, 1)) = 1 whatever In and N are; sqrtf as there. */
static vec3 refract_dir(vec3 in, vec3 nrm, double iot)
{
  double cosi = 1;
  double etai = 1, etat = iot;
  vec3 n = nrm;
  if (cosi < 0)
    cosi = -cosi;
  else
  {
    double tmp = etai;
    etai = etat;
    etat = tmp;
    n = vec3_scalar_mult(nrm, -1);
  }
  double eta = etai / etat;
  double k = 1 - eta * eta * (1 - cosi * cosi);
  if (k < 0)
  {
    vec3 zero = {0 / 255.0, 0 / 255.0, 0 / 255.0};
    return zero;
  }
  return vec3_add(vec3_scalar_mult(in, eta), vec3_scalar_mult(n, eta * cosi - sqrtf((float)k)));
}

/* raytracer.c:386-391 */
static vec3 checker(vec3 color, double u, double v, double m)
{
  double on = (double)((fmod(u * m, 1.0) > 0.5) ^ (fmod(v * m, 1.0) < 0.5));
  double c = 0.3 * (1 - on) + 0.7 * on;
  return vec3_scalar_mult(color, c);
}

/* raytracer.c:231-253: rejection-sample the unit ball (x, y, z drawn in that
 * order), normalise, flip into the normal's hemisphere. */
static vec3 hemisphere_dir(Ctx *c, vec3 normal)
{
  vec3 p;
  int tries = 0;
  do
  {
    ++tries;
    assert(tries < 100);
    p.x = draw_range(c, -1, 1);
    p.y = draw_range(c, -1, 1);
    p.z = draw_range(c, -1, 1);
  } while (vec3_length(p) > 1);
  vec3 d = vec3_normalize(p);
  if (vec3_dot(d, normal) < 0)
    return vec3_scalar_mult(d, -1);
  return d;
}

/* raytracer.c:255 */
static double mix(double a, double b, double m) { return b * m + a * (1 - m); }

/* raytracer.c:482-554, recursive exactly as there so the radiance nesting
 * (and therefore every rounding) is the reference's. */
static vec3 trace(Ctx *c, const Ray *ray, int depth)
{
  const vec3 background = {10 / 255.0, 10 / 255.0, 10 / 255.0};
  Closest hit;
  c->rays++;
  if (depth > c->max_depth || !closest_hit(c, ray, &hit))
    return background;

  vec3 albedo, emission;
  uint flags;
  if (hit.is_mesh)
  {
    albedo = c->meshes[hit.index].color;
    emission = c->meshes[hit.index].emission;
    flags = c->meshes[hit.index].flags;
  }
  else
  {
    albedo = c->objs[hit.index].color;
    emission = c->objs[hit.index].emission;
    flags = c->objs[hit.index].flags;
  }

  /* russian roulette :497-502 -- the draw is consumed even when prob >= 1 */
  double prob = MAX(albedo.x, MAX(albedo.y, albedo.z));
  if (draw(c) < prob)
    albedo = vec3_scalar_mult(albedo, 1 / prob);
  else
    return emission;

  if (flags & M_CHECKERED)
    albedo = checker(albedo, hit.u, hit.v, 100000);

  Ray next;
  next.origin = hit.point;
  vec3 radiance;
  if (flags & M_REFRACTION)
  {
    /* :514-529 two children, the "refracted" one traced fully first */
    double facing = -vec3_dot(ray->direction, hit.normal);
    double fresnel = mix(pow(1 - facing, 3), 1, 0.1);
    double kr = fresnel, kt = (1 - fresnel) * 1.0;
    next.direction = vec3_normalize(refract_dir(vec3_scalar_mult(ray->direction, -1), hit.normal, 1.0));
    vec3 through = trace(c, &next, depth + 1);
    next.direction = vec3_normalize(reflect_dir(vec3_scalar_mult(ray->direction, 1), hit.normal));
    vec3 bounced = trace(c, &next, depth + 1);
    radiance = vec3_add(vec3_scalar_mult(through, kt), vec3_scalar_mult(bounced, kr));
  }
  else if (flags & M_REFLECTION)
  {
    next.direction = reflect_dir(ray->direction, hit.normal); /* :542 not normalised */
    radiance = trace(c, &next, depth + 1);
  }
  else
  {
    next.direction = hemisphere_dir(c, hit.normal);
    double cos_theta = vec3_dot(next.direction, hit.normal);
    radiance = vec3_scalar_mult(trace(c, &next, depth + 1), cos_theta);
  }
  return vec3_add(emission, vec3_mult(albedo, radiance));
}

/* raytracer.c:556-641: cast_ray, the Whitted integrator the reference keeps compiled behind
 * the `#if 1` of render() (:207-211).  One fixed point light, Phong terms with the light's
 * colour, a shadow ray without a distance limit, mirror and "refraction" children. */
static vec3 whitted(Ctx *c, const Ray *ray, int depth)
{
  const vec3 background = {10 / 255.0, 10 / 255.0, 10 / 255.0};
  const vec3 zero = {0 / 255.0, 0 / 255.0, 0 / 255.0};
  Closest hit;
  c->rays++;
  if (depth > c->max_depth || !closest_hit(c, ray, &hit))
    return background;

  vec3 out_color = zero;
  const vec3 light_pos = {2, 7, 2};
  const vec3 light_color = {1, 1, 1};

  Ray light_ray;
  light_ray.origin = hit.point;
  light_ray.direction = vec3_normalize(vec3_sub(light_pos, hit.point));
  /* :571 intersect(.., NULL): anything in front of the point, at any distance, shadows it */
  Closest blocker;
  int in_shadow = closest_hit(c, &light_ray, &blocker);

  vec3 object_color;
  uint flags;
  if (hit.is_mesh)
  {
    object_color = c->meshes[hit.index].color;
    flags = c->meshes[hit.index].flags;
  }
  else
  {
    object_color = c->objs[hit.index].color;
    flags = c->objs[hit.index].flags;
  }
  const double ka = 0.25, kd = 0.5, ks = 0.8, alpha = 10.0;
  if (flags & M_CHECKERED)
    object_color = checker(object_color, hit.u, hit.v, 10);

  vec3 ambient = vec3_scalar_mult(light_color, ka);
  vec3 diffuse = vec3_scalar_mult(light_color, kd * MAX(0.0, vec3_dot(hit.normal, light_ray.direction)));
  vec3 reflected = reflect_dir(light_ray.direction, hit.normal);
  vec3 view_dir = vec3_normalize(vec3_sub(hit.point, ray->origin));
  vec3 specular = vec3_scalar_mult(light_color, ks * pow(MAX(vec3_dot(view_dir, reflected), 0.0), alpha));
  vec3 surface = vec3_mult(
      vec3_add(ambient, vec3_scalar_mult(vec3_add(specular, diffuse), in_shadow ? 0 : 1)), object_color);

  vec3 reflection = zero, refraction = zero;
  double kr = 0, kt = 0;
  Ray next;
  next.origin = hit.point;
  if (flags & M_REFLECTION)
  {
    kr = 1.0;
    next.direction = vec3_normalize(reflect_dir(ray->direction, hit.normal));
    reflection = whitted(c, &next, depth + 1);
  }
  if (flags & M_REFRACTION)
  {
    double transparency = 0.5;
    double facing = -vec3_dot(ray->direction, hit.normal);
    double fresnel = mix(pow(1 - facing, 3), 1, 0.1);
    kr = fresnel;
    kt = (1 - fresnel) * transparency;
    next.direction = vec3_normalize(refract_dir(ray->direction, hit.normal, 1.0));
    refraction = whitted(c, &next, depth + 1);
  }
  out_color = vec3_add(out_color, surface);
  out_color = vec3_add(out_color, vec3_add(vec3_scalar_mult(reflection, kr), vec3_scalar_mult(refraction, kt)));
  return out_color;
}

/* raytracer.c:375-384 */
static Ray camera_ray(const Camera *cam, double u, double v)
{
  Ray r;
  vec3 on_plane = vec3_add(cam->lower_left_corner,
                           vec3_add(vec3_scalar_mult(cam->horizontal, u), vec3_scalar_mult(cam->vertical, v)));
  r.origin = cam->position;
  r.direction = vec3_normalize(vec3_sub(cam->position, on_plane));
  return r;
}

/* raytracer.c:203-208 with the stream re-seeded per (pixel, sample) */
static vec3 one_sample(Ctx *c, const Camera *cam, int w, int h, uint32_t x, uint32_t y, uint32_t s,
                       uint64_t seed)
{
  c->rng = rt_rng_seed(seed, y * (uint32_t)w + x, s);
  double u = (double)(x + draw(c)) / ((double)w - 1.0);
  double v = (double)(y + draw(c)) / ((double)h - 1.0);
  Ray ray = camera_ray(cam, u, v);
  return c->whitted ? whitted(c, &ray, 0) : trace(c, &ray, 0);
}

/* raytracer.c:218-220: gamma 5, clamp (NaN -> 1 through MIN), truncate */
static uint8_t tone(double x) { return (uint8_t)(255.0 * CLAMP(pow(x, 1 / 5.0))); }

static void ctx_init(Ctx *c, const Object *objs, size_t n, const MeshObject *meshes, size_t nm, int max_depth)
{
  memset(c, 0, sizeof *c);
  c->objs = objs;
  c->n_objs = n;
  c->meshes = meshes;
  c->n_meshes = nm;
  c->max_depth = max_depth;
}

static void ctx_stats(const Ctx *c, long long stats[4])
{
  stats[0] = c->rays;
  stats[1] = c->tests;
  stats[2] = c->casts;
  stats[3] = c->draws;
}

/* ---- exported --------------------------------------------------------------- */

void pto_render_pixels_with(int integrator, const Object *objs, size_t n_objs, const MeshObject *meshes,
                            size_t n_meshes, const Camera *cam, int w, int h, int spp, int max_depth,
                            uint64_t seed, const uint32_t *pixels, size_t npix, double *out_mean,
                            uint8_t *out_rgb8, long long stats[4])
{
  Ctx c;
  ctx_init(&c, objs, n_objs, meshes, n_meshes, max_depth);
  c.whitted = integrator == 1;
  for (size_t k = 0; k < npix; k++)
  {
    uint32_t p = pixels ? pixels[k] : (uint32_t)k;
    uint32_t x = p % (uint32_t)w, y = p / (uint32_t)w;
    vec3 sum = {0, 0, 0};
    for (uint32_t s = 0; s < (uint32_t)spp; s++) /* :201-213 sequential sum */
      sum = vec3_add(sum, one_sample(&c, cam, w, h, x, y, s, seed));
    sum = vec3_scalar_mult(sum, 1.0 / (double)spp); /* :215 */
    if (out_mean)
    {
      out_mean[3 * k + 0] = sum.x;
      out_mean[3 * k + 1] = sum.y;
      out_mean[3 * k + 2] = sum.z;
    }
    if (out_rgb8)
    {
      out_rgb8[3 * k + 0] = tone(sum.x);
      out_rgb8[3 * k + 1] = tone(sum.y);
      out_rgb8[3 * k + 2] = tone(sum.z);
    }
  }
  ctx_stats(&c, stats);
}

void pto_render_pixels(const Object *objs, size_t n_objs, const MeshObject *meshes, size_t n_meshes,
                       const Camera *cam, int w, int h, int spp, int max_depth, uint64_t seed,
                       const uint32_t *pixels, size_t npix, double *out_mean, uint8_t *out_rgb8,
                       long long stats[4])
{
  pto_render_pixels_with(0, objs, n_objs, meshes, n_meshes, cam, w, h, spp, max_depth, seed, pixels, npix,
                         out_mean, out_rgb8, stats);
}

void pto_trace_sample_with(int integrator, const Object *objs, size_t n_objs, const MeshObject *meshes,
                           size_t n_meshes, const Camera *cam, int w, int h, int max_depth, uint32_t x,
                           uint32_t y, uint32_t s, uint64_t seed, double out_rgb[3], long long stats[4])
{
  Ctx c;
  ctx_init(&c, objs, n_objs, meshes, n_meshes, max_depth);
  c.whitted = integrator == 1;
  vec3 r = one_sample(&c, cam, w, h, x, y, s, seed);
  out_rgb[0] = r.x;
  out_rgb[1] = r.y;
  out_rgb[2] = r.z;
  ctx_stats(&c, stats);
}

void pto_trace_sample(const Object *objs, size_t n_objs, const MeshObject *meshes, size_t n_meshes,
                      const Camera *cam, int w, int h, int max_depth, uint32_t x, uint32_t y,
                      uint32_t s, uint64_t seed, double out_rgb[3], long long stats[4])
{
  pto_trace_sample_with(0, objs, n_objs, meshes, n_meshes, cam, w, h, max_depth, x, y, s, seed, out_rgb, stats);
}

void pto_tonemap(const double *mean, size_t npix, uint8_t *out_rgb8)
{
  for (size_t k = 0; k < 3 * npix; k++)
    out_rgb8[k] = tone(mean[k]);
}

static vec3 arr3(const double *p)
{
  vec3 v = {p[0], p[1], p[2]};
  return v;
}
static void out3(double *o, vec3 v)
{
  o[0] = v.x;
  o[1] = v.y;
  o[2] = v.z;
}

/* raytracer.c:47-75.  60 degree vertical FOV; lower_left_corner =
 * (pos - H/2) - (V/2 - (-forward)); llc_old there (:66-68) is dead. */
void pto_init_camera(Camera *cam, const double pos[3], const double target[3], int w, int h)
{
  double theta = 60.0 * (PI / 180);
  double half = tan(theta / 2);
  double viewport_height = 2.0 * half;
  double aspect = (double)w / (double)h;
  double viewport_width = aspect * viewport_height;
  vec3 position = arr3(pos), up_hint = {0, 1, 0};
  vec3 forward = vec3_normalize(vec3_sub(arr3(target), position));
  vec3 right = vec3_normalize(vec3_cross(up_hint, forward));
  vec3 up = vec3_normalize(vec3_cross(forward, right));
  cam->position = position;
  cam->vertical = vec3_scalar_mult(up, viewport_height);
  cam->horizontal = vec3_scalar_mult(right, viewport_width);
  vec3 half_v = vec3_scalar_div(cam->vertical, 2);
  vec3 half_h = vec3_scalar_div(cam->horizontal, 2);
  cam->lower_left_corner =
      vec3_sub(vec3_sub(cam->position, half_h), vec3_sub(half_v, vec3_scalar_mult(forward, -1)));
}

void pto_camera_ray(const Camera *cam, double u, double v, double out[6])
{
  Ray r = camera_ray(cam, u, v);
  out3(out, r.origin);
  out3(out + 3, r.direction);
}

int pto_intersect_sphere(const double ray[6], const double center[3], double radius, double *t)
{
  Ray r = {arr3(ray), arr3(ray + 3)};
  *t = DBL_MAX;
  return sphere_test(&r, arr3(center), radius, t);
}

int pto_intersect_triangle(const double ray[6], const double verts[15], double out_tuv[3])
{
  Ray r = {arr3(ray), arr3(ray + 3)};
  Vertex a = {arr3(verts), {verts[3], verts[4]}};
  Vertex b = {arr3(verts + 5), {verts[8], verts[9]}};
  Vertex c = {arr3(verts + 10), {verts[13], verts[14]}};
  out_tuv[0] = DBL_MAX;
  out_tuv[1] = 0;
  out_tuv[2] = 0;
  return triangle_test(&r, &a, &b, &c, &out_tuv[0], &out_tuv[1], &out_tuv[2]);
}

void pto_surface_normal(const double v[9], double out[3])
{
  out3(out, surface_normal(arr3(v), arr3(v + 3), arr3(v + 6)));
}

void pto_reflect(const double in[3], const double n[3], double out[3])
{
  out3(out, reflect_dir(arr3(in), arr3(n)));
}

void pto_refract(const double in[3], const double n[3], double iot, double out[3])
{
  out3(out, refract_dir(arr3(in), arr3(n), iot));
}

void pto_checkered(const double color[3], double u, double v, double m, double out[3])
{
  out3(out, checker(arr3(color), u, v, m));
}

int pto_intersect_scene(const double ray[6], const Object *objs, size_t n, double out_pn[6],
                        double out_tuv[3], uint32_t *id)
{
  Ctx c;
  Closest hit;
  Ray r = {arr3(ray), arr3(ray + 3)};
  ctx_init(&c, objs, n, NULL, 0, 0);
  memset(&hit, 0, sizeof hit);
  hit.t = DBL_MAX;
  int ok = closest_hit(&c, &r, &hit);
  out3(out_pn, hit.point);
  out3(out_pn + 3, hit.normal);
  out_tuv[0] = hit.t;
  out_tuv[1] = hit.u;
  out_tuv[2] = hit.v;
  *id = (uint32_t)hit.index;
  return ok;
}

/* the scan with meshes; out_tuv: closest t, then Hit.u / .v as the scan leaves them; id: sphere
 * index, or n + mesh index (Hit.object_id of the revived block); tests: primitive tests */
int pto_intersect_mesh_scene(const double ray[6], const Object *objs, size_t n, const MeshObject *meshes,
                             size_t n_meshes, double out_pn[6], double out_tuv[3], uint32_t *id, long long *tests)
{
  Ctx c;
  Closest hit;
  Ray r = {arr3(ray), arr3(ray + 3)};
  ctx_init(&c, objs, n, meshes, n_meshes, 0);
  memset(&hit, 0, sizeof hit);
  hit.t = DBL_MAX;
  int ok = closest_hit(&c, &r, &hit);
  out3(out_pn, hit.point);
  out3(out_pn + 3, hit.normal);
  out_tuv[0] = hit.t;
  out_tuv[1] = hit.u;
  out_tuv[2] = hit.v;
  *id = (uint32_t)(hit.is_mesh ? n + hit.index : hit.index);
  *tests = c.tests;
  return ok;
}

void pto_random_doubles(uint64_t seed, uint32_t pixel, uint32_t sample, int count, double *out)
{
  Ctx c;
  ctx_init(&c, NULL, 0, NULL, 0, 0);
  c.rng = rt_rng_seed(seed, pixel, sample);
  for (int k = 0; k < count; k++)
    out[k] = draw(&c);
}
