/* raytracer_amd.c -- the reference's raytracer.h API on top of the HIP shim.
 *
 * This file is the host half of the drop-in: it exports every symbol the
 * reference's raytracer.o exports (reference
 * raytracer.h:135-164) with the same signatures and struct layouts, and
 * implements render() by handing the whole per-pixel loop nest
 * (reference raytracer.c:176-223) to librt_hip.so (include/rt_hip.h).
 *
 * The small host-side helpers (init_camera, intersect_sphere,
 * intersect_triangle, calculate_surface_normal, point_at) are the same
 * arithmetic, in the same order, as the device code, so a caller that probes
 * single primitives on the host sees what the GPU computes.
 *
 * No CPU rendering path: if the shim cannot run, render() says why on stderr
 * and exits with EXIT_FAILURE (the reference's own failure convention,
 * main.c:415-419; render() returns void, so there is no error channel).
 */
#include "raytracer.h"
#include "rt_hip.h"
#include "rt_rng.h"

long long ray_count = 0;
long long intersection_test_count = 0;

static int g_max_depth = MAX_DEPTH;
static uint64_t g_seed = 1666943821ull; /* reference main.c:182 */
static int g_devices = 0; /* 0: not set -- RT_DEVICES of the environment, else 1 (devices_to_use) */
static int g_integrator = RT_TRACE_PATH;
static uint64_t g_host_rng = 0;
static double g_last_seconds = 0;
static long long g_last_bounces = 0;
static int g_last_cancelled = 0;

_Static_assert(sizeof(Object) == sizeof(RtHipSphere), "Object must be passable as RtHipSphere");
_Static_assert(offsetof(Object, radius) == offsetof(RtHipSphere, radius), "Object.radius");
_Static_assert(offsetof(Object, center) == offsetof(RtHipSphere, center), "Object.center");
_Static_assert(offsetof(Object, color) == offsetof(RtHipSphere, color), "Object.color");
_Static_assert(offsetof(Object, emission) == offsetof(RtHipSphere, emission), "Object.emission");
_Static_assert(sizeof(Vertex) == sizeof(RtHipVertex), "Vertex must be passable as RtHipVertex");
_Static_assert(sizeof(Camera) == sizeof(RtHipCamera), "Camera must be passable as RtHipCamera");

/* ---- settings ------------------------------------------------------------------- */

void rt_set_max_depth(int max_depth) { g_max_depth = max_depth < 0 ? 0 : max_depth; }
int rt_get_max_depth(void) { return g_max_depth; }
void rt_set_seed(uint64_t seed)
{
  g_seed = seed;
  g_host_rng = 0;
}
uint64_t rt_get_seed(void) { return g_seed; }
void rt_set_devices(int n_devices) { g_devices = n_devices < 1 ? 1 : n_devices; }
void rt_set_integrator(int integrator)
{
  if (integrator == RT_TRACE_PATH || integrator == RT_CAST_RAY)
    g_integrator = integrator;
}
int rt_get_integrator(void) { return g_integrator; }
double rt_last_render_seconds(void) { return g_last_seconds; }
void rt_set_cancel_flag(const volatile int *flag) { rt_hip_set_cancel_flag(flag); }
int rt_last_render_cancelled(void) { return g_last_cancelled; }
long long rt_last_ray_bounces(void) { return g_last_bounces; }

/* ---- host-side RNG (reference raytracer.c:227-229) ------------------------------ */

/* The host stream is the (seed, pixel = 2^32-1, sample = 2^32-1) stream of
 * rt_rng.h: no image pixel can have that index (images are < 2^32 pixels). */
double random_double(void)
{
  if (!g_host_rng)
    g_host_rng = rt_rng_seed(g_seed, 0xFFFFFFFFu, 0xFFFFFFFFu);
  return rt_rng_double(&g_host_rng);
}

double random_range(double min, double max) { return random_double() * (max - min) + min; }

/* ---- primitives ----------------------------------------------------------------- */

vec3 point_at(const Ray *ray, double t)
{
  return vec3_add(ray->origin, vec3_scalar_mult(ray->direction, t));
}

/* Winding as the reference has it (raytracer.c:42-45): cross(v2-v0, v1-v0).
 * For (-1,1,1),(1,1,1),(1,1,-1) this is (0,-1,0) -- the reference's own
 * test.c:78 expects (0,1,0) and fails; the function, not the test, is the
 * behaviour every caller sees, so it is what is reproduced. */
vec3 calculate_surface_normal(vec3 v0, vec3 v1, vec3 v2)
{
  return vec3_normalize(vec3_cross(vec3_sub(v2, v0), vec3_sub(v1, v0)));
}

vec3 clamp(const vec3 v)
{
  vec3 r = {CLAMP(v.x), CLAMP(v.y), CLAMP(v.z)};
  return r;
}

/* reference raytracer.c:77-118 */
bool intersect_sphere(const Ray *ray, vec3 center, double radius, Hit *hit)
{
  intersection_test_count++;
  vec3 to_center = vec3_sub(center, ray->origin);
  double tca = vec3_dot(to_center, ray->direction);
  if (tca < 0)
    return false;
  double d2 = vec3_dot(to_center, to_center) - tca * tca;
  double r2 = radius * radius;
  if (d2 > r2)
    return false;
  double thc = sqrt(r2 - d2);
  double near_t = tca - thc, far_t = tca + thc;
  if (near_t > far_t)
  {
    double swap = near_t;
    near_t = far_t;
    far_t = swap;
  }
  if (near_t < 0)
    near_t = far_t; /* origin inside the sphere: leave through the far side */
  if (!(near_t > EPSILON))
    return false;
  hit->t = near_t;
  return true;
}

/* reference raytracer.c:120-174 */
bool intersect_triangle(const Ray *ray, Vertex vertex0, Vertex vertex1, Vertex vertex2, Hit *hit)
{
  intersection_test_count++;
  vec3 e1 = vec3_sub(vertex1.pos, vertex0.pos);
  vec3 e2 = vec3_sub(vertex2.pos, vertex0.pos);
  vec3 h = vec3_cross(ray->direction, e2);
  double det = vec3_dot(e1, h);
  if (det > -EPSILON && det < EPSILON)
    return false;
  double f = 1.0 / det;
  vec3 s = vec3_sub(ray->origin, vertex0.pos);
  double u = f * vec3_dot(s, h);
  if (u < 0.0 || u > 1.0)
    return false;
  vec3 q = vec3_cross(s, e1);
  double v = f * vec3_dot(ray->direction, q);
  if (v < 0.0 || u + v > 1.0)
    return false;
  double t = f * vec3_dot(e2, q);
  if (!(t > EPSILON))
    return false;
  vec2 tex = vec2_add(vec2_add(vec2_scalar_mult(vertex0.tex, 1 - u - v), vec2_scalar_mult(vertex1.tex, u)),
                      vec2_scalar_mult(vertex2.tex, v));
  hit->t = t;
  hit->u = tex.x;
  hit->v = tex.y;
  return true;
}

void print_v(const char *msg, const vec3 v) { printf("%s: (vec3) { %f, %f, %f }\n", msg, v.x, v.y, v.z); }

void print_m(const mat4 m)
{
  for (int r = 0; r < 4; r++)
  {
    for (int c = 0; c < 4; c++)
      printf(" %6.1f, ", m[r * 4 + c]);
    printf("\n");
  }
}

/* ---- camera (reference raytracer.c:47-75) ---------------------------------------- */

void init_camera(Camera *camera, vec3 position, vec3 target, Options *options)
{
  const double fov = 60.0 * (PI / 180); /* fixed 60 degree vertical field of view */
  const double half = tan(fov / 2);
  const double view_h = 2.0 * half;
  const double aspect = (double)options->width / (double)options->height;
  const double view_w = aspect * view_h;

  vec3 y_axis = {0, 1, 0};
  vec3 forward = vec3_normalize(vec3_sub(target, position));
  vec3 right = vec3_normalize(vec3_cross(y_axis, forward));
  vec3 up = vec3_normalize(vec3_cross(forward, right));

  camera->position = position;
  camera->vertical = vec3_scalar_mult(up, view_h);
  camera->horizontal = vec3_scalar_mult(right, view_w);
  vec3 half_v = vec3_scalar_div(camera->vertical, 2);
  vec3 half_h = vec3_scalar_div(camera->horizontal, 2);
  /* (pos - H/2) - (V/2 - (-forward)): with get_camera_ray's pos - (llc + Hu + Vv)
   * this yields an upright image with row 0 at the top. */
  camera->lower_left_corner =
      vec3_sub(vec3_sub(camera->position, half_h), vec3_sub(half_v, vec3_scalar_mult(forward, -1)));
}

/* ---- render ---------------------------------------------------------------------- */

/* rt_set_devices() wins; a host that never calls it (the reference's main.c, unmodified, behind this library) can be
 * given a device count through RT_DEVICES=N in the environment. */
static int devices_to_use(void)
{
  if (g_devices > 0)
    return g_devices;
  const char *e = getenv("RT_DEVICES");
  const int n = e ? atoi(e) : 0;
  return n >= 1 ? n : 1;
}

void render_ex(uint8_t *framebuffer, float *linear_rgb, Object *objects, size_t n_objects,
               MeshObject *meshes, size_t n_meshes, Camera *camera, Options *options)
{
  RtHipMesh *hm = NULL;
  if (n_meshes)
  {
    hm = (RtHipMesh *)calloc(n_meshes, sizeof *hm);
    if (!hm)
    {
      fprintf(stderr, "render: out of memory\n");
      exit(EXIT_FAILURE);
    }
    for (size_t m = 0; m < n_meshes; m++)
    {
      hm[m].flags = meshes[m].flags;
      memcpy(hm[m].color, &meshes[m].color, sizeof hm[m].color);
      memcpy(hm[m].emission, &meshes[m].emission, sizeof hm[m].emission);
      hm[m].num_triangles = meshes[m].mesh.num_triangles;
      hm[m].vertices = (const RtHipVertex *)meshes[m].mesh.vertices;
    }
  }

  RtHipParams p;
  memset(&p, 0, sizeof p);
  p.width = options->width;
  p.height = options->height;
  p.samples = options->samples;
  p.max_depth = g_max_depth;
  p.seed = g_seed;
  p.integrator = g_integrator == RT_CAST_RAY ? RT_HIP_CAST_RAY : RT_HIP_TRACE_PATH;

  uint64_t stats[RT_HIP_NSTATS] = {0, 0, 0, 0};
  double seconds = 0;
  int rc = rt_hip_render_image((const RtHipSphere *)objects, n_objects, hm, n_meshes, (const RtHipCamera *)camera,
                               &p, devices_to_use(), linear_rgb, framebuffer, stats, &seconds);
  free(hm);
  g_last_cancelled = rc == RT_HIP_ECANCELLED;
  if (rc != RT_HIP_OK && rc != RT_HIP_ECANCELLED)
  {
    fprintf(stderr, "render: GPU path failed (%d): %s\n", rc, rt_hip_last_error());
    exit(EXIT_FAILURE);
  }
  /* the reference's globals accumulate over calls (raytracer.c:36-37, 79, 484) */
  ray_count += (long long)stats[RT_HIP_STAT_RAYS];
  intersection_test_count += (long long)stats[RT_HIP_STAT_TESTS];
  g_last_seconds = seconds;
  g_last_bounces = (long long)stats[RT_HIP_STAT_CASTS];
}

void render(uint8_t *framebuffer, Object *objects, size_t n_objects, Camera *camera, Options *options)
{
  render_ex(framebuffer, NULL, objects, n_objects, NULL, 0, camera, options);
}
