/* raytracer.h -- the drop-in boundary: the reference's public C API, served by
 * an MI355X (gfx950) path tracer.
 *
 * A caller written against gue-ni/raytracer.c's raytracer.h (its main.c, its
 * test.c) compiles and links against this header + libraytracer_amd.so
 * unchanged: every struct below has the reference's field order, size and
 * offsets (checked against the compiled reference in tests/test_oracle_ref.py:
 * Object 88 B, Camera 96 B, Options 56 B, Ray 48 B, Hit 80 B, Vertex 40 B), and
 * every function the reference's raytracer.o exports is exported here with
 * the same signature (reference raytracer.h:135-164).
 *
 * What is different behind the boundary: render() hands the per-pixel
 * trace_path()/intersect() loop (reference raytracer.c:176-223, 482-554,
 * 393-464, 77-174) to hand-written HIP through the C-ABI in rt_hip.h.  There
 * is no CPU fallback: without a GPU render() reports the HIP error on stderr
 * and exits, as the reference's main.c:415-419 does for its own failures.
 *
 * Build-defined extensions (not in the reference) are grouped at the end and
 * prefixed rt_ / named *_ex.
 */
#ifndef RAYTRACER_H
#define RAYTRACER_H

/* libc headers a caller of the reference's header gets transitively (its main.c and test.c
 * rely on that) */
#include <assert.h>
#include <float.h>
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vector.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants and helper macros (reference raytracer.h:21-56) -------------
 * Same names, same values, same expansions as seen by a caller; in particular the
 * argument-evaluation and NaN behaviour of MAX / MIN / CLAMP, which the tonemap and the
 * Russian roulette depend on, and the CLAMP_BETWEEN quirk. */

#if !defined(PI)
#define PI 3.14159265359 /* the reference's truncated pi; it feeds parity (atan2 texture coordinate) */
#endif
#define EPSILON 1e-8     /* self-intersection / parallel-ray threshold */
#if !defined(MAX_DEPTH)
#define MAX_DEPTH 5      /* default bounce limit; at run time: rt_set_max_depth() */
#endif
#define MONTE_CARLO_SAMPLES 1

#define MAX(a, b) ((a) > (b) ? (a) : (b)) /* NaN in `a` selects b, NaN in `b` selects b */
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define CLAMP(x) (MAX(0, MIN(x, 1)))      /* hence CLAMP(NaN) == 1: a NaN pixel tonemaps to 255 */
/* Reproduced quirk (reference raytracer.h:30): the first argument is ignored, so the "cosine"
 * refract() clamps is always 1 and its refracted ray goes straight on. */
#define CLAMP_BETWEEN(x, min_v, max_v) (MAX(min_v, MIN(max_v, 1)))
#define ABS(x) ((x < 0) ? (-x) : (x))
#define EQ(a, b) (ABS((a) - (b)) < EPSILON)

/* vec3 / Ray literals; C++ callers get aggregate initialisation instead of compound literals */
#if defined(__cplusplus)
#define VECTOR(x, y, z) (vec3{(x), (y), (z)})
#define RAY(o, d) (Ray{(o), (d)})
#else
#define VECTOR(x, y, z) ((vec3){(x), (y), (z)})
#define RAY(o, d) ((Ray){.origin = o, .direction = d})
#endif
#define RGB(r, g, b) (VECTOR((r) / 255.0, (g) / 255.0, (b) / 255.0)) /* 8-bit components -> [0, 1] */

#define BLACK RGB(0, 0, 0)
#define WHITE RGB(255, 255, 255)
#define RED RGB(255, 0, 0)
#define GREEN RGB(0, 192, 48)
#define BLUE RGB(0, 0, 255)
#define ZERO_VECTOR RGB(0, 0, 0)
#define ONE_VECTOR (VECTOR(1.0, 1.0, 1.0))
#define BACKGROUND RGB(10, 10, 10) /* radiance of a miss AND of a depth-terminated path (raytracer.c:487-490) */
#define RANDOM_COLOR VECTOR(random_double(), random_double(), random_double())

/* material flags (raytracer.h:53-56): one of the first three decides the bounce -- tested in
 * the order M_REFRACTION, M_REFLECTION, else diffuse (raytracer.c:514-545) -- optionally
 * or-ed with M_CHECKERED */
#define M_DEFAULT ((uint)1 << 1)    /* diffuse */
#define M_REFLECTION ((uint)1 << 2) /* perfect mirror */
#define M_REFRACTION ((uint)1 << 3) /* the reference's two-child "glass" */
#define M_CHECKERED ((uint)1 << 4)  /* albedo x 0.3 / 0.7 by texture coordinate */

/* ---- types (reference raytracer.h:60-131): field order, sizes and offsets are the ABI ----- */

typedef uint32_t uint;

typedef struct Vertex
{
  vec3 pos; /* offset 0 */
  vec2 tex; /* offset 24; (0, 0) when the OBJ has no vt */
} Vertex;   /* 40 bytes */

typedef struct Ray
{
  vec3 origin;
  vec3 direction; /* unit length except after a mirror bounce (raytracer.c:542) */
} Ray;            /* 48 bytes */

/* Unindexed triangle soup: vertices[3*i .. 3*i+2] is triangle i. */
typedef struct TriangleMesh
{
  size_t num_triangles;
  Vertex *vertices;
} TriangleMesh;

/* The live scene element: a sphere with its material inline. */
typedef struct Object
{
  uint flags;    /* offset  0: M_* */
  double radius; /* offset  8 */
  vec3 center;   /* offset 16 */
  vec3 color;    /* offset 40: albedo, components in [0, 1] */
  vec3 emission; /* offset 64: radiance, may exceed 1 */
} Object;        /* 88 bytes */

typedef struct Hit
{
  double t;       /* ray parameter of the closest hit */
  double u, v;    /* texture coordinates: atan2-based on a sphere, interpolated on a triangle */
  vec3 point;
  vec3 normal;    /* unit; geometric, un-flipped */
  uint object_id; /* index into objects[] */
} Hit;            /* 80 bytes */

typedef struct Camera
{
  vec3 position;
  vec3 horizontal;        /* viewport edge vectors */
  vec3 vertical;
  vec3 lower_left_corner; /* see init_camera: with get_camera_ray's subtraction, row 0 is the top */
} Camera;                 /* 96 bytes */

typedef struct Options
{
  vec3 background; /* unused by render() (BACKGROUND is a macro) */
  char *result;    /* output file name */
  char *obj;       /* mesh file name */
  int width, height, samples;
} Options;         /* 56 bytes */

/* Types the reference declares but no live code path uses (its retired mesh-capable Object,
 * raytracer.h:62-102); kept so that code naming them still compiles. */
typedef struct Material
{
  uint flags;
  vec3 color, emission;
  double ka, ks, kd;
} Material;
typedef struct Sphere
{
  vec3 center;
  double radius;
} Sphere;
typedef union Geometry
{
  TriangleMesh *mesh;
  Sphere *sphere;
} Geometry;
typedef enum GeometryType
{
  GEOMETRY_SPHERE,
  GEOMETRY_MESH,
} GeometryType;

/* ---- the reference's exported functions (raytracer.h:135-164), same signatures ------------- */

/* framebuffer: caller-owned width*height*3 bytes, RGB, row 0 = top.  Blocks until the image is
 * complete; adds the device counters to ray_count and intersection_test_count. */
void render(uint8_t *framebuffer, Object *objects, size_t n_objects, Camera *camera, Options *options);
void init_camera(Camera *camera, vec3 position, vec3 target, Options *options);

/* Host-side single-primitive tests, bit-identical to the device code's. */
bool intersect_sphere(const Ray *ray, vec3 center, double radius, Hit *hit);
bool intersect_triangle(const Ray *ray, Vertex vertex0, Vertex vertex1, Vertex vertex2, Hit *hit);
vec3 calculate_surface_normal(vec3 v0, vec3 v1, vec3 v2); /* normalize(cross(v2 - v0, v1 - v0)) */
vec3 point_at(const Ray *ray, double t);

/* Uniform [0,1) and [min,max) from the host-side stream (rt_set_seed). */
double random_double(void);
double random_range(double min, double max);

vec3 clamp(const vec3 v);
void print_v(const char *msg, const vec3 v);
void print_m(const mat4 m);

/* Declared by the reference (raytracer.h:158) but never defined there; defined
 * here: Wavefront OBJ -> unindexed triangle soup, polygons fan-triangulated,
 * positions read as float then widened (what the reference's vendored
 * tinyobj_loader_c yields), tex = (0,0) where the file has no vt.
 * mesh->vertices is malloc'd; the caller frees it. */
bool load_obj(const char *filename, TriangleMesh *mesh);

extern long long ray_count;               /* trace_path()-equivalent calls */
extern long long intersection_test_count; /* primitive tests */

/* ---- build-defined extensions ----------------------------------------------- */

/* A triangle mesh with a material: the scene element the reference's retired
 * Object layout (raytracer.h:95-102) and commented-out mesh scan
 * (raytracer.c:417-435) describe.  Meshes are scanned after the spheres, in
 * array order, triangles in index order. */
typedef struct
{
  uint flags;
  vec3 color, emission;
  TriangleMesh mesh;
} MeshObject;

/* render() plus meshes and an optional linear output: linear_rgb, if not
 * NULL, receives width*height*3 floats, the per-pixel sample MEAN before the
 * gamma-5 tonemap.  framebuffer may be NULL when only linear_rgb is wanted. */
void render_ex(uint8_t *framebuffer, float *linear_rgb, Object *objects, size_t n_objects,
               MeshObject *meshes, size_t n_meshes, Camera *camera, Options *options);

/* Run-time settings the reference fixes at compile time or takes from libc
 * state.  Defaults: depth MAX_DEPTH (5), seed 1666943821 (main.c:182), 1 GPU. */
void rt_set_max_depth(int max_depth);
void rt_set_seed(uint64_t seed);
void rt_set_devices(int n_devices); /* a host that never calls it: RT_DEVICES=N of the environment, else 1 */
int rt_get_max_depth(void);
uint64_t rt_get_seed(void);

/* Which integrator render() runs -- the reference chooses at compile time with the `#if 1`
 * of raytracer.c:207-211: RT_TRACE_PATH (default) = trace_path (:482-554), RT_CAST_RAY =
 * cast_ray (:556-641, Whitted-style: one point light, Phong, shadow rays, mirror and
 * "refraction" children).  Other values are ignored. */
enum { RT_TRACE_PATH = 0, RT_CAST_RAY = 1 };
void rt_set_integrator(int integrator);
int rt_get_integrator(void);

/* Cooperative cancellation (what the reference's SIGINT handler, main.c:37-48,422, is for):
 * while a flag is registered, long renders poll it between slabs of tiles; when it becomes
 * non-zero render()/render_ex() return early with the finished part of the image in the
 * framebuffer (the rest stays as the caller initialised it / zero) and
 * rt_last_render_cancelled() reports 1.  Set the flag from a signal handler. */
void rt_set_cancel_flag(const volatile int *flag);
int rt_last_render_cancelled(void);

/* Kernel-only wall time of the last render()/render_ex(), seconds, and the
 * count of scene casts (rays that ran the intersection scan). */
double rt_last_render_seconds(void);
long long rt_last_ray_bounces(void);

#ifdef __cplusplus
}
#endif

#endif /* RAYTRACER_H */
