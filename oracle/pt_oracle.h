/* oracle/pt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * C interface of the CPU restatement of the reference hot path (pt_oracle.c).
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may use
 * it.  All entry points are re-entrant (state lives in the caller's arguments),
 * so a caller may run them from several threads or processes on disjoint
 * pixel sets.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "raytracer.h" /* include/: boundary struct layouts only */

/* stats[0] = rays   : trace_path()-equivalent calls, incl. depth-terminated
 *                     ones (reference ray_count, raytracer.c:484)
 * stats[1] = tests  : primitive tests (intersection_test_count, :79,:122)
 * stats[2] = casts  : rays that ran the scene scan ("ray-bounces")
 * stats[3] = draws  : RNG draws */
void pto_render_pixels(const Object *objs, size_t n_objs, const MeshObject *meshes, size_t n_meshes,
                       const Camera *cam, int w, int h, int spp, int max_depth, uint64_t seed,
                       const uint32_t *pixels, size_t npix, double *out_mean, uint8_t *out_rgb8,
                       long long stats[4]);

void pto_trace_sample(const Object *objs, size_t n_objs, const MeshObject *meshes, size_t n_meshes,
                      const Camera *cam, int w, int h, int max_depth, uint32_t x, uint32_t y,
                      uint32_t s, uint64_t seed, double out_rgb[3], long long stats[4]);

/* The same two with the integrator chosen: 0 = trace_path (raytracer.c:482-554, what render()
 * calls as shipped), 1 = cast_ray (raytracer.c:556-641, the other side of render()'s `#if 1`).
 * For cast_ray, "casts" counts both the primary and the shadow scan of every hit. */
void pto_render_pixels_with(int integrator, const Object *objs, size_t n_objs, const MeshObject *meshes,
                            size_t n_meshes, const Camera *cam, int w, int h, int spp, int max_depth,
                            uint64_t seed, const uint32_t *pixels, size_t npix, double *out_mean,
                            uint8_t *out_rgb8, long long stats[4]);
void pto_trace_sample_with(int integrator, const Object *objs, size_t n_objs, const MeshObject *meshes,
                           size_t n_meshes, const Camera *cam, int w, int h, int max_depth, uint32_t x,
                           uint32_t y, uint32_t s, uint64_t seed, double out_rgb[3], long long stats[4]);

void pto_init_camera(Camera *cam, const double pos[3], const double target[3], int w, int h);
void pto_camera_ray(const Camera *cam, double u, double v, double out[6]);
int pto_intersect_sphere(const double ray[6], const double center[3], double radius, double *t);
int pto_intersect_triangle(const double ray[6], const double verts[15], double out_tuv[3]);
void pto_surface_normal(const double v[9], double out[3]);
void pto_reflect(const double in[3], const double n[3], double out[3]);
void pto_refract(const double in[3], const double n[3], double iot, double out[3]);
void pto_checkered(const double color[3], double u, double v, double m, double out[3]);
int pto_intersect_scene(const double ray[6], const Object *objs, size_t n, double out_pn[6],
                        double out_tuv[3], uint32_t *id);
int pto_intersect_mesh_scene(const double ray[6], const Object *objs, size_t n, const MeshObject *meshes,
                             size_t n_meshes, double out_pn[6], double out_tuv[3], uint32_t *id, long long *tests);
void pto_random_doubles(uint64_t seed, uint32_t pixel, uint32_t sample, int count, double *out);
void pto_tonemap(const double *mean, size_t npix, uint8_t *out_rgb8);

#endif /* PT_ORACLE_H */
