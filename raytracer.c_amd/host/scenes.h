/* scenes.h -- procedural, seeded builders for the five BASELINE.json
 * configurations, emitted as the reference's own scene types (Object[] of
 * raytracer.h, plus the MeshObject extension).  The same arrays feed the HIP
 * path, the CPU oracle and the CPU baseline, so their exact values are inputs,
 * not part of the parity contract.
 */
#ifndef RT_SCENES_H
#define RT_SCENES_H

#include "raytracer.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
  int width, height, samples, max_depth; /* the configuration's nominal settings */
  double cam_pos[3], cam_target[3];
  size_t n_objects; /* spheres */
  size_t n_meshes;
  size_t n_triangles; /* over all meshes */
} RtSceneInfo;

/* config: 1..5 = BASELINE.json configs[0..4].  Returns 0, or -1 if unknown. */
int rt_scene_info(int config, RtSceneInfo *info);

/* Fills objs[info.n_objects] and meshes[info.n_meshes] (mesh vertex arrays are
 * malloc'd: release with rt_scene_free_meshes).  width/height matter only to
 * config 4/5, whose room width follows the aspect ratio (reference
 * main.c:244-247).  Returns 0 on success. */
int rt_scene_build(int config, int width, int height, Object *objs, MeshObject *meshes);
void rt_scene_free_meshes(MeshObject *meshes, size_t n_meshes);

/* Swap vertices 1 and 2 of every triangle, so that the reference's
 * calculate_surface_normal() (cross(v2-v0, v1-v0), raytracer.c:42-45) points
 * outward for a mesh authored counter-clockwise (OBJ convention). */
void rt_mesh_flip_winding(TriangleMesh *mesh);

#ifdef __cplusplus
}
#endif

#endif /* RT_SCENES_H */
