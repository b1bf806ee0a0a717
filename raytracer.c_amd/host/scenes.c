/* scenes.c -- the five BASELINE.json configurations as procedural scenes
 * (SURVEY.md section 8d "Synthetic inputs").
 *
 *  1  256x256     4 spp  depth 4   ground + 3 spheres (diffuse, mirror, light)
 *  2  800x600    64 spp  depth 8   ground + 9 seeded packed spheres
 *  3  1920x1080 256 spp  depth 8   ground + 4 spheres + the 12-triangle cube
 *  4  1920x1080 1024 spp depth 16  the reference's own 38-sphere room
 *  5  3840x2160 4096 spp depth 16  that room, packed spheres replaced by a
 *                                  10,240-triangle UV-sphere mesh
 */
#include "scenes.h"
#include "rt_rng.h"

static Object make_sphere(uint flags, double cx, double cy, double cz, double r, vec3 color, vec3 emission)
{
  Object o;
  memset(&o, 0, sizeof o);
  o.flags = flags;
  o.radius = r;
  o.center = VECTOR(cx, cy, cz);
  o.color = color;
  o.emission = emission;
  return o;
}

/* Everything sits on a ground "plane": a radius-1e4 sphere whose top is y = -5. */
#define GROUND_TOP (-5.0)
static Object ground(void)
{
  return make_sphere(M_DEFAULT, 0, -10000.0 + GROUND_TOP, 0, 10000.0, VECTOR(0.75, 0.75, 0.75), BLACK);
}

/* ---- config 1 ---------------------------------------------------------------- */

static void build_c1(Object *o)
{
  o[0] = ground();
  o[1] = make_sphere(M_DEFAULT, -11, 0, 0, 5, VECTOR(0.75, 0.25, 0.25), BLACK);
  o[2] = make_sphere(M_REFLECTION, 0, 0, 0, 5, WHITE, BLACK);
  o[3] = make_sphere(M_DEFAULT, 11, 0, 0, 5, WHITE, VECTOR(4, 4, 4));
}

/* ---- config 2: seeded rejection packing, as main.c:65-138 does with rand() ---- */

static int overlaps(const Object *a, double cx, double cy, double cz, double r, double padding)
{
  vec3 d = vec3_sub(a->center, VECTOR(cx, cy, cz));
  return vec3_length(d) < a->radius + r + padding;
}

static void build_c2(Object *o)
{
  uint64_t rng = rt_rng_seed(0xC2C2C2C2ull, 0, 0);
  size_t placed = 0;
  o[0] = ground();
  while (placed < 9)
  {
    double r = 2.0 + 4.0 * rt_rng_double(&rng);
    double cx = -22.0 + 44.0 * rt_rng_double(&rng);
    double cz = -18.0 + 30.0 * rt_rng_double(&rng);
    double cy = GROUND_TOP + r;
    int bad = 0;
    for (size_t k = 1; k <= placed && !bad; k++)
      bad = overlaps(&o[k], cx, cy, cz, r, 0.5);
    if (bad)
      continue;
    double a = rt_rng_double(&rng), b = rt_rng_double(&rng), c = rt_rng_double(&rng);
    Object s;
    if (placed < 4) /* diffuse, albedo <= 0.9 */
      s = make_sphere(M_DEFAULT, cx, cy, cz, r, VECTOR(0.2 + 0.7 * a, 0.2 + 0.7 * b, 0.2 + 0.7 * c), BLACK);
    else if (placed < 7) /* mirror */
      s = make_sphere(M_REFLECTION, cx, cy, cz, r, WHITE, BLACK);
    else /* light, emission <= 4 */
      s = make_sphere(M_DEFAULT, cx, cy, cz, r, WHITE, VECTOR(1 + 3 * a, 1 + 3 * b, 1 + 3 * c));
    o[++placed] = s;
  }
}

/* ---- config 3: spheres + the cube of the reference's assets/cube.obj ---------- */

/* The 8 corner positions and 6 quads of that file (a Blender cube of side 2
 * with three coordinates perturbed in the 6th decimal), as the float values an
 * OBJ loader yields.  tests/test_host.py checks load_obj() against this table. */
static const float cube_corner[8][3] = {
    {1.000000f, -1.000000f, -1.000000f}, {1.000000f, -1.000000f, 1.000000f},
    {-1.000000f, -1.000000f, 1.000000f}, {-1.000000f, -1.000000f, -1.000000f},
    {1.000000f, 1.000000f, -0.999999f},  {0.999999f, 1.000000f, 1.000001f},
    {-1.000000f, 1.000000f, 1.000000f},  {-1.000000f, 1.000000f, -1.000000f}};
static const int cube_quad[6][4] = {{1, 2, 3, 4}, {5, 8, 7, 6}, {1, 5, 6, 2},
                                    {2, 6, 7, 3}, {3, 7, 8, 4}, {5, 1, 4, 8}};

static Vertex cube_vertex(int corner_1based, double scale, vec3 shift)
{
  const float *p = cube_corner[corner_1based - 1];
  Vertex v;
  v.pos = vec3_add(vec3_scalar_mult(VECTOR((double)p[0], (double)p[1], (double)p[2]), scale), shift);
  v.tex.x = 0;
  v.tex.y = 0;
  return v;
}

static int build_cube(MeshObject *m, double scale, vec3 shift)
{
  Vertex *v = (Vertex *)malloc(sizeof(Vertex) * 36);
  if (!v)
    return -1;
  size_t k = 0;
  for (int q = 0; q < 6; q++) /* fan triangulation: (0,1,2), (0,2,3) */
    for (int t = 1; t <= 2; t++)
    {
      v[k++] = cube_vertex(cube_quad[q][0], scale, shift);
      v[k++] = cube_vertex(cube_quad[q][t], scale, shift);
      v[k++] = cube_vertex(cube_quad[q][t + 1], scale, shift);
    }
  m->mesh.num_triangles = 12;
  m->mesh.vertices = v;
  rt_mesh_flip_winding(&m->mesh);
  return 0;
}

static int build_c3(Object *o, MeshObject *m)
{
  o[0] = ground();
  o[1] = make_sphere(M_DEFAULT, -15, GROUND_TOP + 4, 4, 4, VECTOR(0.75, 0.25, 0.25), BLACK);
  o[2] = make_sphere(M_DEFAULT, 13, GROUND_TOP + 3, 9, 3, VECTOR(0.25, 0.75, 0.25), BLACK);
  o[3] = make_sphere(M_REFLECTION, 15, GROUND_TOP + 5, -9, 5, WHITE, BLACK);
  o[4] = make_sphere(M_DEFAULT, -4, 19, 6, 4, WHITE, VECTOR(6, 6, 6));
  memset(m, 0, sizeof *m);
  m->flags = M_DEFAULT;
  m->color = RGB(109, 124, 187);
  m->emission = BLACK;
  /* side 2 -> side 12, bottom face resting on the ground */
  return build_cube(m, 6.0, VECTOR(0, GROUND_TOP + 6.0, 0));
}

/* ---- config 4: the reference's room (main.c:244-397) -------------------------- */

/* main.c:350-379, the 30 packed spheres, as a table:
 * flags, centre xyz, radius, emission rgb.  Colour is white for all. */
static const struct
{
  uint flags;
  double cx, cy, cz, r, er, eg, eb;
} packed[30] = {
    {M_DEFAULT, 11.8823, 12.8165, -3.43022, 3.47138, 0, 0, 0},
    {M_REFLECTION, -4.78617, -10.565, -11.8307, 7.8185, 0, 0, 0},
    {M_REFLECTION, 16.3283, 15.7456, 8.02745, 3.38449, 0, 0, 0},
    {M_DEFAULT, -7.74563, 7.10781, -1.14851, 5.68239, 0.129721, 1.08691, 0.15077},
    {M_DEFAULT, 0.604958, 13.8198, -10.0857, 3.63955, 0, 0, 0},
    {M_DEFAULT, 2.72773, -3.47742, 7.21287, 5.756, 0.419482, 0.406897, 0.301653},
    {M_REFLECTION, -11.6808, -15.0112, 10.6413, 3.40004, 0, 0, 0},
    {M_REFLECTION, 5.28438, -2.58167, -3.87996, 2.20867, 0, 0, 0},
    {M_DEFAULT, -15.1722, -0.318264, -14.8739, 3.31716, 0, 0, 0},
    {M_DEFAULT, 7.05345, -11.9375, -4.08415, 5.01176, 0, 0, 0},
    {M_DEFAULT, -6.64606, 12.5952, -11.8074, 3.57727, 2.02456, 1.14375, 0.22395},
    {M_REFLECTION, 15.3284, 7.63569, -7.88126, 2.26494, 0, 0, 0},
    {M_REFLECTION, 5.15508, -13.4632, 12.9555, 4.41505, 0, 0, 0},
    {M_DEFAULT, 6.61409, 15.9581, 13.6585, 2.76828, 0, 0, 0},
    {M_REFLECTION, 0.00113487, 8.35296, -14.4917, 2.58514, 0, 0, 0},
    {M_REFLECTION, 9.63578, 9.63074, -16.0336, 2.22603, 0, 0, 0},
    {M_REFLECTION, 13.105, 1.55555, 2.67293, 4.00109, 0, 0, 0},
    {M_REFLECTION, -0.0637789, 6.39925, 11.777, 4.99425, 0, 0, 0},
    {M_DEFAULT, 7.11587, 6.96992, 7.24724, 3.28273, 0.403171, 1.90743, 1.59559},
    {M_DEFAULT, -17.0139, 4.27765, 11.924, 2.14903, 0, 0, 0},
    {M_DEFAULT, 15.3924, -4.96949, 12.4327, 3.48512, 0.647167, 1.99216, 1.4463},
    {M_REFLECTION, -16.0135, 15.9701, 12.4844, 3.00053, 0, 0, 0},
    {M_DEFAULT, -2.87246, -15.5185, 7.78116, 3.4779, 3.16375, 4.44267, 3.49332},
    {M_REFLECTION, -8.89639, -10.9745, -1.80553, 2.39033, 0, 0, 0},
    {M_DEFAULT, -0.653194, 9.99867, 4.17957, 3.28669, 0.662701, 2.82942, 1.50879},
    {M_DEFAULT, -14.6767, -6.47449, 4.48493, 4.77854, 1.6413, 2.60242, 0.421142},
    {M_DEFAULT, -9.76604, 16.8809, -0.605894, 2.89667, 0.479186, 0.149559, 0.3761},
    {M_DEFAULT, 4.07601, 5.6942, -3.07305, 4.91388, 0, 0, 0},
    {M_REFLECTION, 15.1469, -13.988, 9.5646, 4.6719, 0, 0, 0},
    {M_DEFAULT, -7.2047, -5.0758, 7.74727, 2.86742, 0, 0, 0},
};

/* six radius-1e4 wall spheres (main.c:258-299); room 20*aspect x 20 x 30 */
static size_t build_room_walls(Object *o, int width, int height)
{
  const double aspect = (double)width / (double)height;
  const double depth = 30, half_h = 20, half_w = half_h * aspect, R = 10000;
  const vec3 grey = VECTOR(0.75, 0.75, 0.75);
  o[0] = make_sphere(M_DEFAULT, 0, -R - half_h, 0, R, grey, BLACK);                       /* floor */
  o[1] = make_sphere(M_DEFAULT, 0, 0, -R - depth, R, grey, BLACK);                        /* back */
  o[2] = make_sphere(M_DEFAULT, -R - half_w, 0, 0, R, VECTOR(0.25, 0.75, 0.25), BLACK);   /* left */
  o[3] = make_sphere(M_DEFAULT, R + half_w, 0, 0, R, VECTOR(0.75, 0.25, 0.25), BLACK);    /* right */
  o[4] = make_sphere(M_DEFAULT, 0, R + half_h, 0, R, grey, BLACK);                        /* ceiling */
  o[5] = make_sphere(M_DEFAULT, 0, 0, R + depth * 2, R, grey, BLACK);                     /* front */
  return 6;
}

/* the two lights (main.c:383-395) */
static size_t build_room_lights(Object *o)
{
  const double half_h = 20, light_r = 15;
  o[0] = make_sphere(M_DEFAULT, 0, half_h + light_r * 0.9, 0, light_r, WHITE,
                     RGB(0x00 * 15, 0x32 * 15, 0xA0 * 15));
  o[1] = make_sphere(M_DEFAULT, 2, -half_h + 3, 12, 3, WHITE, RGB(0xD0, 0x00, 0x70));
  return 2;
}

static void build_c4(Object *o, int width, int height)
{
  size_t k = build_room_walls(o, width, height);
  for (size_t p = 0; p < 30; p++, k++)
    o[k] = make_sphere(packed[p].flags, packed[p].cx, packed[p].cy, packed[p].cz, packed[p].r, WHITE,
                       VECTOR(packed[p].er, packed[p].eg, packed[p].eb));
  build_room_lights(o + k);
}

/* ---- config 5: the room + a tessellated sphere of exactly 10,240 triangles ---- */

#define C5_LON 80
#define C5_LAT 64

static Vertex uv_vertex(int lon, int lat, double radius, vec3 centre)
{
  const double pi = 3.14159265358979323846;
  double phi = 2.0 * pi * (double)lon / C5_LON;
  double theta = pi * (double)lat / C5_LAT; /* 0 = north pole */
  Vertex v;
  v.pos = vec3_add(centre, VECTOR(radius * sin(theta) * cos(phi), radius * cos(theta),
                                  radius * sin(theta) * sin(phi)));
  v.tex.x = (double)lon / C5_LON;
  v.tex.y = (double)lat / C5_LAT;
  return v;
}

static int build_uv_sphere(MeshObject *m, double radius, vec3 centre)
{
  size_t ntri = (size_t)C5_LON * C5_LAT * 2; /* pole triangles kept (degenerate) */
  Vertex *v = (Vertex *)malloc(sizeof(Vertex) * 3 * ntri);
  if (!v)
    return -1;
  size_t k = 0;
  for (int lat = 0; lat < C5_LAT; lat++)
    for (int lon = 0; lon < C5_LON; lon++)
    {
      Vertex a = uv_vertex(lon, lat, radius, centre), b = uv_vertex(lon + 1, lat, radius, centre);
      Vertex c = uv_vertex(lon + 1, lat + 1, radius, centre), d = uv_vertex(lon, lat + 1, radius, centre);
      /* ordered so that cross(v2-v0, v1-v0) points away from the centre */
      v[k++] = a; v[k++] = c; v[k++] = b;
      v[k++] = a; v[k++] = d; v[k++] = c;
    }
  m->mesh.num_triangles = ntri;
  m->mesh.vertices = v;
  return 0;
}

static int build_c5(Object *o, MeshObject *m, int width, int height)
{
  size_t k = build_room_walls(o, width, height);
  build_room_lights(o + k);
  memset(m, 0, sizeof *m);
  m->flags = M_DEFAULT;
  m->color = RGB(109, 124, 187);
  m->emission = BLACK;
  return build_uv_sphere(m, 8.0, VECTOR(0, -8, 4));
}

/* ---- public -------------------------------------------------------------------- */

static const RtSceneInfo table[5] = {
    {256, 256, 4, 4, {0, 5, 40}, {0, 0, 0}, 4, 0, 0},
    {800, 600, 64, 8, {0, 8, 45}, {0, 0, 0}, 10, 0, 0},
    {1920, 1080, 256, 8, {16, 9, 42}, {0, 0, 0}, 5, 1, 12},
    {1920, 1080, 1024, 16, {0, 0, 50}, {0, 0, 0}, 38, 0, 0},
    {3840, 2160, 4096, 16, {0, 0, 50}, {0, 0, 0}, 8, 1, (size_t)C5_LON *C5_LAT * 2},
};

int rt_scene_info(int config, RtSceneInfo *info)
{
  if (config < 1 || config > 5 || !info)
    return -1;
  *info = table[config - 1];
  return 0;
}

int rt_scene_build(int config, int width, int height, Object *objs, MeshObject *meshes)
{
  switch (config)
  {
  case 1: build_c1(objs); return 0;
  case 2: build_c2(objs); return 0;
  case 3: return build_c3(objs, meshes);
  case 4: build_c4(objs, width, height); return 0;
  case 5: return build_c5(objs, meshes, width, height);
  default: return -1;
  }
}

void rt_scene_free_meshes(MeshObject *meshes, size_t n_meshes)
{
  for (size_t i = 0; meshes && i < n_meshes; i++)
  {
    free(meshes[i].mesh.vertices);
    meshes[i].mesh.vertices = NULL;
    meshes[i].mesh.num_triangles = 0;
  }
}

void rt_mesh_flip_winding(TriangleMesh *mesh)
{
  for (size_t t = 0; t < mesh->num_triangles; t++)
  {
    Vertex tmp = mesh->vertices[3 * t + 1];
    mesh->vertices[3 * t + 1] = mesh->vertices[3 * t + 2];
    mesh->vertices[3 * t + 2] = tmp;
  }
}
