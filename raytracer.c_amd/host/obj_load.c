/* obj_load.c -- load_obj(): declared by the reference (raytracer.h:158) but
 * never defined there; its vendored tinyobj_loader_c is compiled in and never
 * called.  This is an independent, minimal Wavefront OBJ reader
 * producing what that loader would hand a caller:
 *   - `v x y z` / `vt u v` values are stored as float and widened to double;
 *   - faces of any arity are fan-triangulated: (0,1,2), (0,2,3), ...;
 *   - indices are 1-based, negative = relative to the end;
 *   - a vertex without a vt reference gets tex = (0,0);
 *   - normals, groups, materials, smoothing are ignored (the renderer takes
 *     geometric normals from calculate_surface_normal, reference raytracer.c:431).
 * Output: unindexed triangle soup, vertices[3*i .. 3*i+2] = triangle i, in
 * file order -- the layout the reference's mesh scan walks (raytracer.c:420-424).
 */
#include <ctype.h>
#include <errno.h>

#include "raytracer.h"

typedef struct
{
  float *data;
  size_t count, cap; /* in floats */
} FloatBuf;

typedef struct
{
  Vertex *data;
  size_t count, cap;
} VertexBuf;

static int fb_push(FloatBuf *b, float v)
{
  if (b->count == b->cap)
  {
    size_t ncap = b->cap ? b->cap * 2 : 256;
    float *n = (float *)realloc(b->data, ncap * sizeof(float));
    if (!n)
      return -1;
    b->data = n;
    b->cap = ncap;
  }
  b->data[b->count++] = v;
  return 0;
}

static int vb_push(VertexBuf *b, Vertex v)
{
  if (b->count == b->cap)
  {
    size_t ncap = b->cap ? b->cap * 2 : 96;
    Vertex *n = (Vertex *)realloc(b->data, ncap * sizeof(Vertex));
    if (!n)
      return -1;
    b->data = n;
    b->cap = ncap;
  }
  b->data[b->count++] = v;
  return 0;
}

/* parse up to `want` floats from *s; returns how many were read */
static int parse_floats(const char *s, float *out, int want)
{
  int got = 0;
  while (got < want)
  {
    char *end;
    errno = 0;
    float f = strtof(s, &end);
    if (end == s)
      break;
    out[got++] = f;
    s = end;
  }
  return got;
}

/* resolve a 1-based / negative OBJ index against `count` elements; -1 = bad */
static long resolve(long idx, size_t count)
{
  if (idx > 0 && (size_t)idx <= count)
    return idx - 1;
  if (idx < 0 && (size_t)(-idx) <= count)
    return (long)count + idx;
  return -1;
}

/* one face corner "v", "v/vt", "v//vn", "v/vt/vn" -> Vertex */
static int parse_corner(const char **ps, const FloatBuf *pos, const FloatBuf *tex, Vertex *out)
{
  const char *s = *ps;
  char *end;
  long vi = strtol(s, &end, 10);
  if (end == s)
    return -1;
  long ti = 0;
  int have_t = 0;
  s = end;
  if (*s == '/')
  {
    s++;
    if (*s != '/' && !isspace((unsigned char)*s) && *s)
    {
      ti = strtol(s, &end, 10);
      have_t = end != s;
      s = end;
    }
    if (*s == '/')
    {
      s++;
      (void)strtol(s, &end, 10); /* normal index: ignored */
      s = end;
    }
  }
  *ps = s;
  long p = resolve(vi, pos->count / 3);
  if (p < 0)
    return -1;
  out->pos.x = (double)pos->data[3 * p + 0];
  out->pos.y = (double)pos->data[3 * p + 1];
  out->pos.z = (double)pos->data[3 * p + 2];
  out->tex.x = 0;
  out->tex.y = 0;
  if (have_t)
  {
    long t = resolve(ti, tex->count / 2);
    if (t >= 0)
    {
      out->tex.x = (double)tex->data[2 * t + 0];
      out->tex.y = (double)tex->data[2 * t + 1];
    }
  }
  return 0;
}

bool load_obj(const char *filename, TriangleMesh *mesh)
{
  if (!filename || !mesh)
    return false;
  FILE *f = fopen(filename, "r");
  if (!f)
    return false;

  FloatBuf pos = {0}, tex = {0};
  VertexBuf tris = {0};
  bool ok = true;
  char *line = NULL;
  size_t cap = 0;
  ssize_t len;
  while (ok && (len = getline(&line, &cap, f)) >= 0)
  {
    const char *s = line;
    while (*s == ' ' || *s == '\t')
      s++;
    if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t'))
    {
      float v[3] = {0, 0, 0};
      if (parse_floats(s + 2, v, 3) < 3)
        ok = false;
      for (int k = 0; ok && k < 3; k++)
        ok = fb_push(&pos, v[k]) == 0;
    }
    else if (s[0] == 'v' && s[1] == 't' && (s[2] == ' ' || s[2] == '\t'))
    {
      float v[2] = {0, 0};
      if (parse_floats(s + 3, v, 2) < 1)
        ok = false;
      for (int k = 0; ok && k < 2; k++)
        ok = fb_push(&tex, v[k]) == 0;
    }
    else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t'))
    {
      Vertex first, prev, cur;
      int n = 0;
      s += 2;
      for (;;)
      {
        while (*s == ' ' || *s == '\t')
          s++;
        if (*s == '\0' || *s == '\n' || *s == '\r' || *s == '#')
          break;
        if (parse_corner(&s, &pos, &tex, &cur) != 0)
        {
          ok = false;
          break;
        }
        if (n == 0)
          first = cur;
        else if (n >= 2)
          ok = vb_push(&tris, first) == 0 && vb_push(&tris, prev) == 0 && vb_push(&tris, cur) == 0;
        prev = cur;
        n++;
        if (!ok)
          break;
      }
      if (ok && n < 3)
        ok = false; /* a face needs three corners */
    }
    /* everything else (vn, o, g, s, usemtl, mtllib, comments) is ignored */
  }
  free(line);
  fclose(f);
  free(pos.data);
  free(tex.data);
  if (!ok || tris.count == 0)
  {
    free(tris.data);
    return false;
  }
  mesh->num_triangles = tris.count / 3;
  mesh->vertices = tris.data;
  return true;
}
