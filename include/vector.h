/* vector.h -- fp64 small-vector math of the raytracer.h boundary.
 *
 * Source-compatible with the reference's vector.h (same type and function
 * names, same argument order) so a caller written against it compiles against
 * this one.  The arithmetic ORDER of every operation below is part of the
 * parity contract and follows the reference exactly:
 *   dot       = (ax*bx + ay*by) + az*bz            (reference vector.h:25-26)
 *   normalize = v * (1.0 / sqrt(dot(v,v)))         (vector.h:53-58, reciprocal
 *                                                   then multiply, never divide)
 *   scalar_div(v,s) = v * (1.0 / s)                (vector.h:34-35)
 *   cross as component formulas of vector.h:43-48
 * Deliberate fix: functions are `static inline`, so the
 * header links at every optimisation level (the reference's plain C99
 * `inline` does not link at -O0).
 */
#ifndef VECTOR_M
#define VECTOR_M

#include <assert.h>
#include <math.h>

typedef double REAL;

typedef struct { REAL x, y; } vec2;
typedef struct { REAL x, y, z; } vec3;
typedef struct { REAL x, y, z, w; } vec4;
typedef REAL mat2[4];
typedef REAL mat3[9];
typedef REAL mat4[16];

static inline vec3 vec3_add(vec3 a, vec3 b)
{
  vec3 r = {a.x + b.x, a.y + b.y, a.z + b.z};
  return r;
}

static inline vec3 vec3_sub(vec3 a, vec3 b)
{
  vec3 r = {a.x - b.x, a.y - b.y, a.z - b.z};
  return r;
}

static inline vec3 vec3_mult(vec3 a, vec3 b)
{
  vec3 r = {a.x * b.x, a.y * b.y, a.z * b.z};
  return r;
}

static inline vec3 vec3_scalar_mult(vec3 v, REAL s)
{
  vec3 r = {v.x * s, v.y * s, v.z * s};
  return r;
}

static inline vec3 vec3_scalar_div(vec3 v, REAL s)
{
  return vec3_scalar_mult(v, 1.0 / s);
}

static inline REAL vec3_dot(vec3 a, vec3 b)
{
  return a.x * b.x + a.y * b.y + a.z * b.z;
}

static inline REAL vec3_length(vec3 v)
{
  return sqrt(vec3_dot(v, v));
}

static inline vec3 vec3_cross(vec3 a, vec3 b)
{
  vec3 r;
  r.x = a.y * b.z - a.z * b.y;
  r.y = a.z * b.x - a.x * b.z;
  r.z = a.x * b.y - a.y * b.x;
  return r;
}

static inline int vec3_equal(vec3 a, vec3 b)
{
  return a.x == b.x && a.y == b.y && a.z == b.z;
}

static inline vec3 vec3_normalize(vec3 v)
{
  REAL len = vec3_length(v);
  assert(len > 0);
  return vec3_scalar_mult(v, 1.0 / len);
}

static inline vec2 vec2_add(vec2 a, vec2 b)
{
  vec2 r = {a.x + b.x, a.y + b.y};
  return r;
}

static inline vec2 vec2_scalar_mult(vec2 v, REAL s)
{
  vec2 r = {v.x * s, v.y * s};
  return r;
}

/* Row-major 4x4 times (v,1); returns xyz.  Off the hot path (reference
 * vector.h:63-74, used only by a never-called mesh helper, main.c:140-147). */
static inline vec3 mat4_vector_mult(const mat4 A, vec3 v)
{
  const REAL in[4] = {v.x, v.y, v.z, 1.0};
  REAL out[4];
  for (int row = 0; row < 4; row++)
  {
    REAL acc = 0;
    for (int k = 0; k < 4; k++)
      acc += A[row * 4 + k] * in[k];
    out[row] = acc;
  }
  vec3 r = {out[0], out[1], out[2]};
  return r;
}

/* C = A * B, row-major.  (The reference's version indexes C and B with the
 * wrong stride, vector.h:83; it is never called.  This one is correct.) */
static inline void mat4_mult(const mat4 A, const mat4 B, mat4 C)
{
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
    {
      REAL acc = 0;
      for (int k = 0; k < 4; k++)
        acc += A[i * 4 + k] * B[k * 4 + j];
      C[i * 4 + j] = acc;
    }
}

#endif /* VECTOR_M */
