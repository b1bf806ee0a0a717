// Checks, on the GPU, the register layouts the MFMA form of the phase-1 filter relies on (pt_kernel.hip, filter_mfma):
//   v_mfma_f32_16x16x4_f32: lane l holds A[l % 16][l / 16], B[l / 16][l % 16], D[4 (l / 16) + v][l % 16] (v = 0..3),
//     and D = fma chain over k, bit-identical to fmaf in SOME order of k (all 24 orders are tried);
//   v_permlane32_swap_b32 a, b: lanes 32-63 of a <-> lanes 0-31 of b;  v_permlane16_swap_b32: odd rows of a <-> even rows of b.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_probe tools/mfma_probe.hip ; run: ./tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float *a, const float *b, float *d, unsigned *t)
{
  const unsigned l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
  d[4 * l + 0] = acc.x; d[4 * l + 1] = acc.y; d[4 * l + 2] = acc.z; d[4 * l + 3] = acc.w;
  unsigned v0 = 100 + l, v1 = 200 + l;
  u32x2 r = __builtin_amdgcn_permlane32_swap(v0, v1, false, false);
  u32x2 s = __builtin_amdgcn_permlane16_swap(v0, v1, false, false);
  t[4 * l + 0] = r.x; t[4 * l + 1] = r.y; t[4 * l + 2] = s.x; t[4 * l + 3] = s.y;
}
int main()
{
  float A[16][4], B[4][16], ha[64], hb[64], hd[256];
  unsigned ht[256];
  srand(7);
  auto rnd = [] { return (float)((rand() % 2000001 - 1000000) * 1.37e-3) * (float)(1 + rand() % 1000); };
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i][k] = rnd();
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k][j] = rnd();
  for (int l = 0; l < 64; l++) { ha[l] = A[l % 16][l / 16]; hb[l] = B[l / 16][l % 16]; }
  float *da, *db, *dd; unsigned *dt;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd); hipMalloc(&dt, sizeof ht);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, dt);
  hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost); hipMemcpy(ht, dt, sizeof ht, hipMemcpyDeviceToHost);
  int perm[4] = {0, 1, 2, 3}, best_bad = 1 << 30, best[4] = {0, 0, 0, 0};
  do {
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
      const int i = 4 * (l / 16) + v, j = l % 16;
      float acc = 0.f;
      for (int q = 0; q < 4; q++) acc = fmaf(A[i][perm[q]], B[perm[q]][j], acc);
      bad += memcmp(&acc, &hd[4 * l + v], 4) != 0;
    }
    if (bad < best_bad) { best_bad = bad; memcpy(best, perm, sizeof best); }
  } while (std::next_permutation(perm, perm + 4));
  printf("mfma 16x16x4 f32: layout D[4(l/16)+v][l%%16]; best k order %d%d%d%d: %d of 256 values differ from the fmaf chain\n", best[0], best[1], best[2], best[3], best_bad);
  int bad32 = 0, bad16 = 0;
  for (unsigned l = 0; l < 64; l++) {
    const unsigned e32a = l < 32 ? 100 + l : 200 + (l - 32), e32b = l < 32 ? 100 + (l + 32) : 200 + l;
    const unsigned row = l / 16, e16a = (row & 1) ? 200 + (l - 16) : 100 + l, e16b = (row & 1) ? 200 + l : 100 + (l + 16);
    bad32 += ht[4 * l + 0] != e32a || ht[4 * l + 1] != e32b;
    bad16 += ht[4 * l + 2] != e16a || ht[4 * l + 3] != e16b;
  }
  printf("permlane32_swap: %d lanes differ from the assumed semantics; permlane16_swap: %d\n", bad32, bad16);
  if (bad32 || bad16) for (int l = 0; l < 64; l += 8) printf("lane %2d: p32 (%u, %u)  p16 (%u, %u)\n", l, ht[4 * l], ht[4 * l + 1], ht[4 * l + 2], ht[4 * l + 3]);
  return (best_bad || bad32 || bad16) ? 1 : 0;
}
