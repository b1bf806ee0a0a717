// fp64 VALU issue-rate microbenchmark for gfx950: pins the roofline "peak" used by bench.py.
// Each lane runs 8 independent dependency chains of the instruction under test; the
// loop is long enough that launch overhead is negligible.  Prints lane-ops/s for
//   add : v_add_f64        mul : v_mul_f64        fma : v_fma_f64 (counted as ONE op)
//   mix : alternating mul/add (the shape of the un-contracted dot products in pt_kernel.hip)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fp64_peak.hip -o tools/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void chains(double *out, int iters, double a, double b)
{
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int u = 0; u < 8; u++)
    {
      if (MODE == 0) { x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a; }
      if (MODE == 1) { x0 *= b; x1 *= b; x2 *= b; x3 *= b; x4 *= b; x5 *= b; x6 *= b; x7 *= b; }
      if (MODE == 2) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
                       x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a); }
      if (MODE == 3) { x0 *= b; x1 += a; x2 *= b; x3 += a; x4 *= b; x5 += a; x6 *= b; x7 += a; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
double run(const char *name, int blocks, int iters)
{
  double *d;
  hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 1e-9, 1.0000001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(chains<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9, 1.0000001);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * (double)iters * 64.0;
  double rate = ops / (ms * 1e-3);
  printf("%-4s %8.3f ms  %.3e lane-ops/s  (%.2f T)\n", name, ms, rate, rate * 1e-12);
  hipFree(d);
  return rate;
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s  CUs=%d  clock=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  const int blocks = p.multiProcessorCount * 8, iters = 20000;
  for (int rep = 0; rep < 2; rep++)
  {
    run<0>("add", blocks, iters);
    run<1>("mul", blocks, iters);
    run<2>("fma", blocks, iters);
    run<3>("mix", blocks, iters);
  }
  return 0;
}
