// Issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the path
// tracer leans on, gfx950.  Every wave runs 8 independent chains of one instruction (inline
// asm, so the compiler cannot fold them); with >= 2 waves per SIMD the SIMD is issue-bound
// and cycles/instr = clock * time * SIMDs / wave-instructions.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_costs.hip -o tools/valu_costs
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters)
{
  double d[8];
  unsigned long long q[8];
  unsigned u[8];
  for (int j = 0; j < 8; j++)
  {
    d[j] = 1.0 + threadIdx.x * 1e-3 + j;
    q[j] = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + j);
    u[j] = 12345u * (threadIdx.x + 7 + j);
  }
  const double one = 1.0000001;
  for (int i = 0; i < iters; i++)
  {
#define S_ADD(j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(one));
#define S_MUL(j) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(one));
#define S_FMA(j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[j]) : "v"(one));
#define S_SHL64(j) asm volatile("v_lshlrev_b64 %0, 13, %0" : "+v"(q[j]));
#define S_XOR(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
#define S_ALIGN(j) asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
#define S_CVT(j) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[j]) : "v"(u[j]));
#define S_RSQ(j) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[j]));
#define S_RCP(j) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[j]));
#define S_CMP(j) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(d[j]), "v"(one) : "vcc");
#define S_CND(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 7]) : "vcc");
#define S_LDEXP(j) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d[j]));
#define S_ADD32(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
#define S_FMA32(j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
#define S_PKFMA32(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[j]) : "v"(q[(j + 1) & 7]));
    if (MODE == 0) { CHAIN8(S_ADD) }
    if (MODE == 1) { CHAIN8(S_MUL) }
    if (MODE == 2) { CHAIN8(S_FMA) }
    if (MODE == 3) { CHAIN8(S_SHL64) }
    if (MODE == 4) { CHAIN8(S_XOR) }
    if (MODE == 5) { CHAIN8(S_ALIGN) }
    if (MODE == 6) { CHAIN8(S_CVT) }
    if (MODE == 7) { CHAIN8(S_RSQ) }
    if (MODE == 8) { CHAIN8(S_RCP) }
    if (MODE == 9) { CHAIN8(S_CMP) }
    if (MODE == 10) { CHAIN8(S_CND) }
    if (MODE == 11) { CHAIN8(S_LDEXP) }
    if (MODE == 12) { CHAIN8(S_ADD32) }
    if (MODE == 13) { CHAIN8(S_FMA32) }
    if (MODE == 14) { CHAIN8(S_PKFMA32) }
  }
  double acc = 0;
  for (int j = 0; j < 8; j++)
    acc += d[j] + (double)q[j] + (double)u[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char *name, int cus, double clock_hz)
{
  const int blocks = cus * 8, iters = 4000; // 8 blocks x 4 waves / 4 SIMDs = 8 waves per SIMD
  double *d;
  hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = (double)blocks * 4 * iters * 8;
  const double cyc = ms * 1e-3 * clock_hz * (cus * 4) / wave_instr;
  printf("%-14s %8.3f ms  %6.2f cycles per wave-instruction (at %.2f GHz nominal)\n", name, ms, cyc, clock_hz * 1e-9);
  (void)hipFree(d);
}

int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const double clk = p.clockRate * 1e3;
  printf("%s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
  run<0>("v_add_f64", p.multiProcessorCount, clk);
  run<1>("v_mul_f64", p.multiProcessorCount, clk);
  run<2>("v_fma_f64", p.multiProcessorCount, clk);
  run<3>("v_lshlrev_b64", p.multiProcessorCount, clk);
  run<4>("v_xor_b32", p.multiProcessorCount, clk);
  run<5>("v_alignbit_b32", p.multiProcessorCount, clk);
  run<6>("v_cvt_f64_u32", p.multiProcessorCount, clk);
  run<7>("v_rsq_f64", p.multiProcessorCount, clk);
  run<8>("v_rcp_f64", p.multiProcessorCount, clk);
  run<9>("v_cmp_lt_f64", p.multiProcessorCount, clk);
  run<10>("v_cndmask_b32", p.multiProcessorCount, clk);
  run<11>("v_ldexp_f64", p.multiProcessorCount, clk);
  run<12>("v_add_u32", p.multiProcessorCount, clk);
  run<13>("v_fma_f32", p.multiProcessorCount, clk);
  run<14>("v_pk_fma_f32", p.multiProcessorCount, clk);
  return 0;
}
