// instruction latency microbenchmarks on gfx950 (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long tick(double& x) {
  asm volatile("" : "+v"(x));
  __builtin_amdgcn_sched_barrier(0);
  long long t;
  asm volatile("s_nop 7\n s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(x));
  return t;
}
__global__ void k(double* out, long long* clk, double seed) {
  double x = seed + threadIdx.x * 1e-3;
  long long t0, t1;
  // 1. dependent fma chain
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = fma(x, 1.0000001, 1e-9);
  t1 = tick(x);
  if (threadIdx.x == 0) clk[0] = t1 - t0;
  // 2. dependent rsq chain
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rsq(x + 2.0);
  t1 = tick(x);
  if (threadIdx.x == 0) clk[1] = t1 - t0;
  // 3. readlane -> fma -> readlane chain
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) { double s = readlane_f64(x, i & 63); x = fma(x, s, 1e-9); }
  t1 = tick(x);
  if (threadIdx.x == 0) clk[2] = t1 - t0;
  // 4. independent fmas (8 accumulators)
  double a[8];
  for (int q = 0; q < 8; ++q) a[q] = x + q;
  for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(a[q]));
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) a[i & 7] = fma(a[i & 7], 1.0000001, 1e-9);
  for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(a[q]));
  t1 = tick(x);
  if (threadIdx.x == 0) clk[3] = t1 - t0;
  for (int q = 0; q < 8; ++q) x += a[q];
  // 5. dependent mul chain
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = x * 1.0000001;
  t1 = tick(x);
  if (threadIdx.x == 0) clk[4] = t1 - t0;
  // 6. LDS write -> uniform read -> fma chain
  __shared__ double buf[64];
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 32; ++i) { buf[threadIdx.x] = x; double s = buf[i]; x = fma(x, 1e-9, s); }
  t1 = tick(x);
  if (threadIdx.x == 0) clk[5] = t1 - t0;
  // 7. dependent MFMA chain
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc = {x, x, x, x};
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 32; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-3, acc, 0, 0, 0);
  x += acc[0];
  t1 = tick(x);
  if (threadIdx.x == 0) clk[6] = t1 - t0;
  // 8. independent MFMA (4 accs)
  d4 ac[4] = {{x,x,x,x},{x,x,x,x},{x,x,x,x},{x,x,x,x}};
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < 32; ++i) ac[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-3, ac[i & 3], 0, 0, 0);
  x += ac[0][0] + ac[1][0] + ac[2][0] + ac[3][0];
  t1 = tick(x);
  if (threadIdx.x == 0) clk[7] = t1 - t0;
  out[threadIdx.x] = x;
}
int main() {
  double* out; long long* clk;
  hipMalloc(&out, 64 * 8); hipMalloc(&clk, 16 * 8);
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, clk, 1.5); hipDeviceSynchronize(); }
  long long h[16]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  printf("dep fma        %.1f cyc/op\n", h[0] / 64.0);
  printf("dep rsq(+add)  %.1f cyc/iter\n", h[1] / 64.0);
  printf("readlane+fma   %.1f cyc/iter\n", h[2] / 64.0);
  printf("indep fma x8   %.1f cyc/op\n", h[3] / 64.0);
  printf("dep mul        %.1f cyc/op\n", h[4] / 64.0);
  printf("lds wr+rd+fma  %.1f cyc/iter\n", h[5] / 32.0);
  printf("dep mfma f64   %.1f cyc/op\n", h[6] / 32.0);
  printf("indep mfma x4  %.1f cyc/op\n", h[7] / 32.0);
  return 0;
}
