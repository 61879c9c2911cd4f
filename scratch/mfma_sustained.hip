// Sustained FP64 MFMA rate and shader clock on MI355X (scratch): a register-resident
// v_mfma_f64_16x16x4_f64 loop on every SIMD for ~DUR seconds; reports TFLOP/s per launch
// and the shader clock from s_memtime / s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(double* out, long long* clk, int iters) {
  d4 acc[4];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int q = 0; q < 4; ++q) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u & 3], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
  double* out; long long* clk;
  const int blocks = 256 * 4;  // 4 waves per SIMD
  hipMalloc(&out, blocks * 256 * 8); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {2000, 20000, 200000, 600000}) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 16 * 2048.0;
    printf("iters %7d: %8.2f ms  %6.1f TFLOP/s   shader clock %.3f GHz (memtime/realtime)\n", iters, ms,
           flops / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
  }
  return 0;
}
