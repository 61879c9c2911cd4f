// Microbenchmark: achievable rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on MI355X.
// Calibrates the FP64 peak bench.py prices the Cholesky against (the local guide lists
// no FP64 figure).  Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double* out, int iters, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma(double* out, int iters, double a0, double b0) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-9 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a0, b0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e-3;
}
int main() {
  double* out; hipMalloc(&out, 256 * 1024 * 8 * 8);
  const int iters = 20000;
  for (int wpb : {256, 512, 1024}) {
    for (int blocks : {256, 512}) {
      double t4 = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(wpb), 0, 0, out, iters, 1.0, 1e-9); });
      double fl = (double)blocks * (wpb / 64) * iters * 4 * 2048.0;
      printf("mfma_f64 16x16x4: blocks %d threads %d acc 4: %.1f TFLOP/s\n", blocks, wpb, fl / t4 / 1e12);
      double t1 = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(wpb), 0, 0, out, iters, 1.0, 1e-9); });
      printf("mfma_f64 16x16x4: blocks %d threads %d acc 1: %.1f TFLOP/s\n", blocks, wpb, (double)blocks * (wpb / 64) * iters * 2048.0 / t1 / 1e12);
    }
  }
  for (int wpb : {256, 1024}) {
    double t = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(512), dim3(wpb), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("v_fma_f64: threads %d: %.1f TFLOP/s\n", wpb, 512.0 * wpb * iters * 8 * 2 / t / 1e12);
  }
  return 0;
}
