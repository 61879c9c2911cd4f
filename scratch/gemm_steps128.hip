// Ceiling of a 128x128-tile FP64 GEMM chunk loop (4 waves x 64x64, 16 MFMA tiles per k-step),
// parts switched on one by one as in gemm_steps.hip (scratch).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
static const int KC = 16, LDK = KC + 2;
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(double* out, int nchunk) {
  __shared__ double X[2][128][LDK];
  __shared__ double Y[2][128][LDK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int rb = 64 * (wave >> 1), cb = 64 * (wave & 1);
  const int sr = tid >> 3, sc = (tid & 7) * 2;
  for (int idx = tid; idx < 2 * 128 * LDK; idx += 256) { (&X[0][0][0])[idx] = 1.0 + idx * 1e-9; (&Y[0][0][0])[idx] = 1.0 - idx * 1e-9; }
  __syncthreads();
  d4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  double2 p[4] = {{1.0, 2.0}, {3.0, 4.0}, {1.5, 2.5}, {3.5, 4.5}}, q[4] = {{1.0, 2.0}, {3.0, 4.0}, {1.5, 2.5}, {3.5, 4.5}};
  for (int kc = 0; kc < nchunk; ++kc) {
    const int b = kc & 1;
    if (MODE >= 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        X[b ^ 1][sr + 32 * u][sc] = p[u].x; X[b ^ 1][sr + 32 * u][sc + 1] = p[u].y;
        Y[b ^ 1][sr + 32 * u][sc] = q[u].x; Y[b ^ 1][sr + 32 * u][sc + 1] = q[u].y;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      double a[4], bb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = MODE >= 1 ? X[b][rb + 16 * t + li][4 * ks + lk] : 1.0 + t;
        bb[t] = MODE >= 1 ? Y[b][cb + 16 * t + li][4 * ks + lk] : 1.0 - t;
      }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], bb[tj], acc[ti][tj], 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
  }
  double s = 0;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(double* out) {
  const int blocks = 256 * 2 * 8, nchunk = 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 32);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, nchunk);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * nchunk * 64 * 2048.0;
  printf("128x128 mode %d: %8.2f ms  %6.1f TFLOP/s\n", MODE, ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
  double* out;
  hipMalloc(&out, (size_t)256 * 2 * 8 * 256 * 8);
  run<0>(out); run<1>(out); run<2>(out); run<3>(out);
  return 0;
}
