// Where does the FP64 GEMM inner loop lose MFMA issue slots?  (scratch)
// The k_update2 chunk loop on synthetic data with its parts switched on one by one:
//   MODE 0: MFMA only, operands in registers
//   MODE 1: + operand fragments read from LDS every k-step (no writes, no barrier)
//   MODE 2: + one __syncthreads per chunk
//   MODE 3: + LDS stores of a staged chunk (register data, no global loads)
//   MODE 4: + global loads of the next chunk (full pipeline)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
static const int NB = 64, KC = 16, LDK = KC + 2;
template <int MODE>
__global__ void __launch_bounds__(256, 4) k(const double* __restrict__ G, double* out, int nchunk, size_t ld) {
  __shared__ double X[2][NB][LDK];
  __shared__ double Y[2][NB][LDK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  const int sr = tid >> 3, sc = (tid & 7) * 2;
  for (int idx = tid; idx < 2 * NB * LDK; idx += 256) { (&X[0][0][0])[idx] = 1.0 + idx * 1e-9; (&Y[0][0][0])[idx] = 1.0 - idx * 1e-9; }
  __syncthreads();
  d4 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* g0 = G + ((size_t)(blockIdx.x % 8) * 64 + sr) * ld + sc;  // 8 row tiles x 8192 columns x 2: L2-resident
  const double* g1 = g0 + 32 * ld;
  double2 p0 = {1.0, 2.0}, p1 = {3.0, 4.0}, q0 = {1.5, 2.5}, q1 = {3.5, 4.5};
  double ra0 = 1.0 + lane * 1e-9, ra1 = 1.1, rb0 = 0.9, rb1 = 1.2;
  for (int kc = 0; kc < nchunk; ++kc) {
    const int b = kc & 1;
    if (MODE >= 3) {
      X[b ^ 1][sr][sc] = p0.x; X[b ^ 1][sr][sc + 1] = p0.y; X[b ^ 1][sr + 32][sc] = p1.x; X[b ^ 1][sr + 32][sc + 1] = p1.y;
      Y[b ^ 1][sr][sc] = q0.x; Y[b ^ 1][sr][sc + 1] = q0.y; Y[b ^ 1][sr + 32][sc] = q1.x; Y[b ^ 1][sr + 32][sc + 1] = q1.y;
    }
    if (MODE >= 4) {
      const size_t k0 = (size_t)(kc % 256) * KC;
      p0 = *reinterpret_cast<const double2*>(g0 + k0); p1 = *reinterpret_cast<const double2*>(g1 + k0);
      q0 = *reinterpret_cast<const double2*>(g0 + k0 + 4096); q1 = *reinterpret_cast<const double2*>(g1 + k0 + 4096);
    }
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      double a0 = ra0, a1 = ra1, b0 = rb0, b1 = rb1;
      if (MODE >= 1) {
        a0 = X[b][rb + li][4 * ks + lk]; a1 = X[b][rb + 16 + li][4 * ks + lk];
        b0 = Y[b][cb + li][4 * ks + lk]; b1 = Y[b][cb + 16 + li][4 * ks + lk];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
  }
  out[(size_t)blockIdx.x * 256 + tid] = acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3];
}
template <int MODE> void run(const double* G, double* out, size_t ld) {
  const int blocks = 256 * 4 * 8, nchunk = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, G, out, 64, ld);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, G, out, nchunk, ld);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * nchunk * 16 * 2048.0;
  printf("mode %d: %8.2f ms  %6.1f TFLOP/s\n", MODE, ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t ld = 8192 + 4096, rows = 65536;  // columns used: < 255*16 + 4096 + 16 < ld
  double *G, *out;
  if (hipMalloc(&G, rows * ld * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, (size_t)256 * 4 * 8 * 256 * 8);
  hipMemset(G, 0, rows * ld * 8);
  run<0>(G, out, ld); run<1>(G, out, ld); run<2>(G, out, ld); run<3>(G, out, ld); run<4>(G, out, ld);
  return 0;
}
