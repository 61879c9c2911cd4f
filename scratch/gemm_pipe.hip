// Scratch: does an explicitly software-pipelined fragment schedule lift the 64x64-tile FP64 GEMM
// loop above the compiler-scheduled one (gemm_steps.hip mode 4)?
//   V0: compiler-scheduled loop (= gemm_steps mode 4)
//   V1: fragments of k-step s+1 are read while the MFMAs of k-step s run (two register sets);
//       the single barrier of a chunk sits between k-steps 2 and 3, so the first fragments of
//       the next chunk are read under the last MFMAs of this one
//   V2: V1 on a 128x128 tile (4 waves x 64x64)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
static const int NB = 64, KC = 16, LDK = KC + 2;
#define SB __builtin_amdgcn_sched_barrier(0)
#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int V>
__global__ void __launch_bounds__(256, 4) k64(const double* __restrict__ G, double* out, int nchunk, size_t ld) {
  __shared__ double X[2][NB][LDK];
  __shared__ double Y[2][NB][LDK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  const int sr = tid >> 3, sc = (tid & 7) * 2;
  for (int idx = tid; idx < 2 * NB * LDK; idx += 256) { (&X[0][0][0])[idx] = 1.0 + idx * 1e-9; (&Y[0][0][0])[idx] = 1.0 - idx * 1e-9; }
  __syncthreads();
  d4 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* g0 = G + ((size_t)(blockIdx.x % 8) * 64 + sr) * ld + sc;
  const double* g1 = g0 + 32 * ld;
  double2 p0 = {1.0, 2.0}, p1 = {3.0, 4.0}, q0 = {1.5, 2.5}, q1 = {3.5, 4.5};
#define SSTORE(B)                                                                                              \
  X[B][sr][sc] = p0.x; X[B][sr][sc + 1] = p0.y; X[B][sr + 32][sc] = p1.x; X[B][sr + 32][sc + 1] = p1.y;         \
  Y[B][sr][sc] = q0.x; Y[B][sr][sc + 1] = q0.y; Y[B][sr + 32][sc] = q1.x; Y[B][sr + 32][sc + 1] = q1.y;
#define GLOAD(KC_)                                                                                             \
  { const size_t k0 = (size_t)((KC_) % 256) * KC;                                                              \
    p0 = *reinterpret_cast<const double2*>(g0 + k0); p1 = *reinterpret_cast<const double2*>(g1 + k0);          \
    q0 = *reinterpret_cast<const double2*>(g0 + k0 + 4096); q1 = *reinterpret_cast<const double2*>(g1 + k0 + 4096); }
#define LDF(F, B, KS)                                                                                          \
  F[0] = X[B][rb + li][4 * (KS) + lk]; F[1] = X[B][rb + 16 + li][4 * (KS) + lk];                               \
  F[2] = Y[B][cb + li][4 * (KS) + lk]; F[3] = Y[B][cb + 16 + li][4 * (KS) + lk];
#define MM(F) MFMA(F[0], F[2], acc[0][0]); MFMA(F[0], F[3], acc[0][1]); MFMA(F[1], F[2], acc[1][0]); MFMA(F[1], F[3], acc[1][1]);
  if (V == 0) {
    for (int kc = 0; kc < nchunk; kc += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        SSTORE(h ^ 1);
        GLOAD(kc + h);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { double F[4]; LDF(F, h, ks); MM(F); }
        __syncthreads();
      }
    }
  } else {
    double F0[4], F1[4];
    LDF(F0, 0, 0);
    for (int kc = 0; kc < nchunk; kc += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        SSTORE(h ^ 1);
        GLOAD(kc + h);
        SB; LDF(F1, h, 1); SB; MM(F0); SB;
        LDF(F0, h, 2); SB; MM(F1); SB;
        LDF(F1, h, 3); SB; MM(F0); SB;
        __syncthreads(); SB;
        LDF(F0, h ^ 1, 0); SB; MM(F1); SB;
      }
    }
    acc[0][0][0] += F0[0] + F0[1] + F0[2] + F0[3];
  }
  out[(size_t)(blockIdx.x % 8192) * 256 + tid] = acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3];
#undef SSTORE
#undef GLOAD
#undef LDF
#undef MM
}

// 128x128 tile, 4 waves x 64x64, pipelined
__global__ void __launch_bounds__(256, 2) k128(const double* __restrict__ G, double* out, int nchunk, size_t ld) {
  __shared__ double X[2][128][LDK];
  __shared__ double Y[2][128][LDK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int rb = 64 * (wave >> 1), cb = 64 * (wave & 1);
  const int sr = tid >> 3, sc = (tid & 7) * 2;
  for (int idx = tid; idx < 2 * 128 * LDK; idx += 256) { (&X[0][0][0])[idx] = 1.0 + idx * 1e-9; (&Y[0][0][0])[idx] = 1.0 - idx * 1e-9; }
  __syncthreads();
  d4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* g0 = G + ((size_t)(blockIdx.x % 8) * 128 + sr) * ld + sc;
  double2 p[4], q[4];
  for (int u = 0; u < 4; ++u) { p[u] = make_double2(1.0 + u, 2.0); q[u] = make_double2(1.5, 2.5 + u); }
#define SSTORE(B)                                                                                  \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                  \
    X[B][sr + 32 * u][sc] = p[u].x; X[B][sr + 32 * u][sc + 1] = p[u].y;                            \
    Y[B][sr + 32 * u][sc] = q[u].x; Y[B][sr + 32 * u][sc + 1] = q[u].y; }
#define GLOAD(KC_)                                                                                 \
  { const size_t k0 = (size_t)((KC_) % 256) * KC;                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                \
      p[u] = *reinterpret_cast<const double2*>(g0 + (size_t)32 * u * ld + k0);                     \
      q[u] = *reinterpret_cast<const double2*>(g0 + (size_t)32 * u * ld + k0 + 4096); } }
#define LDF(FA, FB, B, KS)                                                                         \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                  \
    FA[t] = X[B][rb + 16 * t + li][4 * (KS) + lk]; FB[t] = Y[B][cb + 16 * t + li][4 * (KS) + lk]; }
#define MM(FA, FB)                                                                                 \
  _Pragma("unroll") for (int ti = 0; ti < 4; ++ti)                                                 \
    _Pragma("unroll") for (int tj = 0; tj < 4; ++tj) MFMA(FA[ti], FB[tj], acc[ti][tj]);
  double A0[4], B0[4], A1[4], B1[4];
  LDF(A0, B0, 0, 0);
  for (int kc = 0; kc < nchunk; kc += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      SSTORE(h ^ 1);
      GLOAD(kc + h);
      SB; LDF(A1, B1, h, 1); SB; MM(A0, B0); SB;
      LDF(A0, B0, h, 2); SB; MM(A1, B1); SB;
      LDF(A1, B1, h, 3); SB; MM(A0, B0); SB;
      __syncthreads(); SB;
      LDF(A0, B0, h ^ 1, 0); SB; MM(A1, B1); SB;
    }
  }
  double s = A0[0] + B0[0];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int V> void run64(const double* G, double* out, size_t ld, int blocks = 256 * 4 * 8, int nchunk = 2048) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k64<V>, dim3(blocks), dim3(256), 0, 0, G, out, 64, ld);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k64<V>, dim3(blocks), dim3(256), 0, 0, G, out, nchunk, ld);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * nchunk * 16 * 2048.0;
  printf("64x64  V%d blocks %d nchunk %d: %8.2f ms  %6.1f TFLOP/s\n", V, blocks, nchunk, ms, flops / (ms * 1e-3) / 1e12);
}
void run128(const double* G, double* out, size_t ld) {
  const int blocks = 256 * 2 * 8, nchunk = 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k128, dim3(blocks), dim3(256), 0, 0, G, out, 32, ld);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k128, dim3(blocks), dim3(256), 0, 0, G, out, nchunk, ld);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * nchunk * 64 * 2048.0;
  printf("128x128 pipelined: %8.2f ms  %6.1f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t ld = 8192 + 4096, rows = 65536;  // rows used < 8*128 + 128; columns < 255*16 + 4096 + 16 < ld
  double *G, *out;
  if (hipMalloc(&G, rows * ld * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, (size_t)256 * 4 * 8 * 256 * 8);
  hipMemset(G, 0, rows * ld * 8);
  run64<0>(G, out, ld); run64<1>(G, out, ld); run128(G, out, ld);
  run64<0>(G, out, ld, 131072, 128); run64<0>(G, out, ld, 131072 * 4, 32); run64<1>(G, out, ld, 131072 * 4, 32);
  return 0;
}
