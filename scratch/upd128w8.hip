// Scratch: 8-wave variant of the 128x128 trailing update (wave tile 64x32, four waves per SIMD at two
// workgroups per CU) against the shipped 4-wave k_update128, dense trailing matrix, no leftovers.
#include "../ba_amd/csrc/k_chol.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace bae;

__global__ void __launch_bounds__(512, 4)
k_update128w8(double* __restrict__ A, uint32_t ld, uint32_t c0, uint32_t m2, uint32_t kb0, uint32_t kb1, uint32_t sbl) {
  __shared__ double X[2][128][LDK2];
  __shared__ double Y[2][128][LDK2];
  const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
  const uint32_t t = (slot >> (2 * sbl)) * 8u + xcd, within = slot & ((1u << (2 * sbl)) - 1u);
  uint32_t sr_ = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((uint64_t)(sr_ + 1) * (sr_ + 2) / 2 <= t) ++sr_;
  while ((uint64_t)sr_ * (sr_ + 1) / 2 > t) --sr_;
  const uint32_t sc_ = t - (uint32_t)((uint64_t)sr_ * (sr_ + 1) / 2);
  const uint32_t R = (sr_ << sbl) + (within >> sbl), C = (sc_ << sbl) + (within & ((1u << sbl) - 1u));
  if (C > R || R >= m2) return;
  const uint32_t c = c0 + 2 * C, i = c0 + 2 * R;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool diag = (i == c);
  const int li = lane & 15, lk = lane >> 4;
  const int rb = 64 * (wave >> 2), cb = 32 * (wave & 3);
  double4_t acc[4][2];
  for (int ti = 0; ti < 4; ++ti) for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int sr = tid >> 3, sc = (tid & 7) * 2;  // 64 rows x 8 double2 per pass
  const double* Xg = A + ((size_t)i * NB + sr) * ld + sc;
  const double* Yg = A + ((size_t)c * NB + sr) * ld + sc;
  const size_t ld64 = (size_t)64 * ld;
  double2 px[2], py[2];
  double FA0[4], FB0[2], FA1[4], FB1[2];
#define W8_SB __builtin_amdgcn_sched_barrier(0)
#define W8_GLOAD(K0) _Pragma("unroll") for (int u = 0; u < 2; ++u) { \
    px[u] = *reinterpret_cast<const double2*>(Xg + u * ld64 + (K0)); py[u] = *reinterpret_cast<const double2*>(Yg + u * ld64 + (K0)); }
#define W8_SSTORE(B) _Pragma("unroll") for (int u = 0; u < 2; ++u) { \
    X[B][sr + 64 * u][sc] = px[u].x; X[B][sr + 64 * u][sc + 1] = px[u].y; Y[B][sr + 64 * u][sc] = py[u].x; Y[B][sr + 64 * u][sc + 1] = py[u].y; }
#define W8_LDF(FA, FB, B, KS) { _Pragma("unroll") for (int q = 0; q < 4; ++q) FA[q] = X[B][rb + 16 * q + li][4 * (KS) + lk]; \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) FB[q] = Y[B][cb + 16 * q + li][4 * (KS) + lk]; }
#define W8_MM(FA, FB) _Pragma("unroll") for (int ti = 0; ti < 4; ++ti) _Pragma("unroll") for (int tj = 0; tj < 2; ++tj) \
    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[ti], FB[tj], acc[ti][tj], 0, 0, 0);
#define W8_CHUNK(J, STORE, LOADK, NEXT) { if (STORE) W8_SSTORE(((J) + 1) & 1); LOADK; \
    W8_SB; W8_LDF(FA1, FB1, (J) & 1, 1); W8_SB; W8_MM(FA0, FB0); W8_SB; \
    W8_LDF(FA0, FB0, (J) & 1, 2); W8_SB; W8_MM(FA1, FB1); W8_SB; \
    W8_LDF(FA1, FB1, (J) & 1, 3); W8_SB; W8_MM(FA0, FB0); W8_SB; \
    __syncthreads(); W8_SB; if (NEXT) W8_LDF(FA0, FB0, ((J) + 1) & 1, 0); W8_SB; W8_MM(FA1, FB1); W8_SB; }
  uint32_t kb = kb0;
  uint32_t kcur = kb * NB;
  W8_GLOAD(kcur); W8_SSTORE(0); W8_GLOAD(kcur + KC2);
  __syncthreads();
  W8_LDF(FA0, FB0, 0, 0);
  for (++kb; kb < kb1; ++kb) {
    const uint32_t knext = kb * NB;
    W8_CHUNK(0, true, W8_GLOAD(kcur + 2 * KC2), true);
    W8_CHUNK(1, true, W8_GLOAD(kcur + 3 * KC2), true);
    W8_CHUNK(2, true, W8_GLOAD(knext), true);
    W8_CHUNK(3, true, W8_GLOAD(knext + KC2), true);
    kcur = knext;
  }
  W8_CHUNK(0, true, W8_GLOAD(kcur + 2 * KC2), true);
  W8_CHUNK(1, true, W8_GLOAD(kcur + 3 * KC2), true);
  W8_CHUNK(2, true, , true);
  W8_CHUNK(3, false, , false);
  if (diag && cb > rb + 63) return;
  double* Aic = A + ((size_t)i * NB) * ld + (size_t)c * NB;
  for (int ti = 0; ti < 4; ++ti) {
    double4_t cv[2];
    for (int tj = 0; tj < 2; ++tj) for (int reg = 0; reg < 4; ++reg)
      cv[tj][reg] = Aic[(size_t)(rb + 16 * ti + lk + 4 * reg) * ld + cb + 16 * tj + li];
    for (int tj = 0; tj < 2; ++tj) for (int reg = 0; reg < 4; ++reg) {
      const int r = rb + 16 * ti + lk + 4 * reg, cc = cb + 16 * tj + li;
      if (!diag || cc <= r) Aic[(size_t)r * ld + cc] = cv[tj][reg] - acc[ti][tj][reg];
    }
  }
}

__global__ void k_fill(double* A, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t z = i * 0x9E3779B97F4A7C15ull + 12345; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    A[i] = ((double)(z & 0xFFFFFF) / 16777216.0 - 0.5) * 1e-2;
  }
}
int main(int argc, char** argv) {
  const uint32_t nblk = argc > 1 ? atoi(argv[1]) : 520, KOUT = argc > 2 ? atoi(argv[2]) : 16;
  const int variant = argc > 3 ? atoi(argv[3]) : 8;
  const uint32_t ld = nblk * NB;
  const size_t rows = (size_t)nblk * NB + 1;
  double *A, *dsgn; int* colneg;
  if (hipMalloc(&A, rows * ld * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&dsgn, ld * 8); hipMalloc(&colneg, nblk * 4);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, rows * ld);
  std::vector<double> ones(ld, 1.0);
  hipMemcpy(dsgn, ones.data(), ld * 8, hipMemcpyHostToDevice);
  hipMemset(colneg, 0, nblk * 4);
  const uint32_t a_end = KOUT, m = nblk - a_end, m2 = m / 2, sbl2 = 2, sbe2 = 4;
  const uint32_t nsr = (m2 + sbe2 - 1) / sbe2, nsb = nsr * (nsr + 1) / 2, grid = ((nsb + 7) / 8) * 8 * sbe2 * sbe2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int REPS = 11;
  for (int rep = 0; rep < REPS; ++rep) {
    if (rep == 1) hipEventRecord(e0, 0);
    if (variant == 8)
      hipLaunchKernelGGL(k_update128w8, dim3(grid), dim3(512), 0, 0, A, ld, a_end, m2, 0u, KOUT, sbl2);
    else
      hipLaunchKernelGGL(k_update128<false>, dim3(grid), dim3(256), 0, 0, A, ld, nblk, a_end, m2, 0u, KOUT, (const double*)dsgn,
                         (const int*)colneg, sbl2, (const uint8_t*)nullptr, 0u, 1u, 1u, 0u, a_end, 0u);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= (REPS - 1);
  const double tiles = 4.0 * ((double)m2 * (m2 + 1) / 2);
  printf("nblk %u KOUT %u %d-wave: %.3f ms  %.1f TFLOP/s\n", nblk, KOUT, variant, ms, tiles * 2.0 * 64 * 64 * 64 * KOUT / ms / 1e9);
  if (getenv("CHECK")) {
    std::vector<double> h(rows * ld);
    hipMemcpy(h.data(), A, rows * ld * 8, hipMemcpyDeviceToHost);
    uint64_t x = 0; for (size_t k = 0; k < h.size(); ++k) { uint64_t bb; memcpy(&bb, &h[k], 8); x = (x ^ bb) * 0x100000001B3ull + (x >> 7); }
    printf("checksum %016llx\n", (unsigned long long)x);
  }
  return 0;
}
