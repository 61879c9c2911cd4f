// Scratch: the shipped trailing-update kernel (k_update2) alone on a dense synthetic trailing matrix —
// no look-ahead neighbour, no sparsity — to separate kernel efficiency from schedule effects.
#include "../ba_amd/csrc/k_chol.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace bae;
__global__ void k_fill(double* A, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t z = i * 0x9E3779B97F4A7C15ull + 12345; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    A[i] = ((double)(z & 0xFFFFFF) / 16777216.0 - 0.5) * 1e-2;
  }
}
int main(int argc, char** argv) {
  const uint32_t nblk = argc > 1 ? atoi(argv[1]) : 264, KOUT = argc > 2 ? atoi(argv[2]) : 8;
  const int full = argc > 3 ? atoi(argv[3]) : 1;
  const uint32_t ld = nblk * NB;
  const size_t rows = (size_t)nblk * NB + 1;
  double *A, *dsgn; int* colneg;
  if (hipMalloc(&A, rows * ld * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&dsgn, ld * 8); hipMalloc(&colneg, nblk * 4);
  if (getenv("ZERO")) hipMemset(A, 0, rows * ld * 8); else hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, rows * ld);
  std::vector<double> ones(ld, 1.0);
  hipMemcpy(dsgn, ones.data(), ld * 8, hipMemcpyHostToDevice);
  hipMemset(colneg, 0, nblk * 4);
  const uint32_t a_end = KOUT, m = nblk - a_end, sbl = 3, sbe = 8;
  const uint32_t nsr = (m + 1 + sbe - 1) / sbe, nsb = nsr * (nsr + 1) / 2, grid1 = ((nsb + 7) / 8) * 8 * sbe * sbe;
  const int swzf = 1 | (int)(sbl << 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int REPS = getenv("REPS") ? atoi(getenv("REPS")) : 4;
  for (int rep = 0; rep < REPS; ++rep) {
    if (rep == 1) hipEventRecord(e0, 0);
    if (full == 2)
      launch_bulk_update(nullptr, 0, A, ld, nblk, a_end, 0u, KOUT, dsgn, colneg, nullptr, true, 0u, 1u, 1u);
    else if (full)
      hipLaunchKernelGGL(k_update2<false>, dim3(grid1), dim3(256), 0, 0, A, ld, nblk, a_end, 0u, KOUT, (const double*)dsgn, (const int*)colneg, swzf, (const uint8_t*)nullptr, 0u, 1u, 1u);
    else
      hipLaunchKernelGGL(k_update2<true>, dim3(grid1), dim3(256), 0, 0, A, ld, nblk, a_end, 0u, KOUT, (const double*)dsgn, (const int*)colneg, swzf, (const uint8_t*)nullptr, 0u, 1u, 1u);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= (REPS - 1);
  const double tiles = (double)m * (m + 1) / 2 + m;
  printf("nblk %u (n=%u) KOUT %u %s: %.3f ms  %.1f TFLOP/s (tiles %.0f)\n", nblk, ld, KOUT, full == 2 ? "128" : full ? "full" : "capped", ms, tiles * 2.0 * 64 * 64 * 64 * KOUT / ms / 1e9, tiles);
  if (getenv("CHECK")) {
    std::vector<double> h(rows * ld);
    hipMemcpy(h.data(), A, rows * ld * 8, hipMemcpyDeviceToHost);
    uint64_t x = 0; double sum = 0;
    for (size_t k = 0; k < h.size(); ++k) { uint64_t b; memcpy(&b, &h[k], 8); x = (x ^ b) * 0x100000001B3ull + (x >> 7); sum += h[k]; }
    printf("checksum %016llx sum %.17g\n", (unsigned long long)x, sum);
  }
  return 0;
}
