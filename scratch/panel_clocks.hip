// phase clocks of k_panel64 (scratch measurement, not part of the product)
#define BAE_PANEL_CLOCKS 1
#include "../ba_amd/csrc/k_chol.hip"
#include <cstdio>
namespace bae { int Engine::fail(hipError_t, const char*) { return -1; } }
#include <vector>
int main() {
  const uint32_t nblk = 94, ld = nblk * 64;
  std::vector<double> h((size_t)(ld + 1) * ld, 0.0);
  for (uint32_t r = 0; r < ld; ++r) {
    for (uint32_t c = 0; c < r; ++c) h[(size_t)r * ld + c] = 0.01 * ((r * 31 + c * 17) % 13 - 6);
    h[(size_t)r * ld + r] = 100.0 + r % 7;
  }
  double *dA, *dsgn, *linvT; int* st;
  hipMalloc(&dA, h.size() * 8); hipMalloc(&dsgn, ld * 8); hipMalloc(&linvT, (size_t)ld * 64 * 8); hipMalloc(&st, 4);
  hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemset(st, 0, 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(bae::k_panel64, dim3(nblk + 1), dim3(256), 0, 0, dA, ld, 0u, nblk, dsgn, linvT, st);
    hipDeviceSynchronize();
    long long c[16];
    hipMemcpyFromSymbol(c, HIP_SYMBOL(bae::g_clk), sizeof(c));
    const char* names[] = {"start", "tile+sync", "panel0", "panel1", "panel2", "panel3", "diaginv", "sync", "subst", "store"};
    printf("rep %d total %lld cycles\n", rep, c[9] - c[0]);
    for (int i = 1; i < 10; ++i) printf("  %-10s %7lld\n", names[i], c[i] - c[i - 1]);
  }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, 0);
  for (int rep = 0; rep < 20; ++rep)
    hipLaunchKernelGGL(bae::k_panel64, dim3(nblk + 1), dim3(256), 0, 0, dA, ld, 0u, nblk, dsgn, linvT, st);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("avg launch %.2f us\n", ms * 1e3 / 20);
  return 0;
}
