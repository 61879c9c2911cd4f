// Microbenchmark (round 3): do v_mfma_f64_16x16x4_f64 and v_fma_f64 share an execution resource on MI355X?
// (a) one wave issues both kinds interleaved; (b) even waves issue MFMAs, odd waves VALU FMAs.  If the sum of the two
// rates exceeds what either reaches alone, the matrix and vector FP64 pipes are separate.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_concurrency.hip -o mfma_valu_concurrency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ void k_mix(double* out, int iters, double a0, double b0, int split) {
  double4_t acc[NM > 0 ? NM : 1];
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (double4_t){0, 0, 0, 0};
  double x[NV > 0 ? NV : 1];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) x[i] = threadIdx.x * 1e-9 + i;
  const double a = a0 + threadIdx.x * 1e-9, b = b0;
  const int wave = threadIdx.x >> 6;
  const bool do_m = !split || (wave & 1) == 0, do_v = !split || (wave & 1) == 1;
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int i = 0; i < NV; ++i) x[i] = fma(x[i], a0, b0);
    }
  }
  double s = 0;
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e-3;
}

template <int NM, int NV>
void run(const char* name, double* out, int split) {
  const int iters = 20000, blocks = 512, threads = 512;
  const double t = timeit([&] { hipLaunchKernelGGL((k_mix<NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9, split); });
  const double waves = (double)blocks * (threads / 64);
  const double fm = waves * (split ? 0.5 : 1.0) * iters * NM * 2048.0;
  const double fv = waves * (split ? 0.5 : 1.0) * iters * NV * 64 * 2.0;
  printf("%-44s %6.2f ms  mfma %6.1f  valu %6.1f  total %6.1f TFLOP/s\n", name, t * 1e3, fm / t / 1e12, fv / t / 1e12, (fm + fv) / t / 1e12);
}

int main() {
  double* out; hipMalloc(&out, 512 * 1024 * 8);
  run<4, 0>("mfma only (4 per iteration)", out, 0);
  run<0, 32>("valu fma only (32 per iteration)", out, 0);
  run<4, 32>("same wave: 4 mfma + 32 fma", out, 0);     // 4 * 2048 = 8192 flop vs 32 * 128 = 4096 flop
  run<4, 64>("same wave: 4 mfma + 64 fma", out, 0);
  run<4, 16>("same wave: 4 mfma + 16 fma", out, 0);
  run<4, 32>("split waves: even mfma x4, odd fma x32", out, 1);
  run<4, 64>("split waves: even mfma x4, odd fma x64", out, 1);
  return 0;
}
