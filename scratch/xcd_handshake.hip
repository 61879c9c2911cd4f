// Scratch: cost and reliability of handing a 32 KB tile from one workgroup to workgroups on other XCDs INSIDE a kernel
// (producer writes the tile, raises a flag; consumers spin on the flag, read the tile, verify it, acknowledge).
// The question behind it: can a chain kernel compute a substituted tile once and give it to the other
// workgroups of the same launch?
//   mode 0: plain stores, __threadfence() (agent-scope release: L2 write-back), relaxed atomic flag; consumer: flag,
//           __threadfence() (acquire: L2 invalidate), plain loads
//   mode 1: agent-scope relaxed atomic stores / loads for the tile (write-through, no fence), s_waitcnt + barrier, flag
// Optional background kernel on a second stream that keeps dirtying the L2s (what the bulk update does).
// Every spin is bounded (a time-out is reported, nothing hangs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static const int TILE = 4096;  // doubles
static const long long SPIN_LIMIT = 200000000ll;  // 100 MHz ticks: 2 s

struct Shared {
  unsigned int flag;       // iteration published
  unsigned int acks;       // consumers done with the current iteration (cumulative)
  unsigned long long t_publish;
  unsigned int timeouts, errors;
};

__device__ __forceinline__ double val(int it, int idx) { return (double)(it * 131 + 7) + 1e-3 * idx; }

template <int MODE>
__global__ void __launch_bounds__(256) k_handshake(double* tile, Shared* sh, int iters, unsigned long long* lat, unsigned int* xcc_of) {
  const int tid = threadIdx.x;
  const unsigned int nc = gridDim.x - 1;
  __shared__ int abort_;
  if (tid == 0) { abort_ = 0; xcc_of[blockIdx.x] = (unsigned int)__builtin_amdgcn_s_getreg(6164) & 7u; }
  __syncthreads();
  if (blockIdx.x == 0) {  // producer
    for (int it = 1; it <= iters; ++it) {
      for (int u = 0; u < TILE / 256; ++u) {
        const int idx = u * 256 + tid;
        if (MODE == 0) tile[idx] = val(it, idx);
        else __hip_atomic_store(&tile[idx], val(it, idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();  // (s_waitcnt vmcnt(0) + barrier: every store of the workgroup is acknowledged)
      if (tid == 0) {
        if (MODE == 0) __threadfence();
        sh->t_publish = wall_clock64();
        __hip_atomic_store(&sh->flag, (unsigned int)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wait for the consumers before overwriting the tile
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&sh->acks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)it * nc) {
          __builtin_amdgcn_s_sleep(2);
          if ((long long)(wall_clock64() - t0) > SPIN_LIMIT) { atomicAdd(&sh->timeouts, 1u); abort_ = 1; break; }
        }
      }
      __syncthreads();
      if (abort_) return;
    }
    return;
  }
  // consumers
  for (int it = 1; it <= iters; ++it) {
    unsigned long long t_seen = 0;
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(&sh->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)it) {
        __builtin_amdgcn_s_sleep(1);
        if ((long long)(wall_clock64() - t0) > SPIN_LIMIT) { atomicAdd(&sh->timeouts, 1u); abort_ = 1; break; }
      }
      if (MODE == 0) __threadfence();
    }
    __syncthreads();
    if (abort_) return;
    if (MODE == 0) __threadfence();
    int bad = 0;
    for (int u = 0; u < TILE / 256; ++u) {
      const int idx = u * 256 + tid;
      const double v = (MODE == 0) ? tile[idx] : __hip_atomic_load(&tile[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bad += (v != val(it, idx));
    }
    if (bad) atomicAdd(&sh->errors, (unsigned int)bad);
    __syncthreads();
    if (tid == 0) {
      t_seen = wall_clock64();
      const unsigned long long tp = __hip_atomic_load(&sh->t_publish, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lat[(size_t)(blockIdx.x - 1) * iters + (it - 1)] = t_seen - tp;
      __hip_atomic_fetch_add(&sh->acks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// background: read-modify-write streams over a large buffer (keeps every L2 full of dirty lines)
__global__ void __launch_bounds__(256) k_dirty(double* buf, size_t n, int rounds, const unsigned int* stop) {
  for (int r = 0; r < rounds; ++r) {
    if (*reinterpret_cast<const volatile unsigned int*>(stop)) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = buf[i] * 1.0000001 + 1.0;
  }
}

template <int MODE>
int run(int ncons, int iters, bool background) {
  double* tile; Shared* sh; unsigned long long* lat; unsigned int* xcc; double* big = nullptr; unsigned int* stop;
  CHECK(hipMalloc(&tile, TILE * sizeof(double)));
  CHECK(hipMalloc(&sh, sizeof(Shared)));
  CHECK(hipMemset(sh, 0, sizeof(Shared)));
  CHECK(hipMalloc(&lat, (size_t)ncons * iters * sizeof(unsigned long long)));
  CHECK(hipMalloc(&xcc, (ncons + 1) * sizeof(unsigned int)));
  CHECK(hipMalloc(&stop, sizeof(unsigned int)));
  CHECK(hipMemset(stop, 0, sizeof(unsigned int)));
  hipStream_t s0, s1;
  CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
  const size_t nbig = (size_t)1 << 28;  // 2 GB of doubles
  if (background) {
    CHECK(hipMalloc(&big, nbig * sizeof(double)));
    CHECK(hipMemset(big, 0, nbig * sizeof(double)));
    hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, s1, big, nbig, 1000, (const unsigned int*)stop);
  }
  hipLaunchKernelGGL(k_handshake<MODE>, dim3(ncons + 1), dim3(256), 0, s0, tile, sh, iters, lat, xcc);
  CHECK(hipStreamSynchronize(s0));
  if (background) {
    const unsigned int one = 1;
    CHECK(hipMemcpyAsync(stop, &one, sizeof(one), hipMemcpyHostToDevice, s0));
    CHECK(hipStreamSynchronize(s1));
  }
  Shared h;
  CHECK(hipMemcpy(&h, sh, sizeof(h), hipMemcpyDeviceToHost));
  std::vector<unsigned long long> l((size_t)ncons * iters);
  std::vector<unsigned int> x(ncons + 1);
  CHECK(hipMemcpy(l.data(), lat, l.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(x.data(), xcc, x.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
  double same = 0, other = 0, mx = 0; size_t ns = 0, no = 0;
  for (int c = 0; c < ncons; ++c)
    for (int it = iters / 10; it < iters; ++it) {
      const double us = (double)l[(size_t)c * iters + it] / 100.0;
      if (x[c + 1] == x[0]) { same += us; ++ns; } else { other += us; ++no; }
      if (us > mx) mx = us;
    }
  printf("mode %d  consumers %3d  background %d: hand-off (publish -> tile read and verified) same XCD %.2f us (%zu), other XCDs %.2f us (%zu), max %.1f us; "
         "wrong values %u, time-outs %u\n", MODE, ncons, (int)background, ns ? same / ns : 0.0, ns, no ? other / no : 0.0, no, mx, h.errors, h.timeouts);
  hipFree(tile); hipFree(sh); hipFree(lat); hipFree(xcc); hipFree(stop); if (big) hipFree(big);
  hipStreamDestroy(s0); hipStreamDestroy(s1);
  return 0;
}

int main() {
  for (int bg = 0; bg < 2; ++bg) {
    if (run<0>(15, 2000, bg)) return 1;
    if (run<1>(15, 2000, bg)) return 1;
    if (run<0>(127, 500, bg)) return 1;
    if (run<1>(127, 500, bg)) return 1;
  }
  return 0;
}
