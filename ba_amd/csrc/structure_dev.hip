// Device-side build of the static structure (ba_hip_finalize): the same lists as structure.h's
// host builder — observation CSR, linearisation wave ranges, rank-1 term list sorted by tile, tile
// references, per-pose terms — produced on the GPU from the raw residual arrays: stable radix sorts
// (rocPRIM, a one-off setup step, not part of an iteration), scans and simple generation kernels that
// mirror the host loops one to one, so that every list has the same ORDER as the host version and S
// comes out bitwise identical (test_structure_built_on_device_equals_host_build).  The host builder
// stays as the executable specification (CPU test tests/test_structure_lists.py) and as a fallback
// (BA_HIP_HOST_STRUCTURE=1).
//
// Replaces the bookkeeping of the reference's BuildProblem (sorted block insertion,
// BundleAdjuster.cpp:1552-1802) — 1.5 s of host work at BASELINE configs[3], now ~0.1 s.
#include "engine.h"

#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>

namespace bae {

namespace {

struct DevGraph {  // everything the generation kernels read
  int LM, D;
  uint32_t O, L, R, WO, lrow_base;
  const uint32_t* obs_pose;   // sorted by landmark
  const uint32_t* obs_lm;
  const uint32_t* lm_ptr;
  const uint32_t* lm_ref_pose;
  const int32_t* pose_opt;
  const int32_t* lm_opt;
};

__device__ __forceinline__ bool d_listed(const DevGraph& g, uint32_t s) {
  return g.LM != 1 || g.obs_pose[s] != g.lm_ref_pose[g.obs_lm[s]];
}
__device__ __forceinline__ int d_meas_opt(const DevGraph& g, uint32_t s) {
  return d_listed(g, s) ? g.pose_opt[g.obs_pose[s]] : -1;
}
__device__ __forceinline__ int d_ref_opt(const DevGraph& g, uint32_t s) {
  return (g.LM == 1 && d_listed(g, s)) ? g.pose_opt[g.lm_ref_pose[g.obs_lm[s]]] : -1;
}

// key of the block (i, j), i < j: (home tile << 12) | row offset << 6 | column offset (structure.h)
__device__ __forceinline__ unsigned long long d_home_key(uint32_t i, uint32_t j, int D) {
  const uint32_t r = j * (uint32_t)D, c = i * (uint32_t)D;
  const unsigned long long tr = r / 64, tc = c / 64;
  return ((tr * (tr + 1) / 2 + tc) << 12) | ((unsigned long long)(r % 64) << 6) | (c % 64);
}

__global__ void k_iota(uint32_t n, uint32_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}
__global__ void k_count_lm(uint32_t O, const uint32_t* __restrict__ lm, uint32_t* __restrict__ cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < O) atomicAdd(&cnt[lm[i]], 1u);
}
// sorted position s <- residual perm[s]
__global__ void k_gather_obs(uint32_t O, const uint32_t* __restrict__ perm, const double* __restrict__ z,
                             const double* __restrict__ w, const uint32_t* __restrict__ pose,
                             const uint32_t* __restrict__ lm, const uint32_t* __restrict__ cam,
                             const uint8_t* __restrict__ is_cond, double* __restrict__ oz, double* __restrict__ ow0,
                             double* __restrict__ ow, uint32_t* __restrict__ opose, uint32_t* __restrict__ olm,
                             uint32_t* __restrict__ ocam, uint8_t* __restrict__ ocond) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= O) return;
  const uint32_t a = perm[s];
  oz[2 * (size_t)s] = z[2 * (size_t)a]; oz[2 * (size_t)s + 1] = z[2 * (size_t)a + 1];
  ow0[s] = w[a]; ow[s] = w[a];
  opose[s] = pose[a]; olm[s] = lm[a]; ocam[s] = cam[a];
  ocond[s] = is_cond[a];
}

// ---- incidences: (pose opt id, first W row) per landmark, landmark-major ------------------------
__global__ void k_inc_count(DevGraph g, uint32_t* __restrict__ cnt) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= g.L) return;
  uint32_t m = 0;
  if (g.lm_opt[l] >= 0) {
    bool any = false;
    for (uint32_t s = g.lm_ptr[l]; s < g.lm_ptr[l + 1]; ++s) {
      if (!d_listed(g, s)) continue;
      any = true;
      if (g.pose_opt[g.obs_pose[s]] >= 0) ++m;
    }
    if (g.LM == 1 && any && g.pose_opt[g.lm_ref_pose[l]] >= 0) ++m;
  }
  cnt[l] = m;
}
__global__ void k_inc_fill(DevGraph g, const uint32_t* __restrict__ linc_ptr, uint32_t* __restrict__ inc_pose,
                           uint32_t* __restrict__ inc_wrow) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= g.L || g.lm_opt[l] < 0) return;
  uint32_t w = linc_ptr[l];
  bool any = false;
  for (uint32_t s = g.lm_ptr[l]; s < g.lm_ptr[l + 1]; ++s) {
    if (!d_listed(g, s)) continue;
    any = true;
    const int po = g.pose_opt[g.obs_pose[s]];
    if (po >= 0) { inc_pose[w] = (uint32_t)po; inc_wrow[w] = s * g.R + g.WO; ++w; }
  }
  if (g.LM == 1 && any) {
    const int ro = g.pose_opt[g.lm_ref_pose[l]];
    if (ro >= 0) { inc_pose[w] = (uint32_t)ro; inc_wrow[w] = g.lrow_base + 2 * l; ++w; }
  }
}

// ---- off-diagonal rank-1 terms -------------------------------------------------------------------
// per landmark: LM records per pair of incidences on DIFFERENT poses; per-pose Schur terms: LM per
// incidence + 2 LM per pair on the SAME pose
__global__ void k_lm_counts(DevGraph g, const uint32_t* __restrict__ linc_ptr, const uint32_t* __restrict__ inc_pose,
                            uint32_t* __restrict__ rec_cnt, uint32_t* __restrict__ schur_cnt) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= g.L) return;
  const uint32_t q0 = linc_ptr[l], q1 = linc_ptr[l + 1];
  uint32_t diff = 0, same = 0;
  for (uint32_t x = q0; x < q1; ++x)
    for (uint32_t y = x + 1; y < q1; ++y) (inc_pose[x] != inc_pose[y] ? diff : same) += 1;
  rec_cnt[l] = diff * (uint32_t)g.LM;
  schur_cnt[l] = ((q1 - q0) + 2 * same) * (uint32_t)g.LM;
}
__global__ void k_obs_counts(DevGraph g, uint32_t* __restrict__ rec_cnt, uint32_t* __restrict__ jterm_cnt) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= g.O) return;
  const int m = d_meas_opt(g, s), r = d_ref_opt(g, s);
  rec_cnt[s] = (m >= 0 && r >= 0 && m != r) ? 2u : 0u;
  jterm_cnt[s] = (m >= 0 ? 2u : 0u) + (r >= 0 ? 2u : 0u);
}
__global__ void k_lm_records(DevGraph g, const uint32_t* __restrict__ linc_ptr, const uint32_t* __restrict__ inc_pose,
                             const uint32_t* __restrict__ inc_wrow, const uint32_t* __restrict__ rec_off,
                             unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= g.L) return;
  const uint32_t q0 = linc_ptr[l], q1 = linc_ptr[l + 1];
  uint32_t w = rec_off[l];
  for (uint32_t x = q0; x < q1; ++x)
    for (uint32_t y = x + 1; y < q1; ++y) {
      const uint32_t px = inc_pose[x], py = inc_pose[y];
      if (px == py) continue;
      const uint32_t lo = px < py ? x : y, hi = px < py ? y : x;  // block row side i = the smaller opt id
      const unsigned long long key = d_home_key(inc_pose[lo], inc_pose[hi], g.D);
      for (int k = 0; k < g.LM; ++k) {
        keys[w] = key;
        vals[w] = (unsigned long long)(inc_wrow[lo] + g.LM + k) | ((unsigned long long)(inc_wrow[hi] + k) << 32);
        ++w;
      }
    }
}
__global__ void k_obs_records(DevGraph g, const uint32_t* __restrict__ rec_off, uint32_t base,
                              unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= g.O) return;
  const int m = d_meas_opt(g, s), r = d_ref_opt(g, s);
  if (!(m >= 0 && r >= 0 && m != r)) return;
  const uint32_t jm = s * g.R, jr = s * g.R + 2;
  uint32_t w = base + rec_off[s];
  for (uint32_t k = 0; k < 2; ++k, ++w) {  // J_i^T J_j over the u and v rows
    if (m < r) { keys[w] = d_home_key((uint32_t)m, (uint32_t)r, g.D); vals[w] = (unsigned long long)(jm + k) | ((unsigned long long)(jr + k) << 32); }
    else { keys[w] = d_home_key((uint32_t)r, (uint32_t)m, g.D); vals[w] = (unsigned long long)(jr + k) | ((unsigned long long)(jm + k) << 32); }
  }
}
__global__ void k_block_flags(uint32_t n, const unsigned long long* __restrict__ keys, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_block_starts(uint32_t n, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ idx,
                               uint32_t* __restrict__ starts, uint32_t n_blocks) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) starts[idx[i]] = i;
  if (i == 0) starts[n_blocks] = n;
}
// every tile a block touches (structure.h: for_tiles); pass 0 counts, pass 1 fills
template <int PASS>
__global__ void k_tile_refs(uint32_t n_blocks, uint32_t nt, const uint32_t* __restrict__ starts,
                            const unsigned long long* __restrict__ keys, uint32_t* __restrict__ cursor,
                            uint2* __restrict__ refs, int* __restrict__ overflow) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_blocks) return;
  const uint32_t e0 = starts[p], cnt = starts[p + 1] - e0;
  if (cnt >= (1u << 18)) { *overflow = 1; return; }
  const unsigned long long key = keys[e0];
  const unsigned long long t = key >> 12;
  const int roff = (int)((key >> 6) & 63), coff = (int)(key & 63);
  unsigned long long tr = (unsigned long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
  while (tr * (tr + 1) / 2 > t) --tr;
  const unsigned long long tc = t - tr * (tr + 1) / 2;
  for (int dr = 0; dr <= (roff + 5 >= 64 ? 1 : 0); ++dr)
    for (int dc = 0; dc <= (coff + 5 >= 64 ? 1 : 0); ++dc) {
      const unsigned long long r2 = tr + dr, c2 = tc + dc;
      if (c2 > r2 || r2 >= nt) continue;
      const unsigned long long t2 = r2 * (r2 + 1) / 2 + c2;
      if (PASS == 0) {
        atomicAdd(&cursor[t2], 1u);
      } else {
        const uint32_t at = atomicAdd(&cursor[t2], 1u);
        refs[at] = make_uint2(e0, (cnt << 14) | ((uint32_t)(roff - 64 * dr + kRefBias) << 7) |
                                      (uint32_t)(coff - 64 * dc + kRefBias));
      }
    }
}
__global__ void k_tile_pattern(uint32_t nt, const uint32_t* __restrict__ tile_ptr, uint8_t* __restrict__ nz) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt * (nt + 1) / 2) return;
  if (tile_ptr[t + 1] == tile_ptr[t]) return;
  uint32_t tr = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((unsigned long long)(tr + 1) * (tr + 2) / 2 <= t) ++tr;
  while ((unsigned long long)tr * (tr + 1) / 2 > t) --tr;
  const uint32_t tc = t - (uint32_t)((unsigned long long)tr * (tr + 1) / 2);
  nz[(size_t)tr * nt + tc] = 1; nz[(size_t)tc * nt + tr] = 1;
}

// ---- per-pose terms ----------------------------------------------------------------------------------
struct PoseEnt { uint32_t a, b, s; };
__global__ void k_pose_jterms(DevGraph g, const uint32_t* __restrict__ off, uint32_t* __restrict__ keys,
                              PoseEnt* __restrict__ ent) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= g.O) return;
  const int m = d_meas_opt(g, s), r = d_ref_opt(g, s);
  uint32_t w = off[s];
  for (uint32_t k = 0; k < 2; ++k) {
    if (m >= 0) { keys[w] = 2u * (uint32_t)m; ent[w] = {s * g.R + k, s * g.R + k, 2 * s + k}; ++w; }
    if (r >= 0) { keys[w] = 2u * (uint32_t)r; ent[w] = {s * g.R + 2 + k, s * g.R + 2 + k, 2 * s + k}; ++w; }
  }
}
__global__ void k_pose_schur_terms(DevGraph g, const uint32_t* __restrict__ linc_ptr,
                                   const uint32_t* __restrict__ inc_pose, const uint32_t* __restrict__ inc_wrow,
                                   const uint32_t* __restrict__ off, uint32_t base, uint32_t zero_scalar,
                                   uint32_t* __restrict__ keys, PoseEnt* __restrict__ ent) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= g.L) return;
  const uint32_t q0 = linc_ptr[l], q1 = linc_ptr[l + 1];
  uint32_t w = base + off[l];
  const uint32_t LM = (uint32_t)g.LM;
  for (uint32_t x = q0; x < q1; ++x) {
    const uint32_t key = 2u * inc_pose[x] + 1u;
    for (uint32_t k = 0; k < LM; ++k) {
      keys[w] = key; ent[w] = {inc_wrow[x] + LM + k, inc_wrow[x] + k, 2 * g.O + l * LM + k}; ++w;
    }
    for (uint32_t y = x + 1; y < q1; ++y)
      if (inc_pose[x] == inc_pose[y])
        for (uint32_t k = 0; k < LM; ++k) {
          keys[w] = key; ent[w] = {inc_wrow[x] + LM + k, inc_wrow[y] + k, zero_scalar}; ++w;
          keys[w] = key; ent[w] = {inc_wrow[y] + LM + k, inc_wrow[x] + k, zero_scalar}; ++w;
        }
  }
}
// first index whose key is >= 2 p (pose_ptr) / >= 2 p + 1 (pose_mid) in the sorted key array
__global__ void k_pose_bounds(uint32_t Pact, uint32_t n, const uint32_t* __restrict__ keys,
                              uint32_t* __restrict__ pose_ptr, uint32_t* __restrict__ pose_mid) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > Pact) return;
  auto lower = [&](uint32_t want) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  pose_ptr[p] = lower(2u * p);
  if (p < Pact) pose_mid[p] = lower(2u * p + 1u);
}

template <typename T>
int scan_exclusive(Engine* e, DBuf<char>& tmp, const T* in, T* out, size_t n) {
  size_t bytes = 0;
  BAE_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), e->stream));
  BAE_HIP(tmp.alloc(std::max<size_t>(bytes, 16)));
  BAE_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), e->stream));
  return 0;
}
template <typename K, typename V>
int sort_pairs(Engine* e, DBuf<char>& tmp, const K* kin, K* kout, const V* vin, V* vout, size_t n, int bits) {
  size_t bytes = 0;
  BAE_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, e->stream));
  BAE_HIP(tmp.alloc(std::max<size_t>(bytes, 16)));
  BAE_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, e->stream));
  return 0;
}
int bits_for(unsigned long long maxval) {
  int b = 1;
  while ((1ull << b) <= maxval) ++b;
  return b;
}
// temporary of build_lists_device: released on every exit path (early error returns included)
template <typename T>
struct TBuf : DBuf<T> {
  TBuf() = default;
  TBuf(const TBuf&) = delete;
  TBuf& operator=(const TBuf&) = delete;
  ~TBuf() { this->release(); }
};

// 64-bit sum of 32-bit counts (block partials, one 64-bit atomic per block)
__global__ void __launch_bounds__(256) k_sum_counts64(const uint32_t* __restrict__ cnt, size_t n,
                                                       unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[4];
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += cnt[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// Total of a count array whose 32-bit exclusive scan is `off`.  The counts grow quadratically with the track
// length, so the total is summed in 64 bits FIRST: a total of 2^32 or more wraps the scan silently, and then the
// lists would be built truncated — that is an error here, before anything indexes with the wrapped offsets.
int total_of(Engine* e, const uint32_t* off, const uint32_t* cnt, size_t n, uint32_t* out, const char* what) {
  *out = 0;
  (void)off;
  if (n == 0) return 0;
  DBuf<unsigned long long> acc;
  BAE_HIP(acc.alloc(1));
  unsigned long long tot = 0;
  hipError_t err = hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), e->stream);
  if (err == hipSuccess) {
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_sum_counts64, dim3(grid), dim3(256), 0, e->stream, cnt, n, acc.p);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpyAsync(&tot, acc.p, sizeof(tot), hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  acc.release();
  if (err != hipSuccess) return e->fail(err, "total_of");
  if (tot >= 0xFFFFFFFFull) {
    e->err = std::string(what) + " exceeds 2^32 entries";
    return -1;
  }
  *out = (uint32_t)tot;
  return 0;
}

}  // namespace

#define GRID(n) dim3((unsigned)(((size_t)(n) + 255) / 256)), dim3(256), 0, e->stream

// Fills the Engine's list buffers and the scalar fields of e->st from e->prob.
int build_lists_device(Engine* e, const std::function<void(const char*)>& stage) {
  const Problem& pb = e->prob;
  Structure& st = e->st;
  const int LM = e->lm_dim, D = e->pose_dim;
  st = Structure();
  st.P = pb.num_poses; st.L = pb.num_lms; st.O = pb.num_proj; st.C = pb.num_cams;
  if (st.O > 0 && st.C == 0) return e->fail_msg("projection residuals without a camera");
  if (st.O > 0 && LM == 0) return e->fail_msg("projection residuals need LmSize 1 or 3");
  st.pose_opt.assign(st.P, -1);
  for (uint32_t p = 0; p < st.P; ++p)
    if (pb.pose_active[p]) st.pose_opt[p] = (int32_t)st.Pact++;
  st.lm_opt.assign(st.L, -1);
  for (uint32_t l = 0; l < st.L; ++l)
    if (pb.lm_active[l] && LM > 0) st.lm_opt[l] = (int32_t)st.Lact++;
  st.np = st.Pact * D; st.K = (uint32_t)e->calib_dim;
  st.n = st.np + st.K;
  st.ld = ((st.n + 63) / 64) * 64;
  if (st.ld == 0) st.ld = 64;
  for (uint32_t a = 0; a < st.O; ++a)
    if (pb.proj_pose[a] >= st.P || pb.proj_lm[a] >= st.L || pb.proj_cam[a] >= st.C)
      return e->fail_msg("projection residual references an unknown pose/landmark/camera");
  for (uint32_t l = 0; l < st.L; ++l)
    if (pb.lm_ref_pose[l] >= st.P || (st.C > 0 && pb.lm_ref_cam[l] >= st.C))
      return e->fail_msg("landmark references an unknown pose/camera");
  const uint32_t R = (uint32_t)rows_per_obs(LM), WO = (uint32_t)w_row_offset(LM), O = st.O, L = st.L;
  st.R = R;
  if ((uint64_t)O * R + 2ull * L + 1 >= 0xFFFFFFFFull) return e->fail_msg("more than 2^32 factor rows");
  st.lrow_base = O * R;
  st.n_rows = st.lrow_base + (LM == 1 ? 2 * L : 0) + 1;
  st.zero_scalar = 2 * O + L * (uint32_t)std::max(LM, 1);
  st.n_scalars = st.zero_scalar + 1;
  const uint32_t nt = st.ld / 64;
  const uint64_t tiles_lower = (uint64_t)nt * (nt + 1) / 2;
  const size_t O1 = std::max<size_t>(O, 1), L1 = std::max<size_t>(L, 1);

  // ---- raw arrays to the device -----------------------------------------------------------------------
  TBuf<double> r_z, r_w;
  TBuf<uint32_t> r_pose, r_lm, r_cam, perm_in, key_tmp;
  TBuf<uint8_t> r_cond;
  TBuf<char> tmp;
  int rc = 0;
#define UPV(buf, vec) { BAE_HIP(buf.alloc(std::max<size_t>((vec).size(), 1))); \
    if (!(vec).empty()) BAE_HIP(hipMemcpyAsync(buf.p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, e->stream)); }
  UPV(r_z, pb.proj_z); UPV(r_w, pb.proj_w); UPV(r_pose, pb.proj_pose); UPV(r_lm, pb.proj_lm); UPV(r_cam, pb.proj_cam);
  UPV(e->pose_opt, st.pose_opt); UPV(e->lm_opt, st.lm_opt);
  UPV(e->lm_ref_pose, pb.lm_ref_pose); UPV(e->lm_ref_cam, pb.lm_ref_cam);
  std::vector<uint8_t> is_cond(O1, 0);
  for (uint32_t id : pb.proj_cond) if (id < O) is_cond[id] = 1;
  UPV(r_cond, is_cond);
#undef UPV
  BAE_HIP(hipStreamSynchronize(e->stream));  // is_cond is a local
  if (stage) stage("raw arrays to the device");

  // ---- observations sorted by landmark (stable), CSR -------------------------------------------------
  BAE_HIP(e->obs_rid.alloc(O1)); BAE_HIP(perm_in.alloc(O1)); BAE_HIP(key_tmp.alloc(O1));
  BAE_HIP(e->lm_ptr.alloc((size_t)L + 1));
  BAE_HIP(e->obs_z.alloc(2 * O1)); BAE_HIP(e->obs_w0.alloc(O1)); BAE_HIP(e->obs_w.alloc(O1));
  BAE_HIP(e->obs_pose.alloc(O1)); BAE_HIP(e->obs_cam.alloc(O1)); BAE_HIP(e->obs_lm.alloc(O1));
  BAE_HIP(e->obs_cond.alloc(O1));
  TBuf<uint32_t> lm_cnt;
  BAE_HIP(lm_cnt.alloc((size_t)L + 1));
  BAE_HIP(hipMemsetAsync(lm_cnt.p, 0, ((size_t)L + 1) * 4, e->stream));
  if (O) {
    hipLaunchKernelGGL(k_iota, GRID(O), O, perm_in.p);
    hipLaunchKernelGGL(k_count_lm, GRID(O), O, (const uint32_t*)r_lm.p, lm_cnt.p);
    if ((rc = sort_pairs(e, tmp, (const uint32_t*)r_lm.p, key_tmp.p, (const uint32_t*)perm_in.p, e->obs_rid.p, O,
                         bits_for(L ? L - 1 : 0)))) return rc;
    hipLaunchKernelGGL(k_gather_obs, GRID(O), O, (const uint32_t*)e->obs_rid.p, (const double*)r_z.p,
                       (const double*)r_w.p, (const uint32_t*)r_pose.p, (const uint32_t*)r_lm.p,
                       (const uint32_t*)r_cam.p, (const uint8_t*)r_cond.p, e->obs_z.p, e->obs_w0.p, e->obs_w.p,
                       e->obs_pose.p, e->obs_lm.p, e->obs_cam.p, e->obs_cond.p);
  }
  if ((rc = scan_exclusive(e, tmp, (const uint32_t*)lm_cnt.p, e->lm_ptr.p, (size_t)L + 1))) return rc;
  BAE_HIP(hipGetLastError());
  st.lm_ptr.assign((size_t)L + 1, 0);
  st.obs_perm.assign(O, 0);
  BAE_HIP(hipMemcpyAsync(st.lm_ptr.data(), e->lm_ptr.p, ((size_t)L + 1) * 4, hipMemcpyDeviceToHost, e->stream));
  if (O) BAE_HIP(hipMemcpyAsync(st.obs_perm.data(), e->obs_rid.p, (size_t)O * 4, hipMemcpyDeviceToHost, e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  r_z.release(); r_w.release(); r_pose.release(); r_lm.release(); r_cam.release(); r_cond.release();
  perm_in.release(); key_tmp.release(); lm_cnt.release();
  if (stage) stage("obs sort by landmark");

  // ---- linearisation waves (greedy packing of whole landmarks: sequential, 1 M steps on the host) ----
  {
    std::vector<U2> small, big;
    uint32_t cur = 0, start = 0;
    for (uint32_t l = 0; l < L; ++l) {
      const uint32_t k = st.lm_ptr[l + 1] - st.lm_ptr[l];
      if (k == 0) continue;
      if (k > 64) {
        if (cur) { small.push_back({start, st.lm_ptr[l]}); cur = 0; }
        big.push_back({st.lm_ptr[l], st.lm_ptr[l + 1]});
        continue;
      }
      if (cur + k > 64) { small.push_back({start, st.lm_ptr[l]}); cur = 0; }
      if (cur == 0) start = st.lm_ptr[l];
      cur += k;
    }
    if (cur) small.push_back({start, st.lm_ptr[L]});
    st.n_big_chunks = (uint32_t)big.size();
    small.insert(small.end(), big.begin(), big.end());
    st.n_chunks = (uint32_t)small.size();
    BAE_HIP(e->wave_rng.alloc(std::max<size_t>(st.n_chunks, 1)));
    if (st.n_chunks) BAE_HIP(hipMemcpy(e->wave_rng.p, small.data(), (size_t)st.n_chunks * sizeof(U2), hipMemcpyHostToDevice));
  }

  DevGraph g = {LM, D, O, L, R, WO, st.lrow_base, e->obs_pose.p, e->obs_lm.p, e->lm_ptr.p,
                e->lm_ref_pose.p, e->pose_opt.p, e->lm_opt.p};
  // ---- incidences -----------------------------------------------------------------------------------------
  TBuf<uint32_t> linc_cnt, linc_ptr, inc_pose, inc_wrow;
  BAE_HIP(linc_cnt.alloc((size_t)L + 1)); BAE_HIP(linc_ptr.alloc((size_t)L + 1));
  BAE_HIP(hipMemsetAsync(linc_cnt.p, 0, ((size_t)L + 1) * 4, e->stream));
  if (L) hipLaunchKernelGGL(k_inc_count, GRID(L), g, linc_cnt.p);
  if ((rc = scan_exclusive(e, tmp, (const uint32_t*)linc_cnt.p, linc_ptr.p, (size_t)L + 1))) return rc;
  BAE_HIP(hipMemcpyAsync(&st.n_inc, linc_ptr.p + L, 4, hipMemcpyDeviceToHost, e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  BAE_HIP(inc_pose.alloc(std::max<size_t>(st.n_inc, 1))); BAE_HIP(inc_wrow.alloc(std::max<size_t>(st.n_inc, 1)));
  if (L) hipLaunchKernelGGL(k_inc_fill, GRID(L), g, (const uint32_t*)linc_ptr.p, inc_pose.p, inc_wrow.p);

  // ---- rank-1 terms of the off-diagonal blocks -----------------------------------------------------------
  TBuf<uint32_t> lrec_cnt, lrec_off, orec_cnt, orec_off, schur_cnt, schur_off, jt_cnt, jt_off;
  BAE_HIP(lrec_cnt.alloc(L1)); BAE_HIP(lrec_off.alloc(L1)); BAE_HIP(schur_cnt.alloc(L1)); BAE_HIP(schur_off.alloc(L1));
  BAE_HIP(orec_cnt.alloc(O1)); BAE_HIP(orec_off.alloc(O1)); BAE_HIP(jt_cnt.alloc(O1)); BAE_HIP(jt_off.alloc(O1));
  uint32_t n_lm_recs = 0, n_obs_recs = 0, n_schur = 0, n_jterms = 0;
  if (L) {
    hipLaunchKernelGGL(k_lm_counts, GRID(L), g, (const uint32_t*)linc_ptr.p, (const uint32_t*)inc_pose.p, lrec_cnt.p,
                       schur_cnt.p);
    if ((rc = scan_exclusive(e, tmp, (const uint32_t*)lrec_cnt.p, lrec_off.p, L))) return rc;
    if ((rc = scan_exclusive(e, tmp, (const uint32_t*)schur_cnt.p, schur_off.p, L))) return rc;
    if ((rc = total_of(e, lrec_off.p, lrec_cnt.p, L, &n_lm_recs, "gather list (landmark records)"))) return rc;
    if ((rc = total_of(e, schur_off.p, schur_cnt.p, L, &n_schur, "per-pose term list (Schur terms)"))) return rc;
  }
  if (O) {
    hipLaunchKernelGGL(k_obs_counts, GRID(O), g, orec_cnt.p, jt_cnt.p);
    if ((rc = scan_exclusive(e, tmp, (const uint32_t*)orec_cnt.p, orec_off.p, O))) return rc;
    if ((rc = scan_exclusive(e, tmp, (const uint32_t*)jt_cnt.p, jt_off.p, O))) return rc;
    if ((rc = total_of(e, orec_off.p, orec_cnt.p, O, &n_obs_recs, "gather list (observation records)"))) return rc;
    if ((rc = total_of(e, jt_off.p, jt_cnt.p, O, &n_jterms, "per-pose term list (J terms)"))) return rc;
  }
  const uint64_t n_recs64 = (uint64_t)n_lm_recs + n_obs_recs;
  if (n_recs64 >= 0xFFFFFFFFull) return e->fail_msg("gather list exceeds 2^32 entries");
  const uint32_t n_recs = (uint32_t)n_recs64;
  st.n_pair_entries = n_recs;
  BAE_HIP(e->pair_ent.alloc(std::max<size_t>(n_recs, 1)));
  BAE_HIP(e->tile_ptr.alloc(tiles_lower + 1));
  BAE_HIP(hipMemsetAsync(e->tile_ptr.p, 0, (tiles_lower + 1) * 4, e->stream));
  st.tile_nz.assign((size_t)nt * nt, 0);
  if (n_recs) {
    TBuf<unsigned long long> k0, k1, v0;
    BAE_HIP(k0.alloc(n_recs)); BAE_HIP(k1.alloc(n_recs)); BAE_HIP(v0.alloc(n_recs));
    if (L) hipLaunchKernelGGL(k_lm_records, GRID(L), g, (const uint32_t*)linc_ptr.p, (const uint32_t*)inc_pose.p,
                              (const uint32_t*)inc_wrow.p, (const uint32_t*)lrec_off.p, k0.p, v0.p);
    if (O) hipLaunchKernelGGL(k_obs_records, GRID(O), g, (const uint32_t*)orec_off.p, n_lm_recs, k0.p, v0.p);
    BAE_HIP(hipGetLastError());
    if (stage) { BAE_HIP(hipStreamSynchronize(e->stream)); stage("pair terms generated"); }
    static_assert(sizeof(uint2) == sizeof(unsigned long long), "a term is two row indices");
    if ((rc = sort_pairs(e, tmp, (const unsigned long long*)k0.p, k1.p, (const unsigned long long*)v0.p,
                         reinterpret_cast<unsigned long long*>(e->pair_ent.p), n_recs,
                         bits_for((tiles_lower << 12) | 4095ull)))) return rc;
    k0.release(); v0.release();
    if (stage) { BAE_HIP(hipStreamSynchronize(e->stream)); stage("pair terms sorted"); }
    // blocks, tile references
    TBuf<uint32_t> flag, fidx, starts, cursor;
    TBuf<int> ovf;
    BAE_HIP(flag.alloc(n_recs)); BAE_HIP(fidx.alloc(n_recs)); BAE_HIP(ovf.alloc(1));
    BAE_HIP(hipMemsetAsync(ovf.p, 0, sizeof(int), e->stream));
    hipLaunchKernelGGL(k_block_flags, GRID(n_recs), n_recs, (const unsigned long long*)k1.p, flag.p);
    if ((rc = scan_exclusive(e, tmp, (const uint32_t*)flag.p, fidx.p, n_recs))) return rc;
    if ((rc = total_of(e, fidx.p, flag.p, n_recs, &st.n_pairs, "pose-pair block list"))) return rc;
    BAE_HIP(starts.alloc((size_t)st.n_pairs + 1));
    hipLaunchKernelGGL(k_block_starts, GRID(n_recs), n_recs, (const uint32_t*)flag.p, (const uint32_t*)fidx.p, starts.p,
                       st.n_pairs);
    BAE_HIP(cursor.alloc(tiles_lower + 1));
    BAE_HIP(hipMemsetAsync(cursor.p, 0, (tiles_lower + 1) * 4, e->stream));
    hipLaunchKernelGGL(k_tile_refs<0>, GRID(st.n_pairs), st.n_pairs, nt, (const uint32_t*)starts.p,
                       (const unsigned long long*)k1.p, cursor.p, (uint2*)nullptr, ovf.p);
    if ((rc = scan_exclusive(e, tmp, (const uint32_t*)cursor.p, e->tile_ptr.p, tiles_lower + 1))) return rc;
    uint32_t n_refs = 0;
    BAE_HIP(hipMemcpyAsync(&n_refs, e->tile_ptr.p + tiles_lower, 4, hipMemcpyDeviceToHost, e->stream));
    int overflow = 0;
    BAE_HIP(hipMemcpyAsync(&overflow, ovf.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    if (overflow) return e->fail_msg("a pose pair shares more than 2^18 terms");
    st.n_tile_refs = n_refs;
    BAE_HIP(e->tile_ref.alloc(std::max<size_t>(n_refs, 1)));
    BAE_HIP(hipMemcpyAsync(cursor.p, e->tile_ptr.p, (tiles_lower + 1) * 4, hipMemcpyDeviceToDevice, e->stream));
    hipLaunchKernelGGL(k_tile_refs<1>, GRID(st.n_pairs), st.n_pairs, nt, (const uint32_t*)starts.p,
                       (const unsigned long long*)k1.p, cursor.p, e->tile_ref.p, ovf.p);
    TBuf<uint8_t> nz;
    BAE_HIP(nz.alloc((size_t)nt * nt));
    BAE_HIP(hipMemsetAsync(nz.p, 0, (size_t)nt * nt, e->stream));
    hipLaunchKernelGGL(k_tile_pattern, GRID(tiles_lower), nt, (const uint32_t*)e->tile_ptr.p, nz.p);
    BAE_HIP(hipGetLastError());
    BAE_HIP(hipMemcpyAsync(st.tile_nz.data(), nz.p, (size_t)nt * nt, hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    flag.release(); fidx.release(); starts.release(); cursor.release(); ovf.release(); nz.release(); k1.release();
  } else {
    BAE_HIP(e->tile_ref.alloc(1));
  }
  if (stage) stage("tile references");

  // ---- per-pose terms: J terms (segment 0) then Schur terms (segment 1), stable sort by pose -----------
  const uint64_t n_pe64 = (uint64_t)n_jterms + n_schur;
  if (n_pe64 >= 0xFFFFFFFFull) return e->fail_msg("per-pose term list exceeds 2^32 entries");
  const uint32_t n_pe = (uint32_t)n_pe64;
  st.n_pose_entries = n_pe;
  BAE_HIP(e->pose_ptr.alloc((size_t)st.Pact + 1)); BAE_HIP(e->pose_mid.alloc(std::max<size_t>(st.Pact, 1)));
  BAE_HIP(e->pose_ent.alloc(std::max<size_t>(3 * (size_t)n_pe, 1)));
  static_assert(sizeof(PoseEnt) == sizeof(U3), "per-pose term = three words");
  {
    TBuf<uint32_t> pk0, pk1;
    TBuf<PoseEnt> pe0;
    BAE_HIP(pk0.alloc(std::max<size_t>(n_pe, 1))); BAE_HIP(pk1.alloc(std::max<size_t>(n_pe, 1)));
    BAE_HIP(pe0.alloc(std::max<size_t>(n_pe, 1)));
    if (O) hipLaunchKernelGGL(k_pose_jterms, GRID(O), g, (const uint32_t*)jt_off.p, pk0.p, pe0.p);
    if (L) hipLaunchKernelGGL(k_pose_schur_terms, GRID(L), g, (const uint32_t*)linc_ptr.p, (const uint32_t*)inc_pose.p,
                              (const uint32_t*)inc_wrow.p, (const uint32_t*)schur_off.p, n_jterms, st.zero_scalar,
                              pk0.p, pe0.p);
    BAE_HIP(hipGetLastError());
    if (n_pe)
      if ((rc = sort_pairs(e, tmp, (const uint32_t*)pk0.p, pk1.p, (const PoseEnt*)pe0.p,
                           reinterpret_cast<PoseEnt*>(e->pose_ent.p), n_pe, bits_for(2ull * st.Pact + 1)))) return rc;
    hipLaunchKernelGGL(k_pose_bounds, GRID(st.Pact + 1), st.Pact, n_pe, (const uint32_t*)pk1.p, e->pose_ptr.p,
                       e->pose_mid.p);
    BAE_HIP(hipGetLastError());
    BAE_HIP(hipStreamSynchronize(e->stream));
    pk0.release(); pk1.release(); pe0.release();
  }
  linc_cnt.release(); linc_ptr.release(); inc_pose.release(); inc_wrow.release();
  lrec_cnt.release(); lrec_off.release(); orec_cnt.release(); orec_off.release();
  schur_cnt.release(); schur_off.release(); jt_cnt.release(); jt_off.release();
  tmp.release();
  // the diagonal D x D blocks (and the padding identity) are always present in the tile pattern
  for (uint32_t p = 0; p < st.Pact; ++p) {
    const uint32_t r0 = p * D / 64, r1 = (p * D + D - 1) / 64;
    for (uint32_t r = r0; r <= r1; ++r)
      for (uint32_t c = r0; c <= r1; ++c) { st.tile_nz[(size_t)r * nt + c] = 1; st.tile_nz[(size_t)c * nt + r] = 1; }
  }
  for (uint32_t t = 0; t < nt; ++t) st.tile_nz[(size_t)t * nt + t] = 1;
  if (stage) stage("pose terms");
  return 0;
}

}  // namespace bae
