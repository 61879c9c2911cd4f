// Host-side static structure of one problem graph (ba_hip_finalize): everything that depends only
// on WHICH poses / landmarks / residuals exist — not on their values — and is therefore built once
// per graph and reused by every Gauss-Newton iteration and every Solve() call until the graph
// changes.  Plain C++17, no HIP: the same code runs in the CPU test harness
// (ba_amd/csrc/hostcheck.cpp, tests/test_structure_lists.py) that checks the lists against a dense
// brute-force Schur complement.
//
// Replaces the bookkeeping of the reference's BuildProblem: sorted block insertion into j_pr_ /
// jt_pr / j_l_ (BundleAdjuster.cpp:1552-1802) and the symbolic side of its block-sparse products
// (SparseBlockMatrixOps.h:182-254).
//
// ---- factor rows (48 bytes = 6 doubles), OBSERVATION-MAJOR -------------------------------------
// Observations are sorted by landmark (CSR lm_ptr).  The linearisation kernel (k_linearize, one
// thread per observation) writes for observation a the R = rows_per_obs(LM) consecutive rows
//     LM == 1:  a*6 + 0,1  sqrt(w) * dz_dx_meas (u row, v row)        "J_m"
//               a*6 + 2,3  sqrt(w) * dz_dx_ref                        "J_r"
//               a*6 + 4    W_m   = w J_m^T J_l        (6 x 1)
//               a*6 + 5    NWV_m = -W_m V^-1
//     LM == 3:  a*8 + 0,1  J_m ;  a*8 + 2..4  columns of W_m (6 x 3) ;  a*8 + 5..7  columns of -W_m V^-1
// and for landmark l (LM == 1 only) the two rows  lrow_base + 2 l : W_r = sum_a w J_r^T J_l ,
// lrow_base + 2 l + 1 : -W_r V^-1   (the reference pose's incidence).  A wave writes one contiguous
// span.  Duplicate observations of a landmark from the same pose simply stay separate rows: the
// lists below enumerate every pair of rows of a landmark, which is algebraically the same sum.
//
// ---- lists ----------------------------------------------------------------------------------------
//   wave_rng   observation ranges of the linearisation waves: whole landmarks, at most 64 observations
//              (a landmark with more than 64 gets a range of its own at the end of the list: two-pass
//              variant of the kernel)
//   pair_ent   (rowA, rowB): one rank-1 term rowA (x) rowB of an off-diagonal pose-pair block of S,
//              sorted by (home tile, position in tile) so that a block's terms are contiguous
//   tile_ref   per 64x64 tile of the lower storage of S: the blocks that overlap it —
//              (first term, count << 14 | (row offset + 8) << 7 | (column offset + 8)); a block that
//              straddles a tile boundary is referenced by every tile it touches
//   pose_ent   per active pose: (rowA, rowB, scalar index) terms of its diagonal block and of its
//              right-hand side, J terms first (rhs_p), Schur terms after `mid`
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace bae {

struct Problem {  // host copies, reference ids
  uint32_t num_cams = 0, num_poses = 0, num_lms = 0, num_proj = 0;
  uint32_t num_unary = 0, num_binary = 0, num_imu = 0;
  std::vector<double> cam_params, cam_tvs;                 // [C][4], [C][7]
  std::vector<int32_t> cam_model;                          // [C] 0 LinearCamera, 1 FovCamera (empty: all 0)
  std::vector<double> cam_w;                               // [C] FOV distortion parameter w (model 1)
  std::vector<double> pose_cam_params;                     // [P][4] or empty (use_per_pose_cam_params)
  std::vector<double> imu_noise;                           // r(6) | r_b(6) or empty (from the option sigmas)
  std::vector<double> pose_state;                          // [P][16]
  std::vector<uint8_t> pose_active;
  std::vector<double> lm_xw;                               // [L][4]
  std::vector<uint32_t> lm_ref_pose, lm_ref_cam;
  std::vector<double> lm_zref;                             // [L][2] reference pixels (intrinsics calibration only)
  std::vector<uint8_t> lm_active;
  std::vector<double> proj_z, proj_w;                      // [O][2], [O]
  std::vector<uint32_t> proj_pose, proj_lm, proj_cam;
  std::vector<uint32_t> proj_cond;                         // ids of the conditioning residuals (summary only)
  // pose-pose residuals
  std::vector<uint32_t> un_pose; std::vector<double> un_t, un_cov_inv; std::vector<uint8_t> un_rot;
  std::vector<uint32_t> bin_p1, bin_p2; std::vector<double> bin_t, bin_cov_inv, bin_cov_inv_sqrt, bin_w;
  std::vector<uint8_t> bin_rot;
  std::vector<uint32_t> imu_p1, imu_p2, imu_ptr; std::vector<double> imu_meas, imu_w;
  double gravity[3] = {0, 0, -9.8007};
};

struct U2 { uint32_t x, y; };        // same layout as HIP's uint2
struct U3 { uint32_t a, b, s; };     // 12 bytes

inline constexpr int rows_per_obs(int LM) { return LM == 1 ? 6 : (LM == 3 ? 8 : 0); }
inline constexpr int w_row_offset(int LM) { return LM == 1 ? 4 : 2; }  // first W row inside an observation's rows
static const int kRefBias = 8;  // bias of the (possibly negative) block offsets stored in a tile reference

struct Lists {
  uint32_t P = 0, Pact = 0, L = 0, Lact = 0, O = 0, C = 0;
  uint32_t n = 0, ld = 0;       // unknowns of the reduced system (np + K) and its padded leading dimension
  uint32_t np = 0, K = 0;       // pose unknowns (Pact * D); calibration unknowns behind them (0, or 6 = T_vs)
  uint32_t R = 0;               // rows per observation
  uint32_t lrow_base = 0;       // first landmark row (LM == 1)
  uint32_t n_rows = 0;          // factor rows incl. the trailing all-zero row
  uint32_t zero_scalar = 0;     // index of a scalar that is always 0
  uint32_t n_scalars = 0;
  uint32_t n_chunks = 0;        // linearisation waves; the last n_big_chunks hold one landmark with > 64 observations
  uint32_t n_big_chunks = 0;
  uint32_t n_inc = 0;           // (active pose, active landmark) incidences
  uint32_t n_pairs = 0;         // off-diagonal pose-pair blocks with at least one term
  uint64_t n_pair_entries = 0;
  uint64_t n_tile_refs = 0;
  uint64_t n_pose_entries = 0;
  std::vector<int32_t> pose_opt, lm_opt;
  std::vector<uint32_t> obs_perm;           // sorted position -> residual id
  std::vector<uint32_t> lm_ptr;             // [L+1]
  std::vector<double> obs_z, obs_w0;
  std::vector<uint32_t> obs_pose, obs_cam, obs_lm, obs_rid;
  std::vector<U2> wave_rng;                 // [n_chunks] observation range [x, y) of every linearisation wave
  std::unique_ptr<U2[]> pair_ent;           // [n_pair_entries]
  std::vector<uint32_t> tile_ptr;           // [tiles_lower+1]
  std::vector<U2> tile_ref;
  std::vector<uint32_t> pose_ptr;           // [Pact+1]
  std::vector<uint32_t> pose_mid;           // [Pact]  first Schur term of the pose
  std::vector<U3> pose_ent;
  std::vector<uint8_t> tile_nz;             // nt x nt, symmetric: tiles the projection part of S touches (+ diagonal)
};

inline unsigned structure_threads() {
  if (const char* v = getenv("BA_HIP_HOST_THREADS")) return (unsigned)std::max(1, atoi(v));
  const unsigned hc = std::thread::hardware_concurrency();
  return std::min(16u, std::max(1u, hc));
}

// Returns false and sets `err` on an inconsistent graph.
inline bool build_lists(const Problem& pb, int LM, int D, Lists& st, std::string& err,
                        const std::function<void(const char*)>& stage = nullptr, int K = 0) {
  auto mark_stage = [&](const char* s) { if (stage) stage(s); };
  st = Lists();
  st.P = pb.num_poses; st.L = pb.num_lms; st.O = pb.num_proj; st.C = pb.num_cams;
  if (st.O > 0 && st.C == 0) { err = "projection residuals without a camera"; return false; }
  if (st.O > 0 && LM == 0) { err = "projection residuals need LmSize 1 or 3"; return false; }
  // opt ids: running count of active items in id order (BundleAdjuster.h:309-316,353-360)
  st.pose_opt.assign(st.P, -1);
  for (uint32_t p = 0; p < st.P; ++p)
    if (pb.pose_active[p]) st.pose_opt[p] = (int32_t)st.Pact++;
  st.lm_opt.assign(st.L, -1);
  for (uint32_t l = 0; l < st.L; ++l)
    if (pb.lm_active[l] && LM > 0) st.lm_opt[l] = (int32_t)st.Lact++;
  st.np = st.Pact * D; st.K = (uint32_t)K;
  st.n = st.np + st.K;
  st.ld = ((st.n + 63) / 64) * 64;
  if (st.ld == 0) st.ld = 64;
  for (uint32_t a = 0; a < st.O; ++a)
    if (pb.proj_pose[a] >= st.P || pb.proj_lm[a] >= st.L || pb.proj_cam[a] >= st.C) {
      err = "projection residual references an unknown pose/landmark/camera";
      return false;
    }
  for (uint32_t l = 0; l < st.L; ++l)
    if (pb.lm_ref_pose[l] >= st.P || (st.C > 0 && pb.lm_ref_cam[l] >= st.C)) {
      err = "landmark references an unknown pose/camera";
      return false;
    }
  const uint32_t R = (uint32_t)rows_per_obs(LM), WO = (uint32_t)w_row_offset(LM);
  st.R = R;
  if ((uint64_t)st.O * R + 2ull * st.L + 1 >= 0xFFFFFFFFull) { err = "more than 2^32 factor rows"; return false; }
  st.lrow_base = st.O * R;
  st.n_rows = st.lrow_base + (LM == 1 ? 2 * st.L : 0) + 1;
  st.zero_scalar = 2 * st.O + st.L * (uint32_t)std::max(LM, 1);
  st.n_scalars = st.zero_scalar + 1;

  // ---- observations sorted by landmark (stable in residual id) --------------------
  st.lm_ptr.assign((size_t)st.L + 1, 0);
  for (uint32_t a = 0; a < st.O; ++a) st.lm_ptr[pb.proj_lm[a] + 1]++;
  for (uint32_t l = 0; l < st.L; ++l) st.lm_ptr[l + 1] += st.lm_ptr[l];
  st.obs_perm.assign(st.O, 0);
  {
    std::vector<uint32_t> cur(st.lm_ptr.begin(), st.lm_ptr.end() - 1);
    for (uint32_t a = 0; a < st.O; ++a) st.obs_perm[cur[pb.proj_lm[a]]++] = a;
  }
  st.obs_z.resize(2 * (size_t)st.O); st.obs_w0.resize(st.O);
  st.obs_pose.resize(st.O); st.obs_cam.resize(st.O); st.obs_lm.resize(st.O); st.obs_rid.resize(st.O);
  for (uint32_t s = 0; s < st.O; ++s) {
    const uint32_t a = st.obs_perm[s];
    st.obs_z[2 * (size_t)s] = pb.proj_z[2 * (size_t)a];
    st.obs_z[2 * (size_t)s + 1] = pb.proj_z[2 * (size_t)a + 1];
    st.obs_w0[s] = pb.proj_w[a];
    st.obs_pose[s] = pb.proj_pose[a]; st.obs_cam[s] = pb.proj_cam[a];
    st.obs_lm[s] = pb.proj_lm[a]; st.obs_rid[s] = a;
  }
  mark_stage("obs sort by landmark");

  // ---- linearisation waves: whole landmarks, at most 64 observations; a landmark with more than 64
  // gets a range of its own, listed after the small ones (separate launch: two-pass kernel) ----------
  {
    std::vector<U2> big;
    uint32_t cur = 0, start = 0;  // observations in the open range, its first observation
    for (uint32_t l = 0; l < st.L; ++l) {
      const uint32_t k = st.lm_ptr[l + 1] - st.lm_ptr[l];
      if (k == 0) continue;
      if (k > 64) {
        if (cur) { st.wave_rng.push_back({start, st.lm_ptr[l]}); cur = 0; }
        big.push_back({st.lm_ptr[l], st.lm_ptr[l + 1]});
        continue;
      }
      if (cur + k > 64) { st.wave_rng.push_back({start, st.lm_ptr[l]}); cur = 0; }
      if (cur == 0) start = st.lm_ptr[l];
      cur += k;
    }
    if (cur) st.wave_rng.push_back({start, st.lm_ptr[st.L]});
    st.n_big_chunks = (uint32_t)big.size();
    st.wave_rng.insert(st.wave_rng.end(), big.begin(), big.end());
  }
  st.n_chunks = (uint32_t)st.wave_rng.size();

  // ---- which sides of an observation carry blocks -------------------------------------------------
  // A residual carries pose Jacobian blocks iff it is "listed" (it passed the diff_poses test of
  // AddProjectionResidual, BundleAdjuster.h:489-497) and the pose is active (blocks are only
  // inserted for active poses, BundleAdjuster.cpp:1613-1643).
  auto listed = [&](uint32_t s) { return LM != 1 || st.obs_pose[s] != pb.lm_ref_pose[st.obs_lm[s]]; };
  auto meas_opt = [&](uint32_t s) { return listed(s) ? st.pose_opt[st.obs_pose[s]] : -1; };
  auto ref_opt = [&](uint32_t s) {
    return (LM == 1 && listed(s)) ? st.pose_opt[pb.lm_ref_pose[st.obs_lm[s]]] : -1;
  };
  const unsigned T = structure_threads();
  auto parallel_for = [&](size_t count, const std::function<void(unsigned, size_t, size_t)>& fn) {
    if (T <= 1 || count < (1u << 16)) { fn(0, 0, count); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t, count * t / T, count * (t + 1) / T);
    for (auto& x : th) x.join();
  };

  // Incidences of landmark l (only if the landmark is active): one per observation whose measuring
  // pose carries a block, plus the reference pose's (LM == 1, reference pose active, any listed
  // observation).  inc = (pose opt id, first W row).
  struct Inc { uint32_t pose, wrow; };
  auto incidences = [&](uint32_t l, Inc* out) -> uint32_t {  // out must hold k + 1 items
    if (st.lm_opt[l] < 0) return 0;
    uint32_t m = 0;
    bool any_listed = false;
    for (uint32_t s = st.lm_ptr[l]; s < st.lm_ptr[l + 1]; ++s) {
      if (!listed(s)) continue;
      any_listed = true;
      const int32_t po = st.pose_opt[st.obs_pose[s]];
      if (po >= 0) out[m++] = {(uint32_t)po, s * R + WO};
    }
    if (LM == 1 && any_listed) {
      const int32_t ro = st.pose_opt[pb.lm_ref_pose[l]];
      if (ro >= 0) out[m++] = {(uint32_t)ro, st.lrow_base + 2 * l};
    }
    return m;
  };
  uint32_t kmax = 0;
  for (uint32_t l = 0; l < st.L; ++l) kmax = std::max(kmax, st.lm_ptr[l + 1] - st.lm_ptr[l]);

  // ---- off-diagonal terms: count, generate, sort by (home tile, offsets) -------------------------
  const uint32_t nt = st.ld / 64;
  auto home_key = [&](uint32_t i, uint32_t j) -> uint64_t {  // i < j (opt ids); block at rows j*D, cols i*D
    const uint32_t r = j * (uint32_t)D, c = i * (uint32_t)D;
    const uint64_t tr = r / 64, tc = c / 64;
    return ((tr * (tr + 1) / 2 + tc) << 12) | ((uint64_t)(r % 64) << 6) | (c % 64);
  };
  std::vector<size_t> lm_off((size_t)st.L + 1, 0), obs_off((size_t)st.O + 1, 0);
  {
    std::vector<Inc> tmp(kmax + 1);
    for (uint32_t l = 0; l < st.L; ++l) {
      const uint32_t m = incidences(l, tmp.data());
      size_t cnt = 0;
      for (uint32_t x = 0; x < m; ++x)
        for (uint32_t y = x + 1; y < m; ++y) cnt += tmp[x].pose != tmp[y].pose ? (size_t)LM : 0;
      lm_off[l + 1] = lm_off[l] + cnt;
      st.n_inc += m;
    }
  }
  parallel_for(st.O, [&](unsigned, size_t s0, size_t s1) {
    for (size_t s = s0; s < s1; ++s) {
      const int32_t m = meas_opt((uint32_t)s), r = ref_opt((uint32_t)s);
      obs_off[s + 1] = (m >= 0 && r >= 0 && m != r) ? 2 : 0;
    }
  });
  for (uint32_t s = 0; s < st.O; ++s) obs_off[s + 1] += obs_off[s];
  const size_t n_lm_recs = lm_off[st.L], n_recs = n_lm_recs + obs_off[st.O];
  if (n_recs >= 0xFFFFFFFFull) { err = "gather list exceeds 2^32 entries"; return false; }
  struct Rec { uint64_t key; uint32_t a, b; };
  std::unique_ptr<Rec[]> recs(new Rec[std::max<size_t>(n_recs, 1)]);
  {
    const unsigned parts = (T > 1 && n_lm_recs >= (1u << 16)) ? T : 1;
    auto lm_part = [&](unsigned cpart) {
      std::vector<Inc> tmp(kmax + 1);
      const size_t lo = n_lm_recs * cpart / parts, hi = n_lm_recs * (cpart + 1) / parts;
      uint32_t l = (uint32_t)(std::upper_bound(lm_off.begin(), lm_off.end(), lo) - lm_off.begin());
      l = l ? l - 1 : 0;
      while (l < st.L && lm_off[l] < lo) ++l;  // first landmark starting at or after lo
      for (; l < st.L && lm_off[l] < hi; ++l) {
        if (lm_off[l + 1] == lm_off[l]) continue;
        const uint32_t m = incidences(l, tmp.data());
        size_t w = lm_off[l];
        for (uint32_t x = 0; x < m; ++x)
          for (uint32_t y = x + 1; y < m; ++y) {
            if (tmp[x].pose == tmp[y].pose) continue;
            const Inc& lo_i = tmp[x].pose < tmp[y].pose ? tmp[x] : tmp[y];  // block row side i (smaller opt id)
            const Inc& hi_i = tmp[x].pose < tmp[y].pose ? tmp[y] : tmp[x];
            const uint64_t key = home_key(lo_i.pose, hi_i.pose);
            for (int k = 0; k < LM; ++k)
              recs[w++] = {key, lo_i.wrow + LM + k, hi_i.wrow + k};  // (-W V^-1)_i W_j^T
          }
      }
    };
    if (parts == 1) lm_part(0);
    else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < parts; ++t) th.emplace_back(lm_part, t);
      for (auto& x : th) x.join();
    }
  }
  parallel_for(st.O, [&](unsigned, size_t s0, size_t s1) {
    for (size_t s = s0; s < s1; ++s) {
      if (obs_off[s + 1] == obs_off[s]) continue;
      const int32_t m = meas_opt((uint32_t)s), r = ref_opt((uint32_t)s);
      const uint32_t jm = (uint32_t)s * R, jr = (uint32_t)s * R + 2;
      size_t w = n_lm_recs + obs_off[s];
      for (uint32_t k = 0; k < 2; ++k) {  // J_i^T J_j over the u and v rows
        if (m < r) recs[w++] = {home_key((uint32_t)m, (uint32_t)r), jm + k, jr + k};
        else recs[w++] = {home_key((uint32_t)r, (uint32_t)m), jr + k, jm + k};
      }
    }
  });
  mark_stage("pair terms generated");
  {
    // stable LSD radix sort on the key bits actually used: the order of the terms of a block —
    // and with it every bit of S — does not depend on the thread count
    const uint64_t tiles_lower = (uint64_t)nt * (nt + 1) / 2;
    int bits = 13;
    while ((1ull << bits) < (tiles_lower << 12)) ++bits;
    std::unique_ptr<Rec[]> t2(new Rec[std::max<size_t>(n_recs, 1)]);
    std::vector<std::vector<size_t>> cnt(T > 1 ? T : 1, std::vector<size_t>(65536));
    auto radix_pass = [&](int shift) {
      parallel_for(n_recs, [&](unsigned t, size_t i0, size_t i1) {
        std::vector<size_t>& c = cnt[t];
        std::fill(c.begin(), c.end(), 0);
        for (size_t i = i0; i < i1; ++i) c[(recs[i].key >> shift) & 0xFFFF]++;
      });
      const bool par = !(T <= 1 || n_recs < (1u << 16));
      const unsigned used = par ? T : 1;
      size_t run = 0;
      for (int d = 0; d < 65536; ++d)
        for (unsigned t = 0; t < used; ++t) { const size_t c = cnt[t][d]; cnt[t][d] = run; run += c; }
      parallel_for(n_recs, [&](unsigned t, size_t i0, size_t i1) {
        std::vector<size_t>& c = cnt[t];
        for (size_t i = i0; i < i1; ++i) t2[c[(recs[i].key >> shift) & 0xFFFF]++] = recs[i];
      });
      recs.swap(t2);
    };
    for (int shift = 0; shift < bits; shift += 16) radix_pass(shift);
  }
  mark_stage("pair terms sorted");
  // ---- cut into blocks; tile references --------------------------------------------------------------
  st.n_pair_entries = n_recs;
  st.pair_ent.reset(new U2[std::max<size_t>(n_recs, 1)]);
  const uint64_t tiles_lower = (uint64_t)nt * (nt + 1) / 2;
  st.tile_ptr.assign(tiles_lower + 1, 0);
  st.tile_nz.assign((size_t)nt * nt, 0);
  {
    // block starts
    std::vector<uint32_t> starts;
    {
      const unsigned parts = (T <= 1 || n_recs < (1u << 16)) ? 1 : T;
      std::vector<size_t> nstart(parts + 1, 0);
      parallel_for(n_recs, [&](unsigned t, size_t i0, size_t i1) {
        size_t c = 0;
        for (size_t i = i0; i < i1; ++i) {
          c += (i == 0 || recs[i].key != recs[i - 1].key);
          st.pair_ent[i] = {recs[i].a, recs[i].b};
        }
        nstart[t + 1] = c;
      });
      for (unsigned t = 0; t < parts; ++t) nstart[t + 1] += nstart[t];
      starts.resize(nstart[parts] + 1);
      parallel_for(n_recs, [&](unsigned t, size_t i0, size_t i1) {
        size_t w = nstart[t];
        for (size_t i = i0; i < i1; ++i)
          if (i == 0 || recs[i].key != recs[i - 1].key) starts[w++] = (uint32_t)i;
      });
      starts[nstart[parts]] = (uint32_t)n_recs;
    }
    st.n_pairs = (uint32_t)starts.size() - 1;
    // every tile a block touches: home (dr, dc) = (0,0) and, when the 6 rows / columns cross a
    // multiple of 64, (1,0) / (0,1) / (1,1); tiles above the diagonal cannot hold an element of a
    // block with i < j and are skipped
    auto for_tiles = [&](uint32_t p, auto&& fn) {
      const uint64_t key = recs[starts[p]].key;
      const uint64_t t = key >> 12;
      const int roff = (int)((key >> 6) & 63), coff = (int)(key & 63);
      // invert t = tr (tr + 1) / 2 + tc
      uint64_t tr = (uint64_t)((std::sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
      while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
      while (tr * (tr + 1) / 2 > t) --tr;
      const uint64_t tc = t - tr * (tr + 1) / 2;
      for (int dr = 0; dr <= (roff + 5 >= 64 ? 1 : 0); ++dr)
        for (int dc = 0; dc <= (coff + 5 >= 64 ? 1 : 0); ++dc) {
          const uint64_t r2 = tr + dr, c2 = tc + dc;
          if (c2 > r2 || r2 >= nt) continue;
          fn(r2 * (r2 + 1) / 2 + c2, roff - 64 * dr, coff - 64 * dc);
        }
    };
    // counts and slots through atomics: the order of the references inside a tile is irrelevant
    // (every block of a tile is formed by exactly one thread of the assembly kernel)
    std::unique_ptr<std::atomic<uint32_t>[]> slot(new std::atomic<uint32_t>[tiles_lower + 1]);
    for (uint64_t t = 0; t <= tiles_lower; ++t) slot[t].store(0, std::memory_order_relaxed);
    parallel_for(st.n_pairs, [&](unsigned, size_t p0, size_t p1) {
      for (size_t p = p0; p < p1; ++p)
        for_tiles((uint32_t)p, [&](uint64_t t, int, int) { slot[t + 1].fetch_add(1, std::memory_order_relaxed); });
    });
    for (uint64_t t = 0; t < tiles_lower; ++t) st.tile_ptr[t + 1] = st.tile_ptr[t] + slot[t + 1].load(std::memory_order_relaxed);
    st.n_tile_refs = st.tile_ptr[tiles_lower];
    st.tile_ref.resize(st.n_tile_refs);
    for (uint64_t t = 0; t < tiles_lower; ++t) slot[t].store(st.tile_ptr[t], std::memory_order_relaxed);
    std::atomic<bool> too_many(false);
    parallel_for(st.n_pairs, [&](unsigned, size_t p0, size_t p1) {
      for (size_t p = p0; p < p1; ++p) {
        const uint32_t e0 = starts[p], cnt = starts[p + 1] - starts[p];
        if (cnt >= (1u << 18)) { too_many.store(true); continue; }
        for_tiles((uint32_t)p, [&](uint64_t t, int ro, int co) {
          const uint32_t at = slot[t].fetch_add(1, std::memory_order_relaxed);
          st.tile_ref[at] = {e0, (cnt << 14) | ((uint32_t)(ro + kRefBias) << 7) | (uint32_t)(co + kRefBias)};
        });
      }
    });
    if (too_many.load()) { err = "a pose pair shares more than 2^18 terms"; return false; }
    // tile pattern of the projection part of S
    {
      uint64_t t = 0;
      for (uint32_t r = 0; r < nt; ++r)
        for (uint32_t c = 0; c <= r; ++c, ++t)
          if (st.tile_ptr[t + 1] > st.tile_ptr[t]) { st.tile_nz[(size_t)r * nt + c] = 1; st.tile_nz[(size_t)c * nt + r] = 1; }
    }
  }
  recs.reset();
  mark_stage("tile references");

  // ---- per-pose terms: diagonal block + right-hand side ----------------------------------------------
  // segment 1 (rhs_p and U_ii):   (J row, J row, sqrt(w) r index) for every (observation, side) of the pose
  // segment 2 (Schur part):       (NWV_x + k, W_x + k, b_l index) for each of its incidences, and for two
  //                               incidences x != y of one landmark on this SAME pose both cross terms
  //                               with the zero scalar
  {
    std::vector<uint32_t> cntA(st.Pact, 0), cntB(st.Pact, 0);
    for (uint32_t s = 0; s < st.O; ++s) {
      const int32_t m = meas_opt(s), r = ref_opt(s);
      if (m >= 0) cntA[m] += 2;
      if (r >= 0) cntA[r] += 2;
    }
    std::vector<Inc> tmp(kmax + 1);
    for (uint32_t l = 0; l < st.L; ++l) {
      const uint32_t m = incidences(l, tmp.data());
      for (uint32_t x = 0; x < m; ++x) {
        cntB[tmp[x].pose] += LM;
        for (uint32_t y = x + 1; y < m; ++y)
          if (tmp[x].pose == tmp[y].pose) cntB[tmp[x].pose] += 2 * LM;
      }
    }
    st.pose_ptr.assign((size_t)st.Pact + 1, 0);
    st.pose_mid.assign(st.Pact, 0);
    for (uint32_t p = 0; p < st.Pact; ++p) {
      st.pose_mid[p] = st.pose_ptr[p] + cntA[p];
      st.pose_ptr[p + 1] = st.pose_mid[p] + cntB[p];
    }
    st.n_pose_entries = st.Pact ? st.pose_ptr[st.Pact] : 0;
    st.pose_ent.resize(st.n_pose_entries);
    std::vector<uint32_t> curA(st.pose_ptr.begin(), st.pose_ptr.end() - 1), curB(st.pose_mid);
    for (uint32_t s = 0; s < st.O; ++s) {
      const int32_t m = meas_opt(s), r = ref_opt(s);
      for (uint32_t k = 0; k < 2; ++k) {
        if (m >= 0) st.pose_ent[curA[m]++] = {s * R + k, s * R + k, 2 * s + k};
        if (r >= 0) st.pose_ent[curA[r]++] = {s * R + 2 + k, s * R + 2 + k, 2 * s + k};
      }
    }
    for (uint32_t l = 0; l < st.L; ++l) {
      const uint32_t m = incidences(l, tmp.data());
      for (uint32_t x = 0; x < m; ++x) {
        for (int k = 0; k < LM; ++k)
          st.pose_ent[curB[tmp[x].pose]++] = {tmp[x].wrow + LM + k, tmp[x].wrow + k, 2 * st.O + l * LM + k};
        for (uint32_t y = x + 1; y < m; ++y)
          if (tmp[x].pose == tmp[y].pose)
            for (int k = 0; k < LM; ++k) {
              st.pose_ent[curB[tmp[x].pose]++] = {tmp[x].wrow + LM + k, tmp[y].wrow + k, st.zero_scalar};
              st.pose_ent[curB[tmp[x].pose]++] = {tmp[y].wrow + LM + k, tmp[x].wrow + k, st.zero_scalar};
            }
      }
    }
  }
  // the diagonal D x D blocks (and the padding identity) are always present
  {
    auto mark = [&](uint32_t pi, uint32_t pj) {
      const uint32_t r0 = pi * D / 64, r1 = (pi * D + D - 1) / 64;
      const uint32_t c0 = pj * D / 64, c1 = (pj * D + D - 1) / 64;
      for (uint32_t r = r0; r <= r1; ++r)
        for (uint32_t c = c0; c <= c1; ++c) { st.tile_nz[(size_t)r * nt + c] = 1; st.tile_nz[(size_t)c * nt + r] = 1; }
    };
    for (uint32_t p = 0; p < st.Pact; ++p) mark(p, p);
    for (uint32_t t = 0; t < nt; ++t) st.tile_nz[(size_t)t * nt + t] = 1;
  }
  mark_stage("pose terms");
  return true;
}

}  // namespace bae
