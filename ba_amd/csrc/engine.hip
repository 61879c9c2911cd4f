// Host side of the gfx950 engine: C-ABI entry points of include/ba_hip.h, the
// per-Solve structure build (observation CSR, incidences, gather lists) and the phase
// drivers that enqueue the kernels of k_proj.hip / k_reduce.hip / k_chol.hip.
#include "engine.h"
#include <chrono>
#include <functional>
#include <memory>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "dmath.h"
#include "dpose.h"

namespace bae {

int Engine::fail(hipError_t e, const char* what) {
  err = std::string(what) + ": " + hipGetErrorString(e);
  return -(int)(e == hipSuccess ? 1 : e);
}
void Engine::prof_collect() {
  auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double& ms, uint32_t& n) {
    for (auto& p : v) {
      (void)hipEventSynchronize(p.second);
      float t = 0;
      (void)hipEventElapsedTime(&t, p.first, p.second);
      ms += t; n += 1;
      (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    v.clear();
  };
  drain(ev_syrk, kstats.syrk_ms, kstats.syrk_launches);
  drain(ev_gather, kstats.gather_ms, kstats.gather_launches);
  drain(ev_landmarks, kstats.landmarks_ms, kstats.landmarks_launches);
  drain(ev_imu, kstats.imu_ms, kstats.imu_launches);
  drain(ev_pose, kstats.pose_blocks_ms, kstats.pose_blocks_launches);
}
int Engine::fail_msg(const char* what) {
  err = what;
  return -1;
}

template <typename T>
static int upload(Engine* e, DBuf<T>& buf, const std::vector<T>& v, size_t min_count = 0) {
  const size_t n = std::max(v.size(), min_count);
  BAE_HIP(buf.alloc(std::max<size_t>(n, 1)));
  if (!v.empty()) {
    // the source is pageable host memory that the caller may free right after this returns (most
    // call sites pass short-lived vectors): the copy must have read it by then, so drain the stream
    BAE_HIP(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
  }
  return 0;
}

// Phase timer on the engine's stream.  The two events come from a pool the engine keeps (creating and
// destroying a pair per phase call cost a small window ~10 us each time).
struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  Engine* eng;
  explicit EventTimer(Engine* e) : s(e->stream), eng(e) {
    a = take();
    b = take();
    (void)hipEventRecord(a, s);
  }
  ~EventTimer() { give(); }  // (an error path that never read the timer)
  // mark(): the end point in stream order; read_ms(): wait for it and read — apart, so that a phase call
  // can mark several intervals and pay for ONE synchronisation at its end
  void mark() { (void)hipEventRecord(b, s); marked = true; }
  double stop_ms() { mark(); return read_ms(); }
  bool marked = false;
  double read_ms() {
    if (!a) return 0.0;
    if (!marked) mark();
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    give();
    return ms;
  }
 private:
  hipEvent_t take() {
    if (!eng->timer_events.empty()) { hipEvent_t ev = eng->timer_events.back(); eng->timer_events.pop_back(); return ev; }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
  }
  void give() {
    if (a) eng->timer_events.push_back(a);
    if (b) eng->timer_events.push_back(b);
    a = b = nullptr;
  }
};

// gravity | measurement noise diag (gyro^2 x3, accel^2 x3) | bias random walk.  Not part of the
// structure: refreshed by every ba_hip_begin_solve, so SetGravity / SetImuCalibration / new option
// sigmas between two Solve() calls need no ba_hip_finalize.
static int upload_imu_consts(Engine* e) {
  const Problem& pb = e->prob;
  std::vector<double> c(15);
  for (int i = 0; i < 3; ++i) {
    c[i] = pb.gravity[i];
    c[3 + i] = e->opt.gyro_sigma * e->opt.gyro_sigma;
    c[6 + i] = e->opt.accel_sigma * e->opt.accel_sigma;
    c[9 + i] = e->opt.gyro_bias_sigma * e->opt.gyro_bias_sigma;
    c[12 + i] = e->opt.accel_bias_sigma * e->opt.accel_bias_sigma;
  }
  if (pb.imu_noise.size() == 12)  // ImuCalibrationT::r / r_b given explicitly (SetImuCalibration)
    for (int i = 0; i < 12; ++i) c[3 + i] = pb.imu_noise[i];
  return upload(e, e->imu_consts, c);
}

// Build everything that depends only on the problem graph (not on the state): the lists of
// structure.h on the host (threads), the pose-pose scatter lists, the tile pattern; then upload.
static int build_structure(Engine* e) {
  Problem& pb = e->prob;
  Structure& st = e->st;
  // BA_HIP_DEBUG_SETUP: wall-clock of the stages of the structure build on stderr
  static const bool dbg_setup = getenv("BA_HIP_DEBUG_SETUP") != nullptr;
  auto t_stage = std::chrono::steady_clock::now();
  auto stage = [&](const char* name) {
    if (!dbg_setup) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[setup] %-28s %.3f s\n", name, std::chrono::duration<double>(now - t_stage).count());
    t_stage = now;
  };
  const int LM = e->lm_dim;
  // the static lists: on the device (structure_dev.hip) unless BA_HIP_HOST_STRUCTURE / debug key 5
  // asks for the host builder (structure.h: the executable specification, same lists)
  static const bool env_host = getenv("BA_HIP_HOST_STRUCTURE") != nullptr;
  const bool host_build = env_host || e->dbg_host_structure;
  if (host_build) {
    std::string err;
    if (!build_lists(pb, LM, e->pose_dim, st, err, stage, e->calib_dim)) {
      e->err = err;
      return -1;
    }
  } else {
    const int brc = build_lists_device(e, stage);
    if (brc) return brc;
  }
  st.n_pp_entries = 0;
  // ---- pose-pose residuals: slots [unary | binary | imu], scatter lists per active pose ------
  const uint32_t nu = pb.num_unary, nbn = pb.num_binary, ni = pb.num_imu, nres = nu + nbn + ni;
  std::vector<uint32_t> res_p1(nres), res_p2(nres, 0xffffffffu);
  for (uint32_t i = 0; i < nu; ++i) res_p1[i] = pb.un_pose[i];
  for (uint32_t i = 0; i < nbn; ++i) { res_p1[nu + i] = pb.bin_p1[i]; res_p2[nu + i] = pb.bin_p2[i]; }
  for (uint32_t i = 0; i < ni; ++i) { res_p1[nu + nbn + i] = pb.imu_p1[i]; res_p2[nu + nbn + i] = pb.imu_p2[i]; }
  for (uint32_t s = 0; s < nres; ++s)
    if (res_p1[s] >= st.P || (res_p2[s] != 0xffffffffu && res_p2[s] >= st.P))
      return e->fail_msg("pose-pose residual references an unknown pose");
  std::vector<uint32_t> pp_ptr(st.Pact + 1, 0);
  for (uint32_t s = 0; s < nres; ++s) {
    const int32_t o1 = st.pose_opt[res_p1[s]];
    const int32_t o2 = res_p2[s] != 0xffffffffu ? st.pose_opt[res_p2[s]] : -1;
    if (o1 >= 0) pp_ptr[o1 + 1]++;
    if (o2 >= 0 && res_p2[s] != res_p1[s]) pp_ptr[o2 + 1]++;
  }
  for (uint32_t p = 0; p < st.Pact; ++p) pp_ptr[p + 1] += pp_ptr[p];
  st.n_pp_entries = st.Pact ? pp_ptr[st.Pact] : 0;
  std::vector<uint4> pp_ent(st.n_pp_entries);
  {
    std::vector<uint32_t> cur(pp_ptr.begin(), pp_ptr.end() - 1);
    for (uint32_t s = 0; s < nres; ++s) {
      const int32_t o1 = st.pose_opt[res_p1[s]];
      const bool two = res_p2[s] != 0xffffffffu && res_p2[s] != res_p1[s];
      const int32_t o2 = two ? st.pose_opt[res_p2[s]] : -1;
      if (o1 >= 0) pp_ent[cur[o1]++] = make_uint4(s, 0, o2 >= 0 ? (uint32_t)o2 : 0xffffffffu, 0);
      if (o2 >= 0) pp_ent[cur[o2]++] = make_uint4(s, 1, o1 >= 0 ? (uint32_t)o1 : 0xffffffffu, 0);
    }
  }

  stage("pose-pose lists");
  // ---- 64x64-tile pattern of S (for the tile-sparse factorisation): build_lists marked the tiles
  // of the projection part and the diagonal; add the pose-pose residual blocks ------------------------
  {
    const uint32_t nt = st.ld / 64, D = (uint32_t)e->pose_dim;
    auto mark = [&](uint32_t pi, uint32_t pj) {  // all tiles the D x D block (pi, pj) overlaps
      const uint32_t r0 = pi * D / 64, r1 = (pi * D + D - 1) / 64;
      const uint32_t c0 = pj * D / 64, c1 = (pj * D + D - 1) / 64;
      for (uint32_t r = r0; r <= r1; ++r)
        for (uint32_t c = c0; c <= c1; ++c) { st.tile_nz[(size_t)r * nt + c] = 1; st.tile_nz[(size_t)c * nt + r] = 1; }
    };
    for (uint32_t s = 0; s < nres; ++s) {
      const int32_t o1 = st.pose_opt[res_p1[s]];
      const int32_t o2 = res_p2[s] != 0xffffffffu ? st.pose_opt[res_p2[s]] : -1;
      if (o1 >= 0 && o2 >= 0) mark((uint32_t)o1, (uint32_t)o2);
    }
    // calibration border: rows np .. np + K - 1 are dense (every pose with a projection residual
    // couples to T_vs, BundleAdjuster.cpp:501-518)
    if (st.K)
      for (uint32_t r = st.np / 64; r <= (st.n - 1) / 64; ++r)
        for (uint32_t c = 0; c <= r; ++c) { st.tile_nz[(size_t)r * nt + c] = 1; st.tile_nz[(size_t)c * nt + r] = 1; }
    e->nzL_valid = false;
  }

  stage("tile pattern");
  // ---- upload ------------------------------------------------------------------------------
  int rc;
#define UP(buf, vec) if ((rc = upload(e, e->buf, vec))) return rc
  if (host_build) {
    UP(pose_opt, st.pose_opt); UP(lm_opt, st.lm_opt);
    UP(lm_ref_pose, pb.lm_ref_pose); UP(lm_ref_cam, pb.lm_ref_cam);
    UP(lm_ptr, st.lm_ptr); UP(obs_z, st.obs_z); UP(obs_pose, st.obs_pose); UP(obs_cam, st.obs_cam);
    UP(obs_lm, st.obs_lm); UP(obs_rid, st.obs_rid); UP(obs_w0, st.obs_w0);
    UP(tile_ptr, st.tile_ptr); UP(pose_ptr, st.pose_ptr); UP(pose_mid, st.pose_mid);
    static_assert(sizeof(U2) == sizeof(uint2) && sizeof(U3) == 3 * sizeof(uint32_t), "list records are plain words");
    BAE_HIP(e->pair_ent.alloc(std::max<size_t>(st.n_pair_entries, 1)));
    if (st.n_pair_entries)
      BAE_HIP(hipMemcpyAsync(e->pair_ent.p, st.pair_ent.get(), st.n_pair_entries * sizeof(uint2), hipMemcpyHostToDevice, e->stream));
    BAE_HIP(e->wave_rng.alloc(std::max<size_t>(st.n_chunks, 1)));
    if (st.n_chunks)
      BAE_HIP(hipMemcpyAsync(e->wave_rng.p, st.wave_rng.data(), (size_t)st.n_chunks * sizeof(uint2), hipMemcpyHostToDevice, e->stream));
    BAE_HIP(e->tile_ref.alloc(std::max<size_t>(st.n_tile_refs, 1)));
    if (st.n_tile_refs)
      BAE_HIP(hipMemcpyAsync(e->tile_ref.p, st.tile_ref.data(), st.n_tile_refs * sizeof(uint2), hipMemcpyHostToDevice, e->stream));
    BAE_HIP(e->pose_ent.alloc(std::max<size_t>(3 * st.n_pose_entries, 1)));
    if (st.n_pose_entries)
      BAE_HIP(hipMemcpyAsync(e->pose_ent.p, st.pose_ent.data(), st.n_pose_entries * sizeof(U3), hipMemcpyHostToDevice, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    // the big host lists are only needed on the device from here on
    st.pair_ent.reset();
    std::vector<U2>().swap(st.tile_ref);
    std::vector<U3>().swap(st.pose_ent);
    std::vector<double>().swap(st.obs_z);
    {
      // conditioning residuals (BundleAdjuster.h:503-510): flags in sorted order
      std::vector<uint8_t> is_cond(st.O, 0), cond_sorted(std::max<uint32_t>(st.O, 1), 0);
      for (uint32_t id : pb.proj_cond) if (id < st.O) is_cond[id] = 1;
      for (uint32_t s = 0; s < st.O; ++s) cond_sorted[s] = is_cond[st.obs_rid[s]];
      UP(obs_cond, cond_sorted);
    }
  }
  UP(pose_active, pb.pose_active);
  UP(un_pose, pb.un_pose); UP(un_t, pb.un_t); UP(un_cov_inv, pb.un_cov_inv); UP(un_rot, pb.un_rot);
  UP(bin_p1, pb.bin_p1); UP(bin_p2, pb.bin_p2); UP(bin_t, pb.bin_t); UP(bin_cov_inv, pb.bin_cov_inv);
  UP(bin_cov_inv_sqrt, pb.bin_cov_inv_sqrt); UP(bin_w, pb.bin_w); UP(bin_rot, pb.bin_rot);
  UP(imu_p1, pb.imu_p1); UP(imu_p2, pb.imu_p2); UP(imu_ptr, pb.imu_ptr); UP(imu_meas, pb.imu_meas);
  UP(pp_ptr, pp_ptr); UP(pp_ent, pp_ent); UP(pp_res_p1, res_p1); UP(pp_res_p2, res_p2);
#undef UP
  {
    std::vector<double> ones(std::max<uint32_t>(nu, 1), 1.0);
    if ((rc = upload(e, e->un_scale, ones))) return rc;
    if ((rc = upload_imu_consts(e))) return rc;
    const size_t nr1 = std::max<size_t>(nres, 1);
    BAE_HIP(e->pp_h.alloc(nr1 * 3 * 225)); BAE_HIP(e->pp_g.alloc(nr1 * 30));
    BAE_HIP(e->pp_dz.alloc(nr1 * 2 * 225)); BAE_HIP(e->pp_info.alloc(nr1 * 225));
    BAE_HIP(e->pp_err.alloc(nr1));
    BAE_HIP(e->imu_cov_inv.alloc(std::max<size_t>(ni, 1) * 225));
    BAE_HIP(hipMemsetAsync(e->imu_cov_inv.p, 0, e->imu_cov_inv.bytes(), e->stream));
  }
  e->tvs_eval = pb.cam_tvs;
  e->tvs_eval_prev = pb.cam_tvs;
  if ((rc = upload_cameras(e, false))) return rc;
  // state
  e->cur = 0;
  e->has_snapshot = false;
  for (int b = 0; b < 2; ++b) {
    BAE_HIP(e->pose_state[b].alloc(std::max<size_t>((size_t)st.P * kPoseState, 1)));
    BAE_HIP(e->lm_x[b].alloc(std::max<size_t>((size_t)st.L * 4, 1)));
    BAE_HIP(e->lm_reliable[b].alloc(std::max<size_t>(st.L, 1)));
  }
  if (st.P) BAE_HIP(hipMemcpyAsync(e->pose_state[0].p, pb.pose_state.data(),
                                   (size_t)st.P * kPoseState * sizeof(double), hipMemcpyHostToDevice, e->stream));
  if ((rc = upload(e, e->lm_xw, pb.lm_xw))) return rc;
  if (st.L) {
    BAE_HIP(hipMemcpyAsync(e->lm_x[0].p, pb.lm_xw.data(), (size_t)st.L * 4 * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    BAE_HIP(hipMemsetAsync(e->lm_reliable[0].p, 1, st.L, e->stream));
  }
  const size_t PC = std::max<size_t>((size_t)st.P * std::max(st.C, 1u), 1);
  BAE_HIP(e->tsw.alloc(PC * kRt)); BAE_HIP(e->tws.alloc(PC * kRt));
  BAE_HIP(e->twp.alloc(std::max<size_t>((size_t)st.P * kRt, 1)));
  BAE_HIP(e->pose_mask.alloc(std::max<size_t>((size_t)st.P + st.Pact, 1)));
  BAE_HIP(hipMemsetAsync(e->pose_mask.p, 0, e->pose_mask.bytes(), e->stream));
  BAE_HIP(e->lm_outliers.alloc(std::max<size_t>(st.L, 1)));
  BAE_HIP(hipMemsetAsync(e->lm_outliers.p, 0, e->lm_outliers.bytes(), e->stream));
  // per-iteration buffers
  const size_t O1 = std::max<size_t>(st.O, 1), L1 = std::max<size_t>(st.L, 1);
  const int LM1 = std::max(LM, 1);
  BAE_HIP(e->obs_e.alloc(O1)); BAE_HIP(e->obs_w.alloc(O1));
  BAE_HIP(e->obs_jl.alloc(O1 * 2 * LM1));
  if (st.O && host_build) BAE_HIP(hipMemcpyAsync(e->obs_w.p, st.obs_w0.data(), (size_t)st.O * sizeof(double),
                                                 hipMemcpyHostToDevice, e->stream));
  BAE_HIP(e->frow.alloc((size_t)st.n_rows * kRow));
  BAE_HIP(hipMemsetAsync(e->frow.p, 0, e->frow.bytes(), e->stream));
  BAE_HIP(e->scal.alloc(std::max<size_t>(st.n_scalars, 2 * O1 + L1 * LM1 + 1)));  // last: the zero scalar
  BAE_HIP(hipMemsetAsync(e->scal.p, 0, e->scal.bytes(), e->stream));
  if (st.K && !e->calib_tvs) {
    // parallel_algos.h:115-118 assigns dTransfer_dparams (2 x NumParams) to a 2 x CalibSize block
    const int nparams = (!pb.cam_model.empty() && pb.cam_model[0] == 1) ? 5 : 4;
    if ((int)st.K != nparams)
      return e->fail_msg("CalibSize must equal the parameter count of camera 0 (LinearCamera 4, FovCamera 5)");
    if (pb.lm_zref.size() != 2 * (size_t)st.L)
      return e->fail_msg("intrinsics calibration needs the reference pixel of every landmark (ba_hip_set_landmark_ref_pixels)");
    if ((rc = upload(e, e->lm_zref, pb.lm_zref))) return rc;
  }
  e->cam_params_prev = pb.cam_params;
  e->cam_w_prev = pb.cam_w;
  if (st.K) {
    BAE_HIP(e->crow.alloc((size_t)std::max<size_t>(st.n_scalars, 1) * kRow));
    BAE_HIP(hipMemsetAsync(e->crow.p, 0, e->crow.bytes(), e->stream));
    BAE_HIP(e->border_blocks.alloc(std::max<size_t>((size_t)st.Pact * 36, 1)));
  }
  BAE_HIP(e->lm_vinv.alloc(L1 * LM1 * LM1)); BAE_HIP(e->lm_bl.alloc(L1 * LM1));
  BAE_HIP(hipMemsetAsync(e->lm_vinv.p, 0, e->lm_vinv.bytes(), e->stream));
  BAE_HIP(hipMemsetAsync(e->lm_bl.p, 0, e->lm_bl.bytes(), e->stream));
  e->A_cleared = nullptr;  // new structure (possibly a new allocation / leading dimension): clear the square once
  BAE_HIP(e->A.alloc((size_t)(st.ld + 1) * st.ld));
  BAE_HIP(e->rhs_p.alloc(st.ld)); BAE_HIP(e->rhs_sc.alloc(st.ld));
  BAE_HIP(e->gn_p.alloc(st.ld)); BAE_HIP(e->step_p.alloc(st.ld));
  BAE_HIP(hipMemsetAsync(e->gn_p.p, 0, e->gn_p.bytes(), e->stream));
  BAE_HIP(hipMemsetAsync(e->step_p.p, 0, e->step_p.bytes(), e->stream));
  const size_t nl = std::max<size_t>((size_t)st.Lact * LM1, 1);
  BAE_HIP(e->gn_l.alloc(nl)); BAE_HIP(e->step_l.alloc(nl));
  BAE_HIP(hipMemsetAsync(e->gn_l.p, 0, e->gn_l.bytes(), e->stream));
  BAE_HIP(hipMemsetAsync(e->step_l.p, 0, e->step_l.bytes(), e->stream));
  // reduction scratch: block partials of the element-wise kernels (up to 4 components) and one
  // partial per linearisation wave
  const size_t nparts = std::max<size_t>({(O1 + 255) / 256, (L1 + 255) / 256, (size_t)(st.ld + 255) / 256, 1});
  BAE_HIP(e->partials.alloc(std::max<size_t>(4 * nparts, st.n_chunks)));
  BAE_HIP(e->scalars_out.alloc(128));  // [0,64) results / staging, [64,128) first-level partial sums
  BAE_HIP(e->hist.alloc(2048 + 8));
  BAE_HIP(e->flags.alloc(16));
  BAE_HIP(hipStreamSynchronize(e->stream));
  stage("upload / device buffers");
  return 0;
}

// Launch order of the tile assembly (k_assemble_tiles).  Only tiles of the factor's pattern are
// written every iteration: the others are zero since the one-off clear of the square and nobody ever
// writes them (the factorisation skips structurally zero tiles).  Order: blocks b, b + 8, .. share
// an XCD; XCD x gets the tile columns I = x, x + 8, .. and walks each from the diagonal down, so the
// tiles in flight on one XCD gather the same pose group's rows out of its L2.
int build_tile_order(Engine* e) {
  const Structure& st = e->st;
  if (!e->nzL_valid) {
    int rc = factor_tile_pattern(e);
    if (rc) return rc;
  }
  const uint32_t nt = st.ld / 64;
  // distributed solve with the sparse exchange of S: only the tiles this rank owns or sends (k_chol.hip)
  std::vector<uint8_t> need;
  const int restricted = e->dbg_all_tiles ? 0 : dist_assembly_tiles(e, &need);
  if (restricted < 0) return restricted;
  const uint64_t plan_key = restricted ? e->dist_plan_version : ~0ull;
  if (e->tile_order_version == e->nzL_version && e->tile_order.p && e->tile_order_plan == plan_key) return 0;
  const bool pat = !e->dbg_all_tiles && e->nzL_host.size() == (size_t)nt * nt;
  const std::vector<uint8_t>& which = restricted ? need : e->nzL_host;
  std::vector<uint32_t> order;
  size_t longest = 0;
  if (e->dbg_tile_order == 1) {
    std::vector<std::vector<uint32_t>> queue(8);
    for (uint32_t I = 0; I < nt; ++I)
      for (uint32_t J = I; J < nt; ++J)
        if (!pat || which[(size_t)J * nt + I]) queue[I % 8].push_back((uint32_t)((uint64_t)J * (J + 1) / 2 + I));
    for (auto& q : queue) longest = std::max(longest, q.size());
    order.assign(std::max<size_t>(8 * longest, 1), 0xffffffffu);
    for (uint32_t x = 0; x < 8; ++x)
      for (size_t k = 0; k < queue[x].size(); ++k) order[8 * k + x] = queue[x][k];
  } else {
    // row-major over the lower tiles: a tile row mixes gather-heavy tiles near the diagonal band with
    // fill-only tiles, so latency-bound and bandwidth-bound workgroups overlap on every CU
    for (uint32_t J = 0; J < nt; ++J)
      for (uint32_t I = 0; I <= J; ++I)
        if (!pat || which[(size_t)J * nt + I]) order.push_back((uint32_t)((uint64_t)J * (J + 1) / 2 + I));
    longest = (order.size() + 7) / 8;
    order.resize(std::max<size_t>(8 * longest, 1), 0xffffffffu);
  }
  e->n_tile_order = (uint32_t)(8 * longest);
  int rc = upload(e, e->tile_order, order);
  if (rc) return rc;
  if ((rc = build_tile_desc(e))) return rc;
  e->tile_order_version = e->nzL_version;
  e->tile_order_plan = plan_key;
  return 0;
}

// cameras: params(4) | T_vs Rt(12) | T_sv Rt(12) | T_vs as t,q (7) | w | model (kCamRec doubles); `cam` from
// the rig (prob.cam_tvs), cam_eval (calibration only) from tvs_eval
int upload_cameras(Engine* e, bool eval_only) {
  const Problem& pb = e->prob;
  const uint32_t C = pb.num_cams;
  auto table = [&](const std::vector<double>& tvs) {
    std::vector<double> cam((size_t)C * kCamRec, 0.0);
    for (uint32_t c = 0; c < C; ++c) {
      double* o = &cam[(size_t)c * kCamRec];
      const double* t = &tvs[(size_t)c * 7];
      for (int i = 0; i < 4; ++i) o[i] = pb.cam_params[(size_t)c * 4 + i];
      bad::Rt vs;
      vs.R = bad::quat_to_rot(t[3], t[4], t[5], t[6]);
      vs.t = bad::v3(t[0], t[1], t[2]);
      const bad::Rt sv = bad::inverse(vs);
      for (int i = 0; i < 9; ++i) { o[4 + i] = vs.R.m[i]; o[16 + i] = sv.R.m[i]; }
      o[13] = vs.t.x; o[14] = vs.t.y; o[15] = vs.t.z;
      o[25] = sv.t.x; o[26] = sv.t.y; o[27] = sv.t.z;
      for (int i = 0; i < 7; ++i) o[28 + i] = t[i];
      const bool fov = c < pb.cam_model.size() && pb.cam_model[c] == 1;
      o[35] = fov ? pb.cam_w[c] : 0.0;
      o[36] = fov ? 1.0 : 0.0;
    }
    return cam;
  };
  e->has_fov = false;
  for (uint32_t c = 0; c < C && c < pb.cam_model.size(); ++c) e->has_fov |= pb.cam_model[c] == 1;
  int rc;
  if (!eval_only && (rc = upload(e, e->cam, table(pb.cam_tvs)))) return rc;
  if (e->calib_dim && (rc = upload(e, e->cam_eval, table(e->tvs_eval)))) return rc;
  return 0;
}

static int set_masks_device(Engine* e, const std::vector<uint16_t>& by_id) {
  const Structure& st = e->st;
  std::vector<uint16_t> m((size_t)st.P + st.Pact, 0);
  for (uint32_t p = 0; p < st.P; ++p) {
    m[p] = by_id[p];
    if (st.pose_opt[p] >= 0) m[st.P + st.pose_opt[p]] = by_id[p];
  }
  BAE_HIP(hipMemcpyAsync(e->pose_mask.p, m.data(), m.size() * sizeof(uint16_t), hipMemcpyHostToDevice,
                         e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

}  // namespace bae

using namespace bae;

// ===================================================================================
extern "C" {

int ba_hip_create(int lm_dim, int pose_dim, int device, void* stream, ba_hip_engine** out) {
  if (!out) return -1;
  *out = nullptr;
  if (!(lm_dim == 0 || lm_dim == 1 || lm_dim == 3)) return -1;
  if (!(pose_dim == 6 || pose_dim == 9 || pose_dim == 15)) return -1;
  int ndev = 0;
  hipError_t err = hipGetDeviceCount(&ndev);
  if (err != hipSuccess || ndev == 0) return -(int)(err == hipSuccess ? hipErrorNoDevice : err);
  if (device < 0 || device >= ndev) return -(int)hipErrorInvalidDevice;
  err = hipSetDevice(device);
  if (err != hipSuccess) return -(int)err;
  Engine* e = new Engine();
  e->lm_dim = lm_dim; e->pose_dim = pose_dim; e->device = device;
  memset(&e->opt, 0, sizeof(e->opt));
  e->opt.projection_outlier_threshold = 1.0;
  e->opt.use_robust_norm_for_proj_residuals = 1;
  e->opt.use_triangular_matrices = 1;
  e->opt.gyro_sigma = 5.3088444e-5; e->opt.accel_sigma = 0.001883649;
  e->opt.gyro_bias_sigma = 1.4125375e-4; e->opt.accel_bias_sigma = 1.2589254e-2;
  memset(&e->timers, 0, sizeof(e->timers));
  if (stream) {
    e->stream = (hipStream_t)stream;
  } else {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically smallest = highest priority
    err = hipStreamCreateWithPriority(&e->stream, hipStreamDefault, hi);
    if (err != hipSuccess) { delete e; return -(int)err; }
    e->own_stream = true;
  }
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // Second stream for the bulk trailing updates of the factorisation: lowest priority, so the
    // serial panel chain on the main stream wins the dispatch.  (CU masks for it were measured in
    // round 1 — every mask was slower than none.)
    if (!e->stream2) err = hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, lo);
    if (err != hipSuccess) { delete e; return -(int)err; }
  }
  *out = reinterpret_cast<ba_hip_engine*>(e);
  return 0;
}

// (every entry point starts outside a deferred-sum scope, whatever an earlier call's error path left behind)
#define ENG(h) Engine* e = reinterpret_cast<Engine*>(h); e->defer_active = false; e->defer_n = 0

void ba_hip_destroy(ba_hip_engine* h) {
  if (!h) return;
  ENG(h);
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
#define REL(b) e->b.release()
  REL(cam); REL(cam_eval); REL(lm_zref); REL(crow); REL(border_blocks); REL(calib_partials); REL(pose_opt); REL(lm_opt); REL(pose_mask); REL(lm_ref_pose); REL(lm_ref_cam);
  REL(lm_ptr); REL(obs_z); REL(obs_pose); REL(obs_cam); REL(obs_lm); REL(obs_rid); REL(obs_w0); REL(obs_cond);
  REL(wave_rng); REL(tile_order); REL(tile_desc); REL(tile_ptr); REL(tile_ref); REL(pair_ent); REL(pose_ptr); REL(pose_mid); REL(pose_ent);
  REL(imu_frozen); REL(imu_steps); REL(imu_cov_done); REL(pose_cam);
  REL(packed); REL(nzL); REL(dist_msg); REL(dist_rows); REL(dist_srows);
  REL(dist_tiles); REL(dist_sq_list); REL(dist_pairs); REL(dist_sp_srect); REL(dist_sp_rrect); REL(dist_usend); REL(dist_urecv); REL(dist_ssend); REL(dist_srecv); REL(dist_back);
  for (int b = 0; b < 2; ++b) { REL(pose_state[b]); REL(lm_x[b]); REL(lm_reliable[b]); }
  REL(lm_xw); REL(tsw); REL(tws); REL(twp); REL(lm_outliers); REL(obs_e); REL(obs_w); REL(obs_e_state[0]); REL(obs_e_state[1]); REL(obs_jl);
  REL(frow); REL(diag_blocks); REL(scal); REL(lm_vinv); REL(lm_bl); REL(A); REL(A_keep); REL(rhs_p); REL(rhs_sc); REL(gn_p);
  REL(gn_l); REL(step_p); REL(step_l); REL(invdiag); REL(partials); REL(scalars_out); REL(hist);
  REL(flags); REL(pivot_floor);
  REL(pose_active); REL(un_pose); REL(un_t); REL(un_cov_inv); REL(un_scale); REL(un_rot);
  REL(bin_p1); REL(bin_p2); REL(bin_t); REL(bin_cov_inv); REL(bin_cov_inv_sqrt); REL(bin_w); REL(bin_rot);
  REL(imu_p1); REL(imu_p2); REL(imu_ptr); REL(imu_meas); REL(imu_consts); REL(imu_cov_inv);
  REL(pp_h); REL(pp_g); REL(pp_dz); REL(pp_info); REL(pp_err); REL(pp_ptr); REL(pp_res_p1);
  REL(pp_res_p2); REL(pp_ent);
#undef REL
  comm_release(e);
  if (e->ev_imu_done) { (void)hipEventDestroy(e->ev_imu_done); (void)hipEventDestroy(e->ev_imu_start); }
  if (e->ev_fork) { (void)hipEventDestroy(e->ev_fork); (void)hipEventDestroy(e->ev_join); }
  for (hipEvent_t ev : e->timer_events) (void)hipEventDestroy(ev);
  e->timer_events.clear();
  for (hipEvent_t ev : e->ev_panel) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->ev_bulk) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->ev_dist) (void)hipEventDestroy(ev);
  if (e->stream3) (void)hipStreamDestroy(e->stream3);
  if (e->stream4) (void)hipStreamDestroy(e->stream4);
  if (e->stream2) (void)hipStreamDestroy(e->stream2);
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

const char* ba_hip_last_error(const ba_hip_engine* h) {
  return h ? reinterpret_cast<const Engine*>(h)->err.c_str() : "null engine";
}

int ba_hip_set_options(ba_hip_engine* h, const ba_hip_options* o) {
  ENG(h);
  e->opt = *o;
  return 0;
}

int ba_hip_set_cameras(ba_hip_engine* h, uint32_t n, const double* params4, const double* t_vs7) {
  ENG(h);
  e->prob.num_cams = n;
  e->prob.cam_params.assign(params4, params4 + 4 * (size_t)n);
  e->prob.cam_tvs.assign(t_vs7, t_vs7 + 7 * (size_t)n);
  e->prob.cam_model.assign(n, 0);
  e->prob.cam_w.assign(n, 0.0);
  e->finalized = false;
  return 0;
}

int ba_hip_set_camera_models(ba_hip_engine* h, uint32_t n, const int32_t* model, const double* w) {
  ENG(h);
  if (n != e->prob.num_cams) return e->fail_msg("ba_hip_set_camera_models: one entry per camera of ba_hip_set_cameras expected");
  for (uint32_t c = 0; c < n; ++c) {
    if (model[c] != 0 && model[c] != 1) return e->fail_msg("camera model: 0 (LinearCamera) or 1 (FovCamera)");
    if (model[c] == 1 && !(w && std::isfinite(w[c]))) return e->fail_msg("FovCamera: a finite distortion parameter w is needed");
  }
  e->prob.cam_model.assign(model, model + n);
  e->prob.cam_w.assign(n, 0.0);
  for (uint32_t c = 0; c < n; ++c) if (model[c] == 1) e->prob.cam_w[c] = w[c];
  e->finalized = false;
  return 0;
}
int ba_hip_get_camera_fov(ba_hip_engine* h, uint32_t n, double* w) {
  ENG(h);
  if (!w || n != e->prob.cam_w.size()) return e->fail_msg("ba_hip_get_camera_fov: camera count mismatch");
  for (size_t c = 0; c < e->prob.cam_w.size(); ++c) w[c] = e->prob.cam_w[c];
  return 0;
}

int ba_hip_set_pose_cam_params(ba_hip_engine* h, uint32_t n, const double* params4) {
  ENG(h);
  e->err_cache_clear();   // (uploaded without a rebuild: the residuals change)
  if (n == 0 || !params4) {
    e->prob.pose_cam_params.clear();
    return 0;
  }
  e->prob.pose_cam_params.assign(params4, params4 + 4 * (size_t)n);
  return upload(e, e->pose_cam, e->prob.pose_cam_params);
}

int ba_hip_set_poses(ba_hip_engine* h, uint32_t n, const double* t_wp7, const double* v_w3,
                     const double* b6, const uint8_t* is_active) {
  ENG(h);
  Problem& pb = e->prob;
  pb.num_poses = n;
  pb.pose_state.assign((size_t)n * kPoseState, 0.0);
  pb.pose_active.assign(n, 1);
  for (uint32_t p = 0; p < n; ++p) {
    double* s = &pb.pose_state[(size_t)p * kPoseState];
    for (int i = 0; i < 7; ++i) s[i] = t_wp7[(size_t)p * 7 + i];
    if (v_w3) for (int i = 0; i < 3; ++i) s[7 + i] = v_w3[(size_t)p * 3 + i];
    if (b6) for (int i = 0; i < 6; ++i) s[10 + i] = b6[(size_t)p * 6 + i];
    if (is_active) pb.pose_active[p] = is_active[p] ? 1 : 0;
  }
  e->finalized = false;
  return 0;
}

int ba_hip_set_landmarks(ba_hip_engine* h, uint32_t n, const double* x_w4, const uint32_t* ref_pose_id,
                         const uint32_t* ref_cam_id, const uint8_t* is_active) {
  ENG(h);
  Problem& pb = e->prob;
  pb.num_lms = n;
  pb.lm_xw.assign(x_w4, x_w4 + 4 * (size_t)n);
  pb.lm_ref_pose.assign(ref_pose_id, ref_pose_id + n);
  if (ref_cam_id) pb.lm_ref_cam.assign(ref_cam_id, ref_cam_id + n);
  else pb.lm_ref_cam.assign(n, 0);
  if (is_active) pb.lm_active.assign(is_active, is_active + n);
  else pb.lm_active.assign(n, 1);
  e->finalized = false;
  return 0;
}

int ba_hip_set_projection_residuals(ba_hip_engine* h, uint32_t n, const double* z2,
                                    const uint32_t* meas_pose_id, const uint32_t* landmark_id,
                                    const uint32_t* cam_id, const double* weight) {
  ENG(h);
  Problem& pb = e->prob;
  pb.num_proj = n;
  pb.proj_z.assign(z2, z2 + 2 * (size_t)n);
  pb.proj_pose.assign(meas_pose_id, meas_pose_id + n);
  pb.proj_lm.assign(landmark_id, landmark_id + n);
  if (cam_id) pb.proj_cam.assign(cam_id, cam_id + n);
  else pb.proj_cam.assign(n, 0);
  if (weight) pb.proj_w.assign(weight, weight + n);
  else pb.proj_w.assign(n, 1.0);
  e->finalized = false;
  return 0;
}

int ba_hip_set_conditioning_residuals(ba_hip_engine* h, uint32_t n, const uint32_t* residual_id) {
  ENG(h);
  e->prob.proj_cond.assign(residual_id, residual_id + n);
  e->finalized = false;
  return 0;
}

int ba_hip_set_unary_residuals(ba_hip_engine* h, uint32_t n, const uint32_t* pose_id,
                               const double* t_wp7, const double* cov_inv36,
                               const uint8_t* use_rotation) {
  ENG(h);
  Problem& pb = e->prob;
  pb.num_unary = n;
  pb.un_pose.assign(pose_id, pose_id + n);
  pb.un_t.assign(t_wp7, t_wp7 + 7 * (size_t)n);
  pb.un_cov_inv.assign(cov_inv36, cov_inv36 + 36 * (size_t)n);
  if (use_rotation) pb.un_rot.assign(use_rotation, use_rotation + n);
  else pb.un_rot.assign(n, 1);
  e->finalized = false;
  return 0;
}

int ba_hip_set_binary_residuals(ba_hip_engine* h, uint32_t n, const uint32_t* pose1_id,
                                const uint32_t* pose2_id, const double* t_12_7,
                                const double* cov_inv36, const double* cov_inv_sqrt36,
                                const double* weight, const uint8_t* use_rotation) {
  ENG(h);
  Problem& pb = e->prob;
  pb.num_binary = n;
  pb.bin_p1.assign(pose1_id, pose1_id + n);
  pb.bin_p2.assign(pose2_id, pose2_id + n);
  pb.bin_t.assign(t_12_7, t_12_7 + 7 * (size_t)n);
  pb.bin_cov_inv.assign(cov_inv36, cov_inv36 + 36 * (size_t)n);
  pb.bin_cov_inv_sqrt.assign(cov_inv_sqrt36, cov_inv_sqrt36 + 36 * (size_t)n);
  if (weight) pb.bin_w.assign(weight, weight + n);
  else pb.bin_w.assign(n, 1.0);
  if (use_rotation) pb.bin_rot.assign(use_rotation, use_rotation + n);
  else pb.bin_rot.assign(n, 1);
  e->finalized = false;
  return 0;
}

int ba_hip_set_imu_residuals(ba_hip_engine* h, uint32_t n, const uint32_t* pose1_id,
                             const uint32_t* pose2_id, const uint32_t* meas_ptr, const double* meas7,
                             const double* weight) {
  ENG(h);
  Problem& pb = e->prob;
  pb.num_imu = n;
  pb.imu_p1.assign(pose1_id, pose1_id + n);
  pb.imu_p2.assign(pose2_id, pose2_id + n);
  pb.imu_ptr.assign(meas_ptr, meas_ptr + n + 1);
  pb.imu_meas.assign(meas7, meas7 + 7 * (size_t)(n ? meas_ptr[n] : 0));
  if (weight) pb.imu_w.assign(weight, weight + n);
  else pb.imu_w.assign(n, 1.0);
  e->finalized = false;
  return 0;
}

// Host-side RK4 integration of IMU samples with the code the device kernels use (dpose.h):
// ImuResidualT::IntegrateResidual without Jacobians (Types.h:662-738).
int ba_hip_integrate_imu(const double t_wp7[7], const double v_w3[3], const double bg3[3], const double ba3[3],
                         const double g3[3], const double* meas7, uint32_t nmeas, double* states10) {
  if (!t_wp7 || !v_w3 || !bg3 || !ba3 || !g3 || !states10 || (nmeas && !meas7)) return -1;
  bad::ImuState s;
  for (int i = 0; i < 3; ++i) { s.t[i] = t_wp7[i]; s.v[i] = v_w3[i]; }
  for (int i = 0; i < 4; ++i) s.q[i] = t_wp7[3 + i];
  auto put = [&](uint32_t k) {
    double* o = states10 + 10 * (size_t)k;
    for (int i = 0; i < 3; ++i) { o[i] = s.t[i]; o[7 + i] = s.v[i]; }
    for (int i = 0; i < 4; ++i) o[3 + i] = s.q[i];
  };
  put(0);
  for (uint32_t i = 1; i < nmeas; ++i) {
    s = bad::integrate_imu(s, meas7 + 7 * (size_t)(i - 1), meas7 + 7 * (size_t)i, bg3, ba3, g3, false, nullptr,
                           nullptr, nullptr, nullptr);
    put(i);
  }
  return 0;
}

// The same integration with the Jacobians of the reference's signature (Types.h:662-738): dpose_db
// (10 x 6, over the gyro / accelerometer biases), dpose_dpose (10 x 10, over the start state
// [t q v]) and the covariance c_res (10 x 10, in/out: C <- F C F^T + G R G^T per step) — formed, as in
// the reference, only when one of the two Jacobians is asked for AND the noise diagonal r6 is given.
int ba_hip_integrate_imu_jacobians(const double t_wp7[7], const double v_w3[3], const double bg3[3],
                                   const double ba3[3], const double g3[3], const double* meas7, uint32_t nmeas,
                                   const double r6[6], double* states10, double* dpose_db60,
                                   double* dpose_dpose100, double* c_res100) {
  if (!t_wp7 || !v_w3 || !bg3 || !ba3 || !g3 || !states10 || (nmeas && !meas7)) return -1;
  bad::ImuState s;
  for (int i = 0; i < 3; ++i) { s.t[i] = t_wp7[i]; s.v[i] = v_w3[i]; }
  for (int i = 0; i < 4; ++i) s.q[i] = t_wp7[3 + i];
  auto put = [&](uint32_t k) {
    double* o = states10 + 10 * (size_t)k;
    for (int i = 0; i < 3; ++i) { o[i] = s.t[i]; o[7 + i] = s.v[i]; }
    for (int i = 0; i < 4; ++i) o[3 + i] = s.q[i];
  };
  put(0);
  bad::DM<10, 6> db, dy_db;
  bad::DM<10, 10> dd, dy_dy, cov;
  db.zero();
  dd.identity();
  const bool jac = (dpose_db60 || dpose_dpose100) && r6;
  if (c_res100) for (int i = 0; i < 100; ++i) cov.m[i] = c_res100[i];
  for (uint32_t i = 1; i < nmeas; ++i) {
    const double* z0 = meas7 + 7 * (size_t)(i - 1);
    const double* z1 = meas7 + 7 * (size_t)i;
    if (jac) {
      s = bad::integrate_imu(s, z0, z1, bg3, ba3, g3, true, &dy_db, &dy_dy, c_res100 ? &cov : nullptr, r6);
      db = bad::madd(dy_db, bad::mm(dy_dy, db));  // Types.h:712-714
      dd = bad::mm(dy_dy, dd);                    // Types.h:716-718
    } else {
      s = bad::integrate_imu(s, z0, z1, bg3, ba3, g3, false, nullptr, nullptr, nullptr, nullptr);
    }
    put(i);
  }
  if (dpose_db60) for (int i = 0; i < 60; ++i) dpose_db60[i] = db.m[i];
  if (dpose_dpose100) for (int i = 0; i < 100; ++i) dpose_dpose100[i] = dd.m[i];
  if (c_res100 && jac) for (int i = 0; i < 100; ++i) c_res100[i] = cov.m[i];
  return 0;
}

// ImuResidualT::GetPoseDerivative (Types.h:376-416): k = d/dt [t; rotation vector; v] of the state at
// time z_start.time + dt between two samples, with dk_db (9 x 6) and dk_dx (9 x 10), both optional
int ba_hip_imu_pose_derivative(const double state10[10], const double g3[3], const double z_start7[7],
                               const double z_end7[7], const double bg3[3], const double ba3[3], double dt,
                               double k9[9], double* dk_db54, double* dk_dx90) {
  if (!state10 || !g3 || !z_start7 || !z_end7 || !bg3 || !ba3 || !k9) return -1;
  bad::ImuState s;
  for (int i = 0; i < 3; ++i) { s.t[i] = state10[i]; s.v[i] = state10[7 + i]; }
  for (int i = 0; i < 4; ++i) s.q[i] = state10[3 + i];
  bad::DM<9, 6> a;
  bad::DM<9, 10> b;
  bad::pose_derivative(s, g3, z_start7, z_end7, bg3, ba3, dt, k9, dk_db54 ? &a : nullptr, dk_dx90 ? &b : nullptr);
  if (dk_db54) for (int i = 0; i < 54; ++i) dk_db54[i] = a.m[i];
  if (dk_dx90) for (int i = 0; i < 90; ++i) dk_dx90[i] = b.m[i];
  return 0;
}

// ImuResidualT::IntegratePose (Types.h:324-373): y = state advanced by k * dt (q <- exp(k_w dt) q, not
// renormalised), with dy_dk (10 x 9) and the quaternion block dy_dy (4 x 4), both optional
int ba_hip_imu_integrate_pose(const double state10[10], const double k9[9], double dt, double out10[10],
                              double* dy_dk90, double* dy_dy16) {
  if (!state10 || !k9 || !out10) return -1;
  bad::ImuState s;
  for (int i = 0; i < 3; ++i) { s.t[i] = state10[i]; s.v[i] = state10[7 + i]; }
  for (int i = 0; i < 4; ++i) s.q[i] = state10[3 + i];
  bad::DM<10, 9> a;
  bad::DM<4, 4> b;
  const bad::ImuState y = bad::integrate_pose(s, k9, dt, dy_dk90 ? &a : nullptr, dy_dy16 ? &b : nullptr);
  for (int i = 0; i < 3; ++i) { out10[i] = y.t[i]; out10[7 + i] = y.v[i]; }
  for (int i = 0; i < 4; ++i) out10[3 + i] = y.q[i];
  if (dy_dk90) for (int i = 0; i < 90; ++i) dy_dk90[i] = a.m[i];
  if (dy_dy16) for (int i = 0; i < 16; ++i) dy_dy16[i] = b.m[i];
  return 0;
}

// The Lie-group / quaternion helpers of the reference's Utils.h on the host (include/ba/Utils.h wraps
// them with the reference's names): one dispatcher over the functions the kernels use (dmath.h, dpose.h).
// a / b: the arguments as doubles (transforms as [t(3) q(4)], quaternions x,y,z,w); out: the result,
// row-major.  Returns the number of doubles written, -1 for an unknown op or a missing argument.
int ba_hip_lie(int op, const double* a, const double* b, double* out) {
  using namespace bad;
  if (!a || !out) return -1;
  auto put = [&](const double* m, int n) { for (int i = 0; i < n; ++i) out[i] = m[i]; return n; };
  const bool two = op == 5 || (op >= 7 && op <= 11) || op == 14 || op == 15 || op == 17;
  if (two && !b) return -1;
  switch (op) {
    case 1: return put(dlog_dq(a).m, 12);                                     // Utils.h:137-185
    case 2: return put(dq_exp_dw(v3(a[0], a[1], a[2])).m, 12);                // :252-266
    case 3: return put(qR(a).m, 16);                                          // dq1q2_dq1(q2) :286-291
    case 4: return put(qL(a).m, 16);                                          // dq1q2_dq2(q1) :277-282
    case 5: return put(dqx_dq(a, v3(b[0], b[1], b[2])).m, 12);                // :295-312
    case 6: { const M3 R = quat_to_rot(a[0], a[1], a[2], a[3]); return put(R.m, 9); }  // dqx_dx :316-333
    case 7: log_decoupled(tq_from7(a), tq_from7(b), out); return 6;           // :354-360
    case 8: {                                                                 // exp_decoupled :364-369
      double qe[4], q[4];
      so3_exp(v3(b[3], b[4], b[5]), qe);
      quat_mul(a + 3, qe, q);
      quat_normalize(q);
      for (int i = 0; i < 3; ++i) out[i] = a[i] + b[i];
      for (int i = 0; i < 4; ++i) out[3 + i] = q[i];
      return 7;
    }
    case 9: return put(dlog_decoupled_dx(tq_from7(a), tq_from7(b)).m, 36);    // :374-384
    case 10: return put(dLog_decoupled_dt1(tq_from7(a), tq_from7(b)).m, 42);  // :388-397
    case 11: return put(dlog_decoupled_dt2(tq_from7(a), tq_from7(b)).m, 42);  // :401-447
    case 12: return put(dexp_decoupled_dx(tq_from7(a)).m, 42);                // :451-489
    case 13: return put(dinv_exp_decoupled_dx(tq_from7(a)).m, 42);            // :493-536
    case 14: {                                                                // dt_x_dt(t, x) 4 x 7 :540-583
      double J[28] = {0};
      const DM<3, 4> d = dqx_dq(a + 3, v3(b[0], b[1], b[2]));
      for (int r = 0; r < 3; ++r) {
        J[r * 7 + r] = b[3];
        for (int c = 0; c < 4; ++c) J[r * 7 + 3 + c] = d(r, c);
      }
      return put(J, 28);
    }
    case 15: return put(dt1_t2_dt1(tq_from7(a), tq_from7(b)).m, 49);          // :587-639
    case 16: return put(dt1_t2_dt2(tq_from7(a)).m, 49);                       // :643-694
    case 17: {                                                                // MultHomogeneous :72-82
      const V3 r = qrot(a + 3, v3(b[0], b[1], b[2]));
      out[0] = r.x + a[0] * b[3]; out[1] = r.y + a[1] * b[3]; out[2] = r.z + a[2] * b[3]; out[3] = b[3];
      return 4;
    }
  }
  return -1;
}

int ba_hip_set_imu_noise(ba_hip_engine* h, const double r6[6], const double rb6[6]) {
  ENG(h);
  e->prob.imu_noise.clear();
  if (r6 && rb6) {
    e->prob.imu_noise.assign(r6, r6 + 6);
    e->prob.imu_noise.insert(e->prob.imu_noise.end(), rb6, rb6 + 6);
  }
  return 0;  // not structural: uploaded by the next ba_hip_begin_solve
}

int ba_hip_set_inertial_covariance_once(ba_hip_engine* h, int on, int reset) {
  ENG(h);
  e->imu_cov_once = on != 0;
  if (reset || !on) e->imu_cov_count = 0;
  return 0;
}

int ba_hip_set_gravity(ba_hip_engine* h, const double g3[3]) {
  ENG(h);
  for (int i = 0; i < 3; ++i) e->prob.gravity[i] = g3[i];
  return 0;
}

int ba_hip_finalize(ba_hip_engine* h) {
  ENG(h);
  e->dog_jrhs_valid = false;  // the factor rows are rebuilt: no cached sum survives
  e->err_cache_clear();
  BAE_HIP(hipSetDevice(e->device));
  Problem& pb = e->prob;
  if (pb.pose_active.size() != pb.num_poses) pb.pose_active.assign(pb.num_poses, 1);
  if (pb.lm_active.size() != pb.num_lms) pb.lm_active.assign(pb.num_lms, 1);
  int rc = build_structure(e);
  if (rc) return rc;
  e->finalized = true;
  return 0;
}

#define NEED_FINAL() \
  if (!e->finalized) return e->fail_msg("ba_hip_finalize has not been called")

int ba_hip_get_conditioning_error(ba_hip_engine* h, double* proj_sq_sum) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  *proj_sq_sum = 0.0;
  if (e->prob.proj_cond.empty() || e->st.O == 0) return 0;
  int rc = launch_pose_prep(e);
  if (rc) return rc;
  if ((rc = launch_residuals(e, 2))) return rc;
  return sum_partials(e, (e->st.O + 255) / 256, 1, proj_sq_sum);
}

int ba_hip_begin_solve(ba_hip_engine* h) {
  ENG(h);
  NEED_FINAL();
  e->dog_jrhs_valid = false;
  e->err_cache_clear();   // x_s is re-derived from x_w, the cameras may have been re-uploaded
  e->err_cache_off = getenv("BA_HIP_NO_ERR_CACHE") != nullptr;
  BAE_HIP(hipSetDevice(e->device));
  if (!e->prob.pose_cam_params.empty() && e->prob.pose_cam_params.size() != 4 * (size_t)e->prob.num_poses)
    return e->fail_msg("per-pose camera parameters: one [fx,fy,u0,v0] per pose expected");
  if (!e->prob.pose_cam_params.empty() && e->has_fov)
    return e->fail_msg("per-pose camera parameters are offered for LinearCamera rigs only");
  int rc = upload_imu_consts(e);
  if (rc) return rc;
  // (calibration: the tables of k_pose_prep keep the T_vs they were last built with across Solve()
  // calls — after a Solve() that ended in a rejected step the reference's poses carry their cached
  // T_sw of the T_vs before that step into the next call, until a step is applied: tvs_eval is not
  // resynchronised with the rig here)
  rc = launch_pose_prep(e);
  if (rc) return rc;
  rc = launch_begin_solve(e);
  if (rc) return rc;
  BAE_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

int ba_hip_set_pose_masks(ba_hip_engine* h, uint32_t n, const uint16_t* masks) {
  ENG(h);
  NEED_FINAL();
  if (n != e->st.P) return e->fail_msg("mask count != pose count");
  e->dog_jrhs_valid = false;
  return set_masks_device(e, std::vector<uint16_t>(masks, masks + n));
}

int ba_hip_linearize(ba_hip_engine* h, ba_hip_errors* out) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  const Structure& st = e->st;
  ba_hip_errors errs = {0, 0, 0, 0};
  int rc;
  e->dog_jrhs_valid = false;
  // projection errors at the linearisation point -> Huber sigma
  EventTimer t_j(e);
  if ((rc = launch_pose_prep(e))) return rc;
  // the errors of the Huber median: left by the last EvaluateResiduals at this state, or computed now
  const bool cached = e->err_cache_on() && e->obs_e_valid[e->cur] && e->obs_e_state[e->cur].n >= st.O && st.O > 0;
  const double* med_in = cached ? e->obs_e_state[e->cur].p : e->obs_e.p;
  if (!cached && (rc = launch_residuals(e, 0))) return rc;
  t_j.mark();
  EventTimer t_r(e);
  double c_huber = 0.0;
  uint64_t n_total = st.O;
  if (e->sharded()) {
    double cnt = (double)st.O;
    BAE_HIP(hipMemcpy(e->scalars_out.p, &cnt, sizeof(double), hipMemcpyHostToDevice));
    if (shard_allreduce(e, e->scalars_out.p, 1, 0) != 0) return e->fail_msg("allreduce hook failed");
    BAE_HIP(hipMemcpy(&cnt, e->scalars_out.p, sizeof(double), hipMemcpyDeviceToHost));
    n_total = (uint64_t)(cnt + 0.5);
  }
  if (n_total > 0) {
    double med = 0.0;
    // std::nth_element at floor(N * 0.5): the upper median (BundleAdjuster.cpp:1356-1358)
    if ((rc = select_kth(e, med_in, st.O, (uint64_t)std::floor(n_total * 0.5), &med))) return rc;
    c_huber = 1.2107 * std::sqrt(med);
  }
  t_r.mark();
  // The inertial kernels run on the second stream.  Sliding windows (few samples: latency-bound wavefront forms)
  // start them right away, beside the projection linearisation.  Large problems start them BEHIND k_linearize, under
  // the tile assembly: the lane-per-sample form keeps 28 KB of private memory per lane, and its scratch traffic beside
  // the bandwidth-bound k_linearize slowed that kernel 4-5x (configs[4]: 3.8-5.6 ms against 1.05 ms alone) for the
  // same wall time (profiles/r03_imu_ab.txt); the assembly is latency-bound on its row gathers
  // and four times as long as both inertial passes.
  const bool imu_late = e->prob.imu_meas.size() / 7 > 4096;
  if (!imu_late && (rc = launch_imu_early(e, c_huber))) return rc;
  EventTimer t_l(e);
  // from here on every sum stays on the device until ONE copy at the end of the call (defer_flush): the
  // assembly and the pose-pose kernels are enqueued without waiting for the linearisation to finish
  double* hs = e->eval_h;  // proj | unary | binary | inertial
  for (int i = 0; i < 4; ++i) hs[i] = 0.0;
  defer_begin(e);
  if ((rc = launch_landmarks(e, c_huber, e->opt.use_robust_norm_for_proj_residuals))) return rc;
  if (imu_late && (rc = launch_imu_early(e, c_huber))) { (void)defer_flush(e); return rc; }
  // proj_error_ of BuildProblem (BundleAdjuster.cpp:1386): sum of w |r|^2 with the new weights, one
  // partial per linearisation wave
  if ((rc = sum_partials(e, st.O ? st.n_chunks : 0, 1, hs))) { (void)defer_flush(e); return rc; }
  t_l.mark();
  EventTimer t_s(e);
  e->factored = false;
  if ((rc = launch_gather_S(e)) || (rc = launch_posepose_build(e, c_huber, hs + 1))) { (void)defer_flush(e); return rc; }
  // copy the reduced rhs into the rhs row of A
  BAE_HIP(hipMemcpyAsync(e->A.p + (size_t)st.ld * st.ld, e->rhs_sc.p, (size_t)st.n * sizeof(double),
                         hipMemcpyDeviceToDevice, e->stream));
  if (e->sharded()) {
    // S (lower storage) with its rhs row, and the unreduced rhs_p, are sums over the
    // landmark shards: one all-reduce each over xGMI.
    if (dist_solve_enabled(e)) {
      // distributed reduced solve: every rank only needs the sum of the column panels it owns
      // (reduce-scatter); the reduced rhs (one row) and rhs_p are summed everywhere
      BAE_HIP(hipStreamSynchronize(e->stream));
      // the union tile pattern decides which tiles of S travel (a collective on first use)
      if (!e->nzL_valid && (rc = factor_tile_pattern(e))) return rc;
      if (shard_allreduce(e, e->rhs_sc.p, st.ld, 0) != 0) return e->fail_msg("allreduce hook failed");
      if (shard_allreduce(e, e->rhs_p.p, st.ld, 0) != 0) return e->fail_msg("allreduce hook failed");
      if ((rc = dist_reduce_scatter_S(e))) return rc;
      BAE_HIP(hipMemcpyAsync(e->A.p + (size_t)st.ld * st.ld, e->rhs_sc.p, (size_t)st.n * sizeof(double),
                             hipMemcpyDeviceToDevice, e->stream));
    } else {
    // The message is the packed lower triangle + rhs row (half of the square storage).
    const size_t cnt = packed_lower_count(st.ld);
    BAE_HIP(e->packed.alloc(cnt));
    if ((rc = launch_pack_lower(e, 0))) return rc;
    BAE_HIP(hipStreamSynchronize(e->stream));
    if (shard_allreduce(e, e->packed.p, cnt, 0) != 0)
      return e->fail_msg("allreduce hook failed");
    if ((rc = launch_pack_lower(e, 1))) return rc;
    if (shard_allreduce(e, e->rhs_p.p, st.ld, 0) != 0)
      return e->fail_msg("allreduce hook failed");
    BAE_HIP(hipMemcpyAsync(e->rhs_sc.p, e->A.p + (size_t)st.ld * st.ld, (size_t)st.n * sizeof(double),
                           hipMemcpyDeviceToDevice, e->stream));
    }
  }
  t_s.mark();
  if ((rc = defer_flush(e))) return rc;
  e->timers.j_evaluation = t_j.read_ms() + t_l.read_ms();
  e->timers.robust_weights = t_r.read_ms();
  e->timers.jtj_schur = t_s.read_ms();
  errs.proj_error = hs[0]; errs.unary_error = hs[1]; errs.binary_error = hs[2]; errs.inertial_error = hs[3];
  if (out) *out = errs;
  return 0;
}

int ba_hip_solve_gn(ba_hip_engine* h) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  const Structure& st = e->st;
  int status = 0, rc;
  EventTimer t(e);
  // CalculateGn runs only with active poses (BundleAdjuster.cpp:959-964, 1089-1094): with none, the
  // calibration unknowns are not solved for either (delta_k stays empty in the reference)
  const bool skip = st.K && st.Pact == 0;
  if (skip) BAE_HIP(hipMemsetAsync(e->gn_p.p, 0, e->gn_p.bytes(), e->stream));
  if (st.n > 0 && !skip) {
    if (e->opt.keep_reduced_system) {
      BAE_HIP(e->A_keep.alloc((size_t)st.ld * st.ld));
      BAE_HIP(hipMemcpyAsync(e->A_keep.p, e->A.p, (size_t)st.ld * st.ld * sizeof(double),
                             hipMemcpyDeviceToDevice, e->stream));
    }
    if (!e->nzL_valid && (rc = factor_tile_pattern(e))) return rc;
    if (dist_solve_enabled(e)) {
      if ((rc = cholesky_solve_dist(e, e->A.p, st.ld, e->gn_p.p, &status, e->nzL.p))) return rc;
    } else if ((rc = cholesky_solve(e, e->A.p, st.n, st.ld, e->gn_p.p, &status, e->nzL.p))) return rc;
    e->factored = true;
  }
  e->timers.solve = t.stop_ms();
  EventTimer tb(e);
  if ((rc = launch_backsub(e))) return rc;
  e->timers.back_substitution = tb.stop_ms();
  return status ? BA_HIP_FACTORIZATION_ERROR : 0;
}

int ba_hip_dogleg_terms(ba_hip_engine* h, int gn_available, ba_hip_dogleg_scalars* out) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  // every sum of this call stays on the device until ONE copy at the end (defer_flush)
  memset(out, 0, sizeof(*out));
  double* dh = e->dog_h;
  // the steepest-descent denominator |J rhs|^2 does not involve the Gauss-Newton step: the second call of
  // an iteration (gn_available, BundleAdjuster.cpp:971-1002) reuses what the first one summed
  const bool reuse = gn_available && e->dog_jrhs_valid;
  defer_begin(e);
  int rc = launch_dogleg(e, gn_available, dh, reuse);
  if (!rc && !reuse) rc = launch_posepose_jrhs(e, dh + 7);
  const int rf = defer_flush(e);
  if (rc || rf) { e->dog_jrhs_valid = false; return rc ? rc : rf; }
  e->dog_jrhs_valid = true;
  out->rhs_p_sq = dh[0]; out->gn_p_sq = dh[1]; out->rhs_gn_p = dh[2];
  out->rhs_l_sq = dh[3]; out->gn_l_sq = dh[4]; out->rhs_gn_l = dh[5];
  out->j_rhs_sq = dh[6] + dh[7];
  if (e->st.K && (rc = launch_calib_dogleg(e, gn_available, out))) return rc;
  return 0;
}

int ba_hip_compose_step(ba_hip_engine* h, double coef_rhs, double coef_gn, ba_hip_step_norms* out) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  double n2[2];
  defer_begin(e);  // both norms in one copy
  int rc = launch_compose_step(e, coef_rhs, coef_gn, n2);
  const int rf = defer_flush(e);
  if (rc || rf) return rc ? rc : rf;
  if (out) { out->step_p_norm = std::sqrt(n2[0]); out->step_l_norm = std::sqrt(n2[1]); }
  return 0;
}

int ba_hip_apply_step(ba_hip_engine* h) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  EventTimer t(e);
  int rc = launch_apply_step(e);
  if (rc) return rc;
  e->cur = 1 - e->cur;
  e->has_snapshot = true;
  e->obs_e_valid[e->cur] = false;   // a new state in this buffer: no evaluation of it yet
  if (e->calib_dim && !e->calib_tvs && e->st.C > 0) {
    // BundleAdjuster.cpp:46-69: params of camera 0 -= delta_k, then every x_s ray is re-derived from
    // the landmark's reference pixel with the new parameters, keeping its length
    double dk[5] = {0, 0, 0, 0, 0};
    BAE_HIP(hipMemcpyAsync(dk, e->step_p.p + e->st.np, e->st.K * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    e->cam_params_prev = e->prob.cam_params;
    e->cam_w_prev = e->prob.cam_w;
    for (int i = 0; i < 4; ++i) e->prob.cam_params[i] -= dk[i];
    if (e->st.K == 5) e->prob.cam_w[0] -= dk[4];
    if ((rc = upload_cameras(e, false))) return rc;
    if ((rc = launch_reset_rays(e))) return rc;
  }
  if (e->calib_tvs && e->st.C > 0) {
    // BundleAdjuster.cpp:72-83: T_vs of camera 0 <- exp_decoupled(T_vs, -delta_k); the step's tail
    // is delta_k.  The pose caches are rebuilt from the new rig below (t_sw.clear(), :114).
    double dk[6];
    BAE_HIP(hipMemcpyAsync(dk, e->step_p.p + e->st.np, sizeof(dk), hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    double* t = e->prob.cam_tvs.data();
    for (int i = 0; i < 3; ++i) t[i] -= dk[i];
    double qe[4], q[4];
    bad::so3_exp(bad::v3(-dk[3], -dk[4], -dk[5]), qe);
    bad::quat_mul(t + 3, qe, q);
    bad::quat_normalize(q);
    for (int i = 0; i < 4; ++i) t[3 + i] = q[i];
    e->tvs_eval_prev = e->tvs_eval;
    e->tvs_eval = e->prob.cam_tvs;
    if ((rc = upload_cameras(e, false))) return rc;
  }
  rc = launch_pose_prep(e);
  if (rc) return rc;
  e->timers.apply_update = t.stop_ms();
  return 0;
}

int ba_hip_rollback(ba_hip_engine* h) {
  ENG(h);
  NEED_FINAL();
  if (!e->has_snapshot) return e->fail_msg("no snapshot to roll back to");
  BAE_HIP(hipSetDevice(e->device));
  e->cur = 1 - e->cur;
  e->has_snapshot = false;
  int rc;
  if (e->calib_tvs) {
    // the reference restores the poses WITH their cached T_sw but not the rig (:1060-1068): the
    // tables go back to the T_vs they were built with, `cam` keeps the rejected update
    e->tvs_eval = e->tvs_eval_prev;
    if ((rc = upload_cameras(e, true))) return rc;
  } else if (e->calib_dim) {
    // the intrinsics ARE restored (params_backup, :1066, :1147); the rays come back with the landmark buffer
    e->prob.cam_params = e->cam_params_prev;
    e->prob.cam_w = e->cam_w_prev;
    if ((rc = upload_cameras(e, false))) return rc;
  }
  rc = launch_pose_prep(e);
  if (rc) return rc;
  BAE_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

int ba_hip_eval_residuals(ba_hip_engine* h, ba_hip_errors* out) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  ba_hip_errors errs = {0, 0, 0, 0};
  EventTimer t(e);
  double* hs = e->eval_h;  // proj | unary | binary | inertial: one copy for the four sums (defer_flush)
  for (int i = 0; i < 4; ++i) hs[i] = 0.0;
  defer_begin(e);
  int rc = launch_residuals(e, 1);
  if (!rc) rc = sum_partials(e, (e->st.O + 255) / 256, 1, hs);
  if (!rc) rc = launch_posepose_eval(e, hs + 1);
  const int rf = defer_flush(e);
  if (rc || rf) return rc ? rc : rf;
  errs.proj_error = hs[0]; errs.unary_error = hs[1]; errs.binary_error = hs[2]; errs.inertial_error = hs[3];
  e->timers.evaluate_residuals = t.stop_ms();
  if (out) *out = errs;
  return 0;
}

int ba_hip_end_solve(ba_hip_engine* h) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  int rc = launch_end_solve(e);
  if (rc) return rc;
  BAE_HIP(hipStreamSynchronize(e->stream));
  return 0;
}

// ---- results --------------------------------------------------------------------------
int ba_hip_get_poses(ba_hip_engine* h, double* t_wp7, double* v_w3, double* b6) {
  ENG(h);
  NEED_FINAL();
  const uint32_t P = e->st.P;
  std::vector<double> s((size_t)P * kPoseState);
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (P) BAE_HIP(hipMemcpy(s.data(), e->pose_state[e->cur].p, s.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (uint32_t p = 0; p < P; ++p) {
    const double* r = &s[(size_t)p * kPoseState];
    if (t_wp7) for (int i = 0; i < 7; ++i) t_wp7[(size_t)p * 7 + i] = r[i];
    if (v_w3) for (int i = 0; i < 3; ++i) v_w3[(size_t)p * 3 + i] = r[7 + i];
    if (b6) for (int i = 0; i < 6; ++i) b6[(size_t)p * 6 + i] = r[10 + i];
  }
  return 0;
}

int ba_hip_get_landmarks(ba_hip_engine* h, double* x_w4) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipStreamSynchronize(e->stream));
  const double* src = e->lm_dim == 1 ? e->lm_xw.p : e->lm_x[e->cur].p;
  if (e->st.L) BAE_HIP(hipMemcpy(x_w4, src, (size_t)e->st.L * 4 * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int ba_hip_get_landmark_flags(ba_hip_engine* h, uint8_t* is_reliable, uint32_t* num_outliers) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (e->st.L && is_reliable)
    BAE_HIP(hipMemcpy(is_reliable, e->lm_reliable[e->cur].p, e->st.L, hipMemcpyDeviceToHost));
  if (e->st.L && num_outliers)
    BAE_HIP(hipMemcpy(num_outliers, e->lm_outliers.p, (size_t)e->st.L * 4, hipMemcpyDeviceToHost));
  return 0;
}

uint32_t ba_hip_num_pose_params(const ba_hip_engine* h) {
  return reinterpret_cast<const Engine*>(h)->st.np;
}
uint32_t ba_hip_num_calib_params(const ba_hip_engine* h) {
  return reinterpret_cast<const Engine*>(h)->st.K;
}
int ba_hip_set_calibration(ba_hip_engine* h, int calib_size, int do_tvs) {
  ENG(h);
  e->dog_jrhs_valid = false;
  if (calib_size != 0 && calib_size != 4 && calib_size != 5)
    return e->fail_msg("CalibSize must be 0, 4 (LinearCamera: fx, fy, u0, v0) or 5 (FovCamera: fx, fy, u0, v0, w)");
  if (calib_size && do_tvs)
    return e->fail_msg("CalibSize > 0 together with DoTvs is not offered: the reference's T_vs block wipes the intrinsics "
                       "columns it shares an entry with (BundleAdjuster.cpp:1775-1783)");
  if ((do_tvs || calib_size) && e->lm_dim != 1)
    return e->fail_msg("calibration columns need LmSize 1 (parallel_algos.h:102-131)");
  e->calib_dim = do_tvs ? 6 : calib_size;
  e->calib_tvs = do_tvs != 0;
  e->finalized = false;
  return 0;
}
int ba_hip_get_calibration_marginals(ba_hip_engine* h, double* cov) {
  ENG(h);
  NEED_FINAL();
  if (!e->st.K) return e->fail_msg("no calibration columns (ba_hip_set_calibration)");
  if (!e->factored) return e->fail_msg("ba_hip_get_calibration_marginals needs the factor of the last ba_hip_solve_gn");
  if (dist_solve_enabled(e)) return e->fail_msg("calibration marginals: not available with the distributed solve");
  BAE_HIP(hipSetDevice(e->device));
  return trailing_marginals(e, e->A.p, e->st.ld, e->st.np, e->st.K, cov);
}
int ba_hip_set_landmark_ref_pixels(ba_hip_engine* h, uint32_t n, const double* z_ref2) {
  ENG(h);
  e->prob.lm_zref.assign(z_ref2, z_ref2 + 2 * (size_t)n);
  e->finalized = false;
  return 0;
}
int ba_hip_get_camera_params(ba_hip_engine* h, uint32_t n, double* params4) {
  ENG(h);
  const std::vector<double>& p = e->prob.cam_params;
  if (!params4 || (size_t)n * 4 != p.size()) return e->fail_msg("ba_hip_get_camera_params: camera count mismatch");
  for (size_t i = 0; i < p.size(); ++i) params4[i] = p[i];
  return 0;
}
int ba_hip_get_cameras(ba_hip_engine* h, uint32_t n, double* t_vs7) {
  ENG(h);
  const std::vector<double>& t = e->prob.cam_tvs;
  if (!t_vs7 || (size_t)n * 7 != t.size()) return e->fail_msg("ba_hip_get_cameras: camera count mismatch");
  for (size_t i = 0; i < t.size(); ++i) t_vs7[i] = t[i];
  return 0;
}
uint32_t ba_hip_num_lm_params(const ba_hip_engine* h) {
  const Engine* e = reinterpret_cast<const Engine*>(h);
  return e->st.Lact * e->lm_dim;
}

int ba_hip_get_S(ba_hip_engine* h, double* s_nxn) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  const uint32_t n = st.n, ld = st.ld, D = e->pose_dim;
  auto block_of = [&](uint32_t r) { return r < st.np ? r / D : st.Pact; };  // the calibration border is one more block
  std::vector<double> a((size_t)n * ld);
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (e->factored && !(e->opt.keep_reduced_system && e->A_keep.p))
    return e->fail_msg("S was factorised in place; set keep_reduced_system to read it after ba_hip_solve_gn");
  const double* src = e->factored ? e->A_keep.p : e->A.p;
  if (n) BAE_HIP(hipMemcpy(a.data(), src, a.size() * sizeof(double), hipMemcpyDeviceToHost));
  // lower storage -> the reference's s_: block (i,j) kept for i <= j only when
  // use_triangular_matrices (SparseBlockMatrixOps.h:236-238), full symmetric otherwise
  for (uint32_t r = 0; r < n; ++r)
    for (uint32_t c = 0; c < n; ++c) {
      const uint32_t bi = block_of(r), bj = block_of(c);
      double v;
      if (bi == bj) v = a[(size_t)r * ld + c];
      else if (bi < bj) v = a[(size_t)c * ld + r];
      else v = e->opt.use_triangular_matrices ? 0.0 : a[(size_t)r * ld + c];
      s_nxn[(size_t)r * n + c] = v;
    }
  return 0;
}

int ba_hip_get_rhs(ba_hip_engine* h, double* rhs_p_sc, double* rhs_p, double* rhs_l) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (rhs_p_sc && st.n) BAE_HIP(hipMemcpy(rhs_p_sc, e->rhs_sc.p, (size_t)st.n * 8, hipMemcpyDeviceToHost));
  if (rhs_p && st.n) BAE_HIP(hipMemcpy(rhs_p, e->rhs_p.p, (size_t)st.n * 8, hipMemcpyDeviceToHost));
  if (rhs_l && st.Lact) {
    std::vector<double> bl((size_t)st.L * e->lm_dim);
    BAE_HIP(hipMemcpy(bl.data(), e->lm_bl.p, bl.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t l = 0; l < st.L; ++l)
      if (st.lm_opt[l] >= 0)
        for (int k = 0; k < e->lm_dim; ++k)
          rhs_l[(size_t)st.lm_opt[l] * e->lm_dim + k] = bl[(size_t)l * e->lm_dim + k];
  }
  return 0;
}

int ba_hip_get_delta_gn(ba_hip_engine* h, double* delta_p, double* delta_l) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (delta_p && st.n) BAE_HIP(hipMemcpy(delta_p, e->gn_p.p, (size_t)st.n * 8, hipMemcpyDeviceToHost));
  if (delta_l && st.Lact) BAE_HIP(hipMemcpy(delta_l, e->gn_l.p, (size_t)st.Lact * e->lm_dim * 8, hipMemcpyDeviceToHost));
  return 0;
}

int ba_hip_get_step(ba_hip_engine* h, double* delta_p, double* delta_l) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (delta_p && st.n) BAE_HIP(hipMemcpy(delta_p, e->step_p.p, (size_t)st.n * 8, hipMemcpyDeviceToHost));
  if (delta_l && st.Lact) BAE_HIP(hipMemcpy(delta_l, e->step_l.p, (size_t)st.Lact * e->lm_dim * 8, hipMemcpyDeviceToHost));
  return 0;
}

int ba_hip_get_proj_weights(ba_hip_engine* h, double* weight) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  std::vector<double> w(st.O);
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (st.O) BAE_HIP(hipMemcpy(w.data(), e->obs_w.p, (size_t)st.O * 8, hipMemcpyDeviceToHost));
  for (uint32_t s = 0; s < st.O; ++s) weight[st.obs_perm[s]] = w[s];
  return 0;
}

int ba_hip_get_proj_residuals(ba_hip_engine* h, double* residual2) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  if (st.O == 0) return 0;
  BAE_HIP(hipSetDevice(e->device));
  DBuf<double> d;
  BAE_HIP(d.alloc((size_t)2 * st.O));
  int rc = launch_residual_vectors(e, d.p);
  if (rc) { d.release(); return rc; }
  std::vector<double> r((size_t)2 * st.O);
  BAE_HIP(hipStreamSynchronize(e->stream));
  BAE_HIP(hipMemcpy(r.data(), d.p, r.size() * sizeof(double), hipMemcpyDeviceToHost));
  d.release();
  for (uint32_t s = 0; s < st.O; ++s) {
    residual2[2 * (size_t)st.obs_perm[s]] = r[2 * (size_t)s];
    residual2[2 * (size_t)st.obs_perm[s] + 1] = r[2 * (size_t)s + 1];
  }
  return 0;
}

int ba_hip_get_proj_jacobians(ba_hip_engine* h, double* j_meas12, double* j_ref12, double* j_lm, double* r2) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  if (st.O == 0) return 0;
  BAE_HIP(hipSetDevice(e->device));
  BAE_HIP(hipStreamSynchronize(e->stream));
  const int LM = e->lm_dim, R = rows_per_obs(LM);
  std::vector<double> rows((size_t)st.O * R * kRow), jl((size_t)st.O * 2 * LM), sc((size_t)2 * st.O);
  BAE_HIP(hipMemcpy(rows.data(), e->frow.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
  BAE_HIP(hipMemcpy(jl.data(), e->obs_jl.p, jl.size() * sizeof(double), hipMemcpyDeviceToHost));
  BAE_HIP(hipMemcpy(sc.data(), e->scal.p, sc.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (uint32_t s = 0; s < st.O; ++s) {
    const size_t a = st.obs_perm[s];
    const double* rr = &rows[(size_t)s * R * kRow];
    if (j_meas12) for (int i = 0; i < 12; ++i) j_meas12[12 * a + i] = rr[i];
    if (j_ref12) for (int i = 0; i < 12; ++i) j_ref12[12 * a + i] = LM == 1 ? rr[12 + i] : 0.0;
    if (j_lm) for (int i = 0; i < 2 * LM; ++i) j_lm[2 * LM * a + i] = jl[(size_t)s * 2 * LM + i];
    if (r2) { r2[2 * a] = sc[2 * (size_t)s]; r2[2 * a + 1] = sc[2 * (size_t)s + 1]; }
  }
  return 0;
}

// 6 doubles per row: with CalibSize 4 / 5 the last two / the last column are zero
int ba_hip_get_calib_jacobians(ba_hip_engine* h, double* j_k12) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  if (!st.K) return e->fail_msg("no calibration columns (ba_hip_set_calibration)");
  if (st.O == 0) return 0;
  BAE_HIP(hipSetDevice(e->device));
  BAE_HIP(hipStreamSynchronize(e->stream));
  std::vector<double> rows((size_t)2 * st.O * kRow);
  BAE_HIP(hipMemcpy(rows.data(), e->crow.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (uint32_t s = 0; s < st.O; ++s)
    for (int i = 0; i < 12; ++i) j_k12[12 * (size_t)st.obs_perm[s] + i] = rows[12 * (size_t)s + i];
  return 0;
}

int ba_hip_get_imu_residuals(ba_hip_engine* h, double* residual15) {
  ENG(h);
  NEED_FINAL();
  const uint32_t ni = e->prob.num_imu;
  if (ni == 0) return 0;
  BAE_HIP(hipSetDevice(e->device));
  DBuf<double> d;
  BAE_HIP(d.alloc((size_t)15 * ni));
  int rc = launch_imu_residual_vectors(e, d.p);
  if (rc) { d.release(); return rc; }
  BAE_HIP(hipStreamSynchronize(e->stream));
  BAE_HIP(hipMemcpy(residual15, d.p, (size_t)15 * ni * sizeof(double), hipMemcpyDeviceToHost));
  d.release();
  return 0;
}

int ba_hip_get_imu_errors(ba_hip_engine* h, double* mahalanobis) {
  ENG(h);
  NEED_FINAL();
  const Problem& pb = e->prob;
  if (pb.num_imu == 0) return 0;
  BAE_HIP(hipSetDevice(e->device));
  BAE_HIP(hipStreamSynchronize(e->stream));
  BAE_HIP(hipMemcpy(mahalanobis, e->pp_err.p + pb.num_unary + pb.num_binary, (size_t)pb.num_imu * sizeof(double),
                    hipMemcpyDeviceToHost));
  return 0;
}

int ba_hip_get_unary_scales(ba_hip_engine* h, double* scale) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (e->prob.num_unary)
    BAE_HIP(hipMemcpy(scale, e->un_scale.p, (size_t)e->prob.num_unary * 8, hipMemcpyDeviceToHost));
  return 0;
}

int ba_hip_check_solve(ba_hip_engine* h, double* residual_norm, double* rhs_norm) {
  ENG(h);
  NEED_FINAL();
  BAE_HIP(hipSetDevice(e->device));
  if (!(e->factored && e->opt.keep_reduced_system && e->A_keep.p))
    return e->fail_msg("ba_hip_check_solve needs keep_reduced_system and a finished ba_hip_solve_gn");
  if (e->sharded()) return e->fail_msg("ba_hip_check_solve: single shard only");
  double o[2];
  int rc = check_solve_residual(e, e->A_keep.p, e->gn_p.p, e->rhs_sc.p, o);
  if (rc) return rc;
  if (residual_norm) *residual_norm = o[0];
  if (rhs_norm) *rhs_norm = o[1];
  return 0;
}

int ba_hip_get_timers(ba_hip_engine* h, ba_hip_timers* t) {
  ENG(h);
  *t = e->timers;
  return 0;
}

int ba_hip_get_structure_stats(ba_hip_engine* h, ba_hip_structure_stats* out) {
  ENG(h);
  NEED_FINAL();
  const Structure& st = e->st;
  memset(out, 0, sizeof(*out));
  out->poses_active = st.Pact; out->landmarks_active = st.Lact; out->observations = st.O;
  out->incidences = st.n_inc; out->factor_rows = st.n_rows;
  out->pair_blocks = st.n_pairs; out->pair_entries = st.n_pair_entries;
  out->tile_refs = st.n_tile_refs; out->pose_entries = st.n_pose_entries; out->linearize_waves = st.n_chunks;
  const uint64_t nt = st.ld / 64;
  out->tiles_lower = nt * (nt + 1) / 2;
  if (!e->nzL_valid && !(e->sharded())) {
    int rc = factor_tile_pattern(e);
    if (rc) return rc;
  }
  if (e->nzL_valid && e->nzL_host.size() == nt * nt && e->nzS_host.size() == nt * nt)
    for (uint64_t i = 0; i < nt; ++i)
      for (uint64_t k = 0; k <= i; ++k) {
        out->tiles_S += e->nzS_host[i * nt + k] ? 1 : 0;
        out->tiles_L += e->nzL_host[i * nt + k] ? 1 : 0;
      }
  if (e->nzL_valid && e->nzL_host.size() == nt * nt) {
    // column k with m_k structurally nonzero tiles below the diagonal: m_k (m_k + 1) / 2 update products, m_k / 2
    // substitution products (a triangular 64x64 solve is half a product), + the rhs row (m_k + 1 products)
    for (uint64_t k = 0; k < nt; ++k) {
      uint64_t m = 0;
      for (uint64_t i = k + 1; i < nt; ++i) m += e->nzL_host[i * nt + k] ? 1 : 0;
      out->factor_tile_products += m * (m + 1) / 2 + (m + 1) / 2 + m + 1;
    }
  }
  return 0;
}

int ba_hip_debug_set(ba_hip_engine* h, int key, int value) {
  ENG(h);
  switch (key) {
    case 1: e->dbg_assemble_variant = value; break;
    case 2: e->dbg_tile_order = value; e->tile_order_version = ~0ull; break;
    case 4: e->dbg_linearize_variant = value; break;
    case 5: e->dbg_host_structure = value; e->finalized = false; break;
    case 6: e->dbg_imu_wave = value; break;
    case 3: e->dbg_all_tiles = value; e->tile_order_version = ~0ull; e->A_cleared = nullptr; break;
    default: return e->fail_msg("ba_hip_debug_set: unknown key");
  }
  return 0;
}

int ba_hip_set_profiling(ba_hip_engine* h, int enable) {
  ENG(h);
  e->prof_collect();
  e->profiling = enable != 0;
  memset(&e->kstats, 0, sizeof(e->kstats));
  return 0;
}
int ba_hip_get_kernel_stats(ba_hip_engine* h, ba_hip_kernel_stats* out) {
  ENG(h);
  (void)hipStreamSynchronize(e->stream);
  e->prof_collect();
  *out = e->kstats;
  return 0;
}

int ba_hip_device_buffer(ba_hip_engine* h, int which, void** dev_ptr, size_t* num_doubles) {
  ENG(h);
  NEED_FINAL();
  if (which == 0) { *dev_ptr = e->A.p; *num_doubles = (size_t)(e->st.ld + 1) * e->st.ld; return 0; }
  if (which == 1) { *dev_ptr = e->scalars_out.p; *num_doubles = e->scalars_out.n; return 0; }
  return e->fail_msg("unknown buffer id");
}

int ba_hip_allreduce_host(ba_hip_engine* h, void* host, size_t count, int dtype) {
  ENG(h);
  if (!e->sharded() || count == 0) return 0;  // single shard: the values are already the totals
  BAE_HIP(hipSetDevice(e->device));
  DBuf<double> d;  // both element types are 8 bytes
  BAE_HIP(d.alloc(count));
  hipError_t err = hipMemcpy(d.p, host, count * 8, hipMemcpyHostToDevice);
  int rc = 0;
  if (err != hipSuccess) rc = e->fail(err, "hipMemcpy");
  if (!rc && shard_allreduce(e, d.p, count, dtype) != 0) rc = e->fail_msg("allreduce hook failed");
  if (!rc && (err = hipMemcpy(host, d.p, count * 8, hipMemcpyDeviceToHost)) != hipSuccess) rc = e->fail(err, "hipMemcpy");
  d.release();
  return rc;
}

int ba_hip_get_comm_stats(ba_hip_engine* h, ba_hip_comm_stats* out) {
  ENG(h);
  if (!out) return e->fail_msg("null output");
  *out = e->cstats;
  return 0;
}

int ba_hip_reset_comm_stats(ba_hip_engine* h) {
  ENG(h);
  memset(&e->cstats, 0, sizeof(e->cstats));
  return 0;
}

int ba_hip_dist_plan_stats(uint32_t nblk, const uint8_t* nz_lower, int nranks, const char* layout, uint32_t kout,
                           ba_hip_dist_plan_stats_t* out) {
  if (!out || nranks < 1 || nblk == 0) return -1;
  const uint32_t G = kout ? kout : choose_kout(nblk);
  OwnMap map;
  std::string name;
  if (!build_own_map((uint32_t)nranks, layout, G, &map, &name)) return -1;
  const DistPlanStats s = dist_plan_stats(build_dist_plan(nblk, map, nz_lower));
  out->factor_bytes = s.factor_bytes;
  out->chain_recv_max = s.chain_recv_max; out->chain_recv_total = s.chain_recv_total;
  out->side_recv_max = s.side_recv_max; out->side_recv_total = s.side_recv_total;
  out->chain_sent_total = s.chain_sent_total; out->side_sent_total = s.side_sent_total;
  out->recv_max = s.recv_max; out->backward_allreduce_bytes = s.backward_allreduce_bytes;
  out->messages_chain = s.messages_chain; out->messages_side = s.messages_side;
  out->panels = s.panels; out->ranks = s.ranks; out->classes = s.classes; out->kout = G;
  return 0;
}

int ba_hip_get_factor_tile_pattern(ba_hip_engine* h, uint32_t nblk, uint8_t* nz_lower) {
  ENG(h);
  NEED_FINAL();
  if (!e->nzL_valid) {
    BAE_HIP(hipSetDevice(e->device));
    const int rc = factor_tile_pattern(e);
    if (rc) return rc;
  }
  if (e->nzL_host.size() != (size_t)nblk * nblk) return e->fail_msg("tile count mismatch");
  memcpy(nz_lower, e->nzL_host.data(), e->nzL_host.size());
  return 0;
}

int ba_hip_solve_is_distributed(ba_hip_engine* h) {
  if (!h) return 0;
  return dist_solve_enabled(reinterpret_cast<Engine*>(h)) ? 1 : 0;
}

int ba_hip_set_collectives(ba_hip_engine* h, ba_hip_collective_fn fn, void* ctx) {
  ENG(h);
  e->coll = fn; e->coll_ctx = ctx;
  e->dist_plan_version = ~0ull;
  return 0;
}

int ba_hip_set_allreduce(ba_hip_engine* h, ba_hip_allreduce_fn fn, void* ctx, int rank, int nranks) {
  ENG(h);
  e->allreduce = fn; e->allreduce_ctx = ctx; e->rank = rank; e->nranks = nranks < 1 ? 1 : nranks;
  e->nzL_valid = false;  // the tile pattern of S is the union over the shards
  e->dist_plan_version = ~0ull;
  e->dog_jrhs_valid = false;  // a local sum may have become a cross-shard one
  return 0;
}

int ba_hip_dense_solve(ba_hip_engine* h, uint32_t n, const double* a_lower, const double* b, double* x) {
  ENG(h);
  BAE_HIP(hipSetDevice(e->device));
  const uint32_t ld = std::max(((n + 63) / 64) * 64, 64u);
  std::vector<double> A((size_t)(ld + 1) * ld, 0.0);
  for (uint32_t r = 0; r < n; ++r)
    for (uint32_t c = 0; c <= r; ++c) A[(size_t)r * ld + c] = a_lower[(size_t)r * n + c];
  for (uint32_t r = n; r < ld; ++r) A[(size_t)r * ld + r] = 1.0;
  for (uint32_t c = 0; c < n; ++c) A[(size_t)ld * ld + c] = b[c];
  DBuf<double> dA, dx;
  BAE_HIP(dA.alloc(A.size()));
  BAE_HIP(dx.alloc(ld));
  BAE_HIP(e->flags.alloc(16));
  BAE_HIP(hipMemcpy(dA.p, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  int status = 0;
  int rc = cholesky_solve(e, dA.p, n, ld, dx.p, &status, nullptr);  // arbitrary matrix: dense
  if (rc == 0) {
    std::vector<double> xx(ld);
    hipError_t err = hipMemcpy(xx.data(), dx.p, (size_t)ld * 8, hipMemcpyDeviceToHost);
    if (err != hipSuccess) rc = e->fail(err, "hipMemcpy");
    for (uint32_t i = 0; i < n; ++i) x[i] = xx[i];
  }
  dA.release(); dx.release();
  if (rc) return rc;
  return status ? BA_HIP_FACTORIZATION_ERROR : 0;
}

int ba_hip_select_kth(ba_hip_engine* h, uint32_t n, const double* values, uint32_t k, double* out) {
  ENG(h);
  BAE_HIP(hipSetDevice(e->device));
  DBuf<double> dv;
  BAE_HIP(dv.alloc(std::max<uint32_t>(n, 1)));
  BAE_HIP(e->hist.alloc(2048 + 8));
  if (n) BAE_HIP(hipMemcpy(dv.p, values, (size_t)n * 8, hipMemcpyHostToDevice));
  int rc = select_kth(e, dv.p, n, k, out);
  dv.release();
  return rc;
}

}  // extern "C"
