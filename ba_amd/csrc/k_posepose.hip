// gfx950 kernels of the pose-pose residuals: unary prior (k_unary), binary pose-pose
// (k_binary), IMU pre-integration (k_imu) — one thread per residual, counts are O(poses) —
// and the deterministic scatter of their D x D Hessian blocks and gradients into the
// reduced system (k_pp_scatter: one workgroup per active pose owns block-row j of S).
//
// Reference code replaced: BundleAdjuster.cpp:1392-1541 (residuals, Huber weighting),
// :357-401 (J^T J and J^T r of the three residual types), :1647-1726 (insertion with
// column masks), :190-256 (EvaluateResiduals), :889-910 (dogleg J*rhs terms).
#include "engine.h"
#include "dpose.h"

using namespace bad;

namespace bae {

static const int kPPH = 225;  // one 15 x 15 block

// slot layout of a residual in the pose-pose arrays: [unary | binary | imu]
__device__ __forceinline__ void store_blocks(const PPBlocks& b, double* __restrict__ pp_h,
                                             double* __restrict__ pp_g, uint32_t slot) {
  double* h = pp_h + (size_t)slot * 3 * kPPH;
  for (int i = 0; i < kPPH; ++i) { h[i] = b.h11.m[i]; h[kPPH + i] = b.h12.m[i]; h[2 * kPPH + i] = b.h22.m[i]; }
  double* g = pp_g + (size_t)slot * 30;
  for (int i = 0; i < 15; ++i) { g[i] = b.g1[i]; g[15 + i] = b.g2[i]; }
}
template <int N>
__device__ __forceinline__ void store_dz(const DM<N, N>& a, double* __restrict__ dst) {
  for (int r = 0; r < 15; ++r)
    for (int c = 0; c < 15; ++c) dst[r * 15 + c] = (r < N && c < N) ? a(r, c) : 0.0;
}

// ---- unary ---------------------------------------------------------------------------
// mode 0: residual + Jacobian to scratch, mahalanobis (with the current scale) to err[]
// mode 1: Huber weight from c_huber, scale *= weight (BundleAdjuster.cpp:1463-1469),
//         blocks, error
// mode 2: EvaluateResiduals (:190-205)
__global__ void k_unary(int n, int mode, double c_huber, const uint32_t* __restrict__ pose,
                        const double* __restrict__ prior7, const double* __restrict__ cov_inv,
                        const uint8_t* __restrict__ use_rot, double* __restrict__ scale,
                        const double* __restrict__ state, double* __restrict__ err,
                        double* __restrict__ pp_h, double* __restrict__ pp_g,
                        double* __restrict__ pp_dz, double* __restrict__ pp_info,
                        uint32_t slot0, double* __restrict__ out_err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Tq t_wp = tq_from7(state + (size_t)pose[i] * kPoseState);
  const Tq t_pr = tq_from7(prior7 + (size_t)i * 7);
  double r[6];
  DM<6, 6> J;
  unary_residual(t_wp, t_pr, use_rot[i], r, &J);
  const double* ci = cov_inv + (size_t)i * 36;
  double sc = scale[i];
  const double md = quad6(ci, r) * sc;
  if (mode == 0) { err[i] = md; return; }
  if (mode == 2) { out_err[i] = md; return; }
  const double e = sqrt(md);
  const double w = (e > c_huber) ? c_huber / e : 1.0;
  sc *= w;
  scale[i] = sc;
  PPBlocks b;
  unary_blocks(r, J, ci, sc, &b);
  store_blocks(b, pp_h, pp_g, slot0 + i);
  double* dz = pp_dz + (size_t)(slot0 + i) * 2 * kPPH;
  store_dz(J, dz);
  for (int k = 0; k < kPPH; ++k) dz[kPPH + k] = 0.0;
  double* info = pp_info + (size_t)(slot0 + i) * kPPH;
  for (int rr = 0; rr < 15; ++rr)
    for (int c = 0; c < 15; ++c) info[rr * 15 + c] = (rr < 6 && c < 6) ? ci[rr * 6 + c] * sc : 0.0;
  out_err[i] = b.err_build;
}

// ---- binary --------------------------------------------------------------------------
__global__ void k_binary(int n, int mode, const uint32_t* __restrict__ p1,
                         const uint32_t* __restrict__ p2, const double* __restrict__ t12,
                         const double* __restrict__ cov_inv, const double* __restrict__ cov_inv_sqrt,
                         const double* __restrict__ weight, const uint8_t* __restrict__ use_rot,
                         const double* __restrict__ state, double* __restrict__ pp_h,
                         double* __restrict__ pp_g, double* __restrict__ pp_dz,
                         double* __restrict__ pp_info, uint32_t slot0, double* __restrict__ out_err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Tq t_w1 = tq_from7(state + (size_t)p1[i] * kPoseState);
  const Tq t_w2 = tq_from7(state + (size_t)p2[i] * kPoseState);
  const Tq t = tq_from7(t12 + (size_t)i * 7);
  PPBlocks b;
  double ev;
  binary_blocks(t_w1, t_w2, t, cov_inv + (size_t)i * 36, cov_inv_sqrt + (size_t)i * 36, weight[i],
                use_rot[i], &b, &ev);
  if (mode == 2) { out_err[i] = ev; return; }
  store_blocks(b, pp_h, pp_g, slot0 + i);
  // dogleg: ||S^1/2 (dz1 g1 + dz2 g2)||^2 = v^T cov_inv v with the UNWEIGHTED J (:1662-1664)
  {
    const Tq t_1w = tq_inv(t_w1);
    const Tq tt = tq_mul(t_1w, t_w2);
    const DM<6, 7> dl = dLog_decoupled_dt1(tt, t);
    DM<6, 6> dz1 = mm(mm(dl, dt1_t2_dt1(t_1w, t_w2)), dinv_exp_decoupled_dx(t_w1));
    DM<6, 6> dz2 = mm(mm(dl, dt1_t2_dt2(t_1w)), dexp_decoupled_dx(t_w2));
    if (!use_rot[i])
      for (int r = 3; r < 6; ++r)
        for (int c = 0; c < 6; ++c) { dz1(r, c) = 0.0; dz2(r, c) = 0.0; }
    double* dz = pp_dz + (size_t)(slot0 + i) * 2 * kPPH;
    store_dz(dz1, dz);
    store_dz(dz2, dz + kPPH);
    double* info = pp_info + (size_t)(slot0 + i) * kPPH;
    const double* ci = cov_inv + (size_t)i * 36;
    for (int rr = 0; rr < 15; ++rr)
      for (int c = 0; c < 15; ++c) info[rr * 15 + c] = (rr < 6 && c < 6) ? ci[rr * 6 + c] : 0.0;
  }
  out_err[i] = b.err_build;
}

// ---- inertial ---------------------------------------------------------------------------
// mode 1: residual, Jacobians, covariance, Huber factor (optional, driven by the
//         PROJECTION c_huber as the reference does, BundleAdjuster.cpp:1497-1521), blocks
// mode 2: EvaluateResiduals (:225-256): re-integration without Jacobians, stored cov_inv
__global__ void __launch_bounds__(64)
k_imu(int n, int mode, int RS, int use_robust, double c_huber, const uint32_t* __restrict__ p1,
      const uint32_t* __restrict__ p2, const uint32_t* __restrict__ mptr,
      const double* __restrict__ meas, const double* __restrict__ grav,
      const double* __restrict__ noise /* r6 | rb6 */, const uint8_t* __restrict__ pose_active,
      const double* __restrict__ state, double* __restrict__ cov_inv_store,
      double* __restrict__ pp_h, double* __restrict__ pp_g, double* __restrict__ pp_dz,
      double* __restrict__ pp_info, uint32_t slot0, double* __restrict__ out_err,
      double* __restrict__ res_out /* mode 2, optional: the residual vectors, 15 per residual */,
      int cov_once, double* __restrict__ frozen, uint8_t* __restrict__ cov_done,
      const double* __restrict__ steps /* k_imu_steps, or null: integrate with Jacobians here */) {
  __shared__ double wave_lds[1280];  // products: both operands + output (<= 3 x 225); the inverse: 4 x 225; accumulation: 420; blocks: 5 x 225; step chain: 986
  if (mode == 3) {
    // Step Jacobians of the pre-integration, one lane per IMU sample (dpose.h: imu_step_jacobians);
    // n = number of samples, RS = number of residuals; `mptr` is the CSR of the samples over the
    // residuals (the residual of sample j by binary search).  A mode of this kernel rather than a kernel
    // of its own: two kernels with different private-memory sizes alternating on one queue make the
    // runtime re-size the queue's scratch at every launch.
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (uint32_t)n) return;
    uint32_t lo = 0, hi = (uint32_t)RS;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (mptr[mid] <= j) lo = mid; else hi = mid;
    }
    const uint32_t ri = lo, k = j - mptr[ri];
    if (k == 0 || j >= mptr[ri + 1]) return;          // sample 0 starts the integration: no step ends there
    if (cov_once && cov_done[ri]) return;             // frozen covariance / bias Jacobian: not needed
    const double gg[3] = {grav[0], grav[1], grav[2]};
    imu_step_jacobians(state + (size_t)p1[ri] * kPoseState, meas + (size_t)mptr[ri] * 7, (int)k, gg,
                       const_cast<double*>(steps) + (size_t)j * 160);
    return;
  }
  // mode 4 = mode 1 with ONE WAVEFRONT PER RESIDUAL: every lane runs the scalar code of the residual
  // (uniform loads), the dense 10x10 / 15x15 products are dealt to the lanes through LDS (dpose.h:
  // WaveCtx; bitwise the scalar results), the outputs are stored lane-strided.  For moderate residual
  // counts: the private memory of a dispatch grows with its wavefronts.
  const bool wave = mode == 4;
  const WaveCtx wctx = {wave_lds, (int)threadIdx.x};
  const WaveCtx* wc = wave ? &wctx : nullptr;
  if (wave) mode = 1;
  const int lane0 = wave ? (int)threadIdx.x : 0, lstep = wave ? 64 : 1;
  const int i = wave ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const double* s1 = state + (size_t)p1[i] * kPoseState;
  const double* s2 = state + (size_t)p2[i] * kPoseState;
  const double* m = meas + (size_t)mptr[i] * 7;
  const int nm = (int)(mptr[i + 1] - mptr[i]);
  const double g[3] = {grav[0], grav[1], grav[2]};
  ImuOut io;
  double* cst = cov_inv_store + (size_t)i * kPPH;
  if (mode == 2) {
    imu_residual(s1, s2, m, nm, g, noise, noise + 6, RS, false, &io);
    if (res_out)
      for (int r = 0; r < 15; ++r) res_out[(size_t)i * 15 + r] = io.r[r];
    double e = 0.0;
    for (int r = 0; r < 15; ++r) {
      double q = 0.0;
      for (int c = 0; c < 15; ++c) q += cst[r * 15 + c] * io.r[c];
      e += io.r[r] * q;
    }
    out_err[i] = e;
    return;
  }
  if (cov_once) {
    // calculate_inertial_covariance_once: the first linearisation of a residual freezes its
    // integration covariance and bias Jacobian
    double* fz = frozen + (size_t)i * 160;
    const double* st = steps ? steps + (size_t)mptr[i] * 160 : nullptr;
    const bool done = cov_done[i] != 0;
    if (wave) __syncthreads();  // every lane has read the flag before lane 0 sets it
    if (done) imu_residual(s1, s2, m, nm, g, noise, noise + 6, RS, true, &io, fz, nullptr, nullptr, wc);
    else imu_residual(s1, s2, m, nm, g, noise, noise + 6, RS, true, &io, nullptr, fz, st, wc);
    cov_done[i] = 1;
  } else {
    imu_residual(s1, s2, m, nm, g, noise, noise + 6, RS, true, &io, nullptr, nullptr,
                 steps ? steps + (size_t)mptr[i] * 160 : nullptr, wc);
  }
  double w = 1.0;
  if (use_robust) {
    double md = 0.0;
    for (int r = 0; r < 15; ++r) {
      double q = 0.0;
      for (int c = 0; c < 15; ++c) q += io.cov_inv(r, c) * io.r[c];
      md += io.r[r] * q;
    }
    const double e = sqrt(md);
    const bool is_cond = !pose_active[p1[i]] && pose_active[p2[i]];
    if (e > c_huber && !is_cond) w = c_huber / e;
  }
  double err_build;
  {
    double* h = pp_h + (size_t)(slot0 + i) * 3 * kPPH;
    double* gg = pp_g + (size_t)(slot0 + i) * 30;
    if (wave) {
      err_build = imu_blocks_wave(io, w, wc, h, gg);
    } else {
      PPBlocks b;
      imu_blocks(io, w, &b, nullptr);
      for (int k = 0; k < kPPH; ++k) { h[k] = b.h11.m[k]; h[kPPH + k] = b.h12.m[k]; h[2 * kPPH + k] = b.h22.m[k]; }
      for (int k = 0; k < 15; ++k) { gg[k] = b.g1[k]; gg[15 + k] = b.g2[k]; }
      err_build = b.err_build;
    }
  }
  double* dz = pp_dz + (size_t)(slot0 + i) * 2 * kPPH;
  double* info = pp_info + (size_t)(slot0 + i) * kPPH;
  for (int k = lane0; k < kPPH; k += lstep) {
    dz[k] = io.dz1.m[k];
    dz[kPPH + k] = io.dz2.m[k];
    info[k] = io.cov_inv.m[k] * w;
    cst[k] = io.cov_inv.m[k] * w;  // res.cov_inv = res.cov_inv * weight (:1526)
  }
  if (lane0 == 0) out_err[i] = err_build;
}

// ---- scatter into the reduced system --------------------------------------------------------
// One workgroup per active pose j (block-row j of the lower storage).  Entries of pose j:
// (slot, side, other pose opt id or -1) in residual order; every thread owns one element
// of the D x D block, so the adds happen in a fixed order (deterministic, no atomics).
// Masked parameters: rows/columns skipped (their Jacobian columns are zero in the
// reference, BundleAdjuster.cpp:1653-1660,1675-1682,1698-1705).
__global__ void __launch_bounds__(256)
k_pp_scatter(int D, uint32_t ld, uint32_t n_pad, const uint32_t* __restrict__ ptr,
             const uint4* __restrict__ ent, const uint16_t* __restrict__ mask_opt,
             const double* __restrict__ pp_h, const double* __restrict__ pp_g,
             double* __restrict__ A, double* __restrict__ rhs_p, double* __restrict__ rhs_sc) {
  const uint32_t j = blockIdx.x;
  const int t = threadIdx.x;
  const int r = t / D, c = t - r * D;
  const bool elem = t < D * D;
  const uint16_t mj = mask_opt[j];
  for (uint32_t e = ptr[j]; e < ptr[j + 1]; ++e) {
    const uint4 en = ent[e];  // x = slot, y = side, z = other opt id (0xffffffff: none)
    const double* h = pp_h + (size_t)en.x * 3 * kPPH;
    const double* g = pp_g + (size_t)en.x * 30;
    if (elem && !(mj & (1u << r)) && !(mj & (1u << c))) {
      const double v = (en.y == 0 ? h : h + 2 * kPPH)[r * 15 + c];
      A[((size_t)j * D + r) * ld + (size_t)j * D + c] += v;
    }
    if (en.z != 0xffffffffu && en.z < j && elem) {
      const uint16_t mo = mask_opt[en.z];
      if (!(mj & (1u << r)) && !(mo & (1u << c))) {
        // block (row pose j, col pose other): side 0 -> H12[r][c]; side 1 -> H12^T = H12[c][r]
        const double v = en.y == 0 ? h[kPPH + r * 15 + c] : h[kPPH + c * 15 + r];
        A[((size_t)j * D + r) * ld + (size_t)en.z * D + c] += v;
      }
    }
    if (t < D && !(mj & (1u << t))) {
      const double gv = g[en.y * 15 + t];
      rhs_p[(size_t)j * D + t] += gv;
      rhs_sc[(size_t)j * D + t] += gv;
    }
  }
  (void)n_pad;
}

// dogleg: sum_res v^T info v,  v = dz1 g_p1 + dz2 g_p2 with masked columns and inactive
// poses dropped (BundleAdjuster.cpp:889-910)
__global__ void k_pp_jrhs(int nres, int D, const uint32_t* __restrict__ res_p1,
                          const uint32_t* __restrict__ res_p2, const int32_t* __restrict__ pose_opt,
                          const uint16_t* __restrict__ pose_mask, const double* __restrict__ pp_dz,
                          const double* __restrict__ pp_info, const double* __restrict__ rhs_p,
                          double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nres) return;
  double v[15];
  for (int r = 0; r < 15; ++r) v[r] = 0.0;
  const double* dz = pp_dz + (size_t)i * 2 * kPPH;
  for (int side = 0; side < 2; ++side) {
    const uint32_t p = side == 0 ? res_p1[i] : res_p2[i];
    if (p == 0xffffffffu) continue;
    const int po = pose_opt[p];
    if (po < 0) continue;
    const uint16_t m = pose_mask[p];
    const double* gp = rhs_p + (size_t)po * D;
    for (int c = 0; c < D; ++c) {
      if (m & (1u << c)) continue;
      const double gc = gp[c];
      for (int r = 0; r < 15; ++r) v[r] += dz[side * kPPH + r * 15 + c] * gc;
    }
  }
  const double* info = pp_info + (size_t)i * kPPH;
  double s = 0.0;
  for (int r = 0; r < 15; ++r) {
    double q = 0.0;
    for (int c = 0; c < 15; ++c) q += info[r * 15 + c] * v[c];
    s += v[r] * q;
  }
  out[i] = s;
}

// fixed-order sum of a small array (one thread; counts are O(poses))
__global__ void k_sum_small(int n, const double* __restrict__ v, double* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += v[i];
    *out = s;
  }
}

static int sum_small(Engine* e, int n, const double* d_v, double* host) {
  *host = 0.0;
  if (e->defer_active && n > 0 && e->defer_n < 40) {  // see Engine::defer_active
    hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(1), 0, e->stream, n, d_v, e->scalars_out.p + 16 + e->defer_n);
    BAE_HIP(hipGetLastError());
    e->defer_host[e->defer_n++] = host;
    return 0;
  }
  if (n > 0) {
    hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(1), 0, e->stream, n, d_v, e->scalars_out.p + 8);
    BAE_HIP(hipGetLastError());
    BAE_HIP(hipMemcpyAsync(host, e->scalars_out.p + 8, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
  }
  if (e->sharded()) {
    BAE_HIP(hipMemcpy(e->scalars_out.p + 8, host, sizeof(double), hipMemcpyHostToDevice));
    if (shard_allreduce(e, e->scalars_out.p + 8, 1, 0) != 0) return e->fail_msg("allreduce hook failed");
    BAE_HIP(hipMemcpy(host, e->scalars_out.p + 8, sizeof(double), hipMemcpyDeviceToHost));
  }
  return 0;
}

// The step pass with ONE WAVEFRONT PER SAMPLE and the RK4 Jacobian chain resident in LDS (dpose.h:
// imu_step_jacobians_wave) — for sliding windows, where a few hundred samples cannot fill the device and the
// lane-per-sample form (k_imu mode 3) is one long private-memory chain per lane: 179 -> ~50 us on a 30-pose
// window.  Every lane repeats the scalar part (state prefix, closed-form blocks), so beyond a few thousand
// samples the lane-per-sample form is the faster one again (configs[2], 50k samples: 3.1 ms against 7.7 ms for
// both passes).  A kernel of its own: its private frame is a tenth of k_imu's.
__global__ void __launch_bounds__(64)
k_imu_steps_wave(int n, int RS, const uint32_t* __restrict__ p1, const uint32_t* __restrict__ mptr,
                 const double* __restrict__ meas, const double* __restrict__ grav, const double* __restrict__ state,
                 int cov_once, const uint8_t* __restrict__ cov_done, double* __restrict__ steps) {
  __shared__ double lds[1024];
  const uint32_t j = blockIdx.x;
  if (j >= (uint32_t)n) return;
  uint32_t lo = 0, hi = (uint32_t)RS;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (mptr[mid] <= j) lo = mid; else hi = mid;
  }
  const uint32_t ri = lo, k = j - mptr[ri];
  if (k == 0 || j >= mptr[ri + 1]) return;
  if (cov_once && cov_done[ri]) return;
  const double gg[3] = {grav[0], grav[1], grav[2]};
  const WaveCtx ws = {lds, (int)threadIdx.x};
  imu_step_jacobians_wave(state + (size_t)p1[ri] * kPoseState, meas + (size_t)mptr[ri] * 7, (int)k, gg,
                          steps + (size_t)j * 160, &ws);
}

// The inertial residuals of BuildProblem (k_imu: one lane per residual, RK4 + 15x15 algebra — a few
// milliseconds of latency-bound work on ~80 wavefronts) depend on the state and on the projection
// Huber constant only, not on the projection linearisation: they run on the engine's second stream
// concurrently with k_linearize / k_assemble_tiles and are joined by launch_posepose_build.
int launch_imu_early(Engine* e, double c_huber_proj) {
  const Problem& pb = e->prob;
  const uint32_t nu = pb.num_unary, nb = pb.num_binary, ni = pb.num_imu;
  if (ni == 0) return 0;
  const double* state = e->pose_state[e->cur].p;
  // BA_HIP_IMU_SERIAL=1 (A/B switch): the inertial kernels on the main stream, ahead of the projection linearisation
  static const bool serial_env = getenv("BA_HIP_IMU_SERIAL") != nullptr;
  hipStream_t s2 = serial_env ? e->stream : e->stream2;
  if (!e->ev_imu_done) {
    BAE_HIP(hipEventCreateWithFlags(&e->ev_imu_done, hipEventDisableTiming));
    BAE_HIP(hipEventCreateWithFlags(&e->ev_imu_start, hipEventDisableTiming));
  }
  // everything the main stream has queued so far (state, uploads) comes first
  BAE_HIP(hipEventRecord(e->ev_imu_start, e->stream));
  BAE_HIP(hipStreamWaitEvent(s2, e->ev_imu_start, 0));
  if (ni && e->imu_cov_once) {
    // frozen integration covariances survive re-uploads of the same (append-only) residual list
    if (ni < e->imu_cov_count) e->imu_cov_count = 0;  // the list was rebuilt: forget everything
    if (e->imu_cov_done.n < ni) {
      DBuf<double> fz; DBuf<uint8_t> dn;
      BAE_HIP(fz.alloc((size_t)ni * 160));
      BAE_HIP(dn.alloc(ni));
      BAE_HIP(hipMemsetAsync(dn.p, 0, ni, s2));
      if (e->imu_cov_count) {
        BAE_HIP(hipMemcpyAsync(fz.p, e->imu_frozen.p, (size_t)e->imu_cov_count * 160 * sizeof(double),
                               hipMemcpyDeviceToDevice, s2));
        BAE_HIP(hipMemcpyAsync(dn.p, e->imu_cov_done.p, e->imu_cov_count, hipMemcpyDeviceToDevice, s2));
      }
      BAE_HIP(hipStreamSynchronize(s2));
      e->imu_frozen.release(); e->imu_cov_done.release();
      e->imu_frozen = fz; e->imu_cov_done = dn;
    } else if (e->imu_cov_count == 0) {
      BAE_HIP(hipMemsetAsync(e->imu_cov_done.p, 0, e->imu_cov_done.n, s2));
    }
    e->imu_cov_count = ni;
  }
  if (ni) {
    // two launches: the step Jacobians of every sample in parallel, then one lane per residual for the
    // sequential part (states, accumulations, the 15x15 algebra).  BA_HIP_IMU_FUSED=1: the single-launch form
    static const int variant_env = getenv("BA_HIP_IMU_VARIANT") ? atoi(getenv("BA_HIP_IMU_VARIANT")) : -1;  // experiments: as debug key 6
    if (variant_env >= 0 && e->dbg_imu_wave < 0) e->dbg_imu_wave = variant_env;
    static const bool fused_env = getenv("BA_HIP_IMU_FUSED") != nullptr;
    const bool fused = fused_env || e->dbg_imu_wave == 2;  // (key 6 = 2: the single-pass form)
    const uint32_t n_meas = (uint32_t)(pb.imu_meas.size() / 7);
    const double* steps = nullptr;
    e->prof_begin(e->ev_imu, s2);
    if (!fused && n_meas) {
      BAE_HIP(e->imu_steps.alloc((size_t)n_meas * 160));
      steps = e->imu_steps.p;
      // a wavefront per sample (k_imu_steps_wave) while the samples are few, one lane per sample (mode 3) beyond
      // (ba_hip_debug_set key 6: 4 forces the former, 0 / 1 the latter)
      static const uint32_t steps_wave_max = getenv("BA_HIP_IMU_STEPS_WAVE_MAX") ? (uint32_t)atoi(getenv("BA_HIP_IMU_STEPS_WAVE_MAX")) : 2048u;
      const bool wave_s = e->dbg_imu_wave < 0 ? n_meas <= steps_wave_max : e->dbg_imu_wave == 4;
      if (wave_s)
        hipLaunchKernelGGL(k_imu_steps_wave, dim3(n_meas), dim3(64), 0, s2, (int)n_meas, (int)ni, e->imu_p1.p, e->imu_ptr.p,
                           e->imu_meas.p, e->imu_consts.p, state, e->imu_cov_once ? 1 : 0, e->imu_cov_done.p,
                           e->imu_steps.p);
      else
        hipLaunchKernelGGL(k_imu, dim3((n_meas + 63) / 64), dim3(64), 0, s2, (int)n_meas, 3, (int)ni, 0, 0.0, e->imu_p1.p,
                         e->imu_p2.p, e->imu_ptr.p, e->imu_meas.p, e->imu_consts.p, e->imu_consts.p + 3,
                         e->pose_active.p, state, e->imu_cov_inv.p, e->pp_h.p, e->pp_g.p, e->pp_dz.p,
                         e->pp_info.p, nu + nb, e->pp_err.p + nu + nb, (double*)nullptr, e->imu_cov_once ? 1 : 0,
                         e->imu_frozen.p, e->imu_cov_done.p, steps);
    }
    // a wavefront per residual (mode 4) up to 16k residuals; one lane per residual beyond
    // (ba_hip_debug_set key 6: 0 = one lane per residual, 1 = a wavefront per residual, -1 = by count)
    static const uint32_t wave_max = getenv("BA_HIP_IMU_WAVE_MAX") ? (uint32_t)atoi(getenv("BA_HIP_IMU_WAVE_MAX")) : 16384u;
    const bool wave = steps && (e->dbg_imu_wave < 0 ? ni <= wave_max : e->dbg_imu_wave != 0);  // (2: fused, no steps)
    hipLaunchKernelGGL(k_imu, dim3(wave ? ni : (ni + 63) / 64), dim3(64), 0, s2, (int)ni, wave ? 4 : 1, e->pose_dim,
                       e->opt.use_robust_norm_for_inertial_residuals, c_huber_proj, e->imu_p1.p,
                       e->imu_p2.p, e->imu_ptr.p, e->imu_meas.p, e->imu_consts.p, e->imu_consts.p + 3,
                       e->pose_active.p, state, e->imu_cov_inv.p, e->pp_h.p, e->pp_g.p, e->pp_dz.p,
                       e->pp_info.p, nu + nb, e->pp_err.p + nu + nb, (double*)nullptr, e->imu_cov_once ? 1 : 0,
                       e->imu_frozen.p, e->imu_cov_done.p, steps);
    e->prof_end(e->ev_imu, s2);
    BAE_HIP(hipGetLastError());
  }
  BAE_HIP(hipEventRecord(e->ev_imu_done, s2));
  e->imu_early_pending = true;
  return 0;
}

// BuildProblem part of the pose-pose residuals + scatter (after launch_gather_S).
int launch_posepose_build(Engine* e, double c_huber_proj, double* h3) {  // h3: unary | binary | inertial error sums
  const Problem& pb = e->prob;
  const uint32_t nu = pb.num_unary, nb = pb.num_binary, ni = pb.num_imu;
  if (nu + nb + ni == 0 && !(e->sharded())) return 0;
  int rc;
  const double* state = e->pose_state[e->cur].p;
  if (nu) {
    hipLaunchKernelGGL(k_unary, dim3((nu + 63) / 64), dim3(64), 0, e->stream, (int)nu, 0, 0.0,
                       e->un_pose.p, e->un_t.p, e->un_cov_inv.p, e->un_rot.p, e->un_scale.p, state,
                       e->pp_err.p, e->pp_h.p, e->pp_g.p, e->pp_dz.p, e->pp_info.p, 0u, e->pp_err.p);
    BAE_HIP(hipGetLastError());
  }
  // Huber sigma over the unary mahalanobis distances (BundleAdjuster.cpp:1457-1461); the
  // unary residuals live on shard 0, the selection is still a global one
  uint64_t n_un_total = nu;
  if (e->sharded()) {
    double cnt = (double)nu;
    BAE_HIP(hipMemcpy(e->scalars_out.p, &cnt, sizeof(double), hipMemcpyHostToDevice));
    if (shard_allreduce(e, e->scalars_out.p, 1, 0) != 0) return e->fail_msg("allreduce hook failed");
    BAE_HIP(hipMemcpy(&cnt, e->scalars_out.p, sizeof(double), hipMemcpyDeviceToHost));
    n_un_total = (uint64_t)(cnt + 0.5);
  }
  if (n_un_total > 0) {
    double med = 0.0;
    if ((rc = select_kth(e, e->pp_err.p, nu, (uint64_t)std::floor(n_un_total * 0.5), &med))) return rc;
    const double c_huber = 1.2107 * std::sqrt(med);
    if (nu) {
      hipLaunchKernelGGL(k_unary, dim3((nu + 63) / 64), dim3(64), 0, e->stream, (int)nu, 1, c_huber,
                         e->un_pose.p, e->un_t.p, e->un_cov_inv.p, e->un_rot.p, e->un_scale.p, state,
                         e->pp_err.p, e->pp_h.p, e->pp_g.p, e->pp_dz.p, e->pp_info.p, 0u, e->pp_err.p);
      BAE_HIP(hipGetLastError());
    }
  }
  if ((rc = sum_small(e, nu, e->pp_err.p, h3))) return rc;
  if (nb) {
    hipLaunchKernelGGL(k_binary, dim3((nb + 63) / 64), dim3(64), 0, e->stream, (int)nb, 1, e->bin_p1.p,
                       e->bin_p2.p, e->bin_t.p, e->bin_cov_inv.p, e->bin_cov_inv_sqrt.p, e->bin_w.p,
                       e->bin_rot.p, state, e->pp_h.p, e->pp_g.p, e->pp_dz.p, e->pp_info.p, nu,
                       e->pp_err.p + nu);
    BAE_HIP(hipGetLastError());
  }
  if ((rc = sum_small(e, nb, e->pp_err.p + nu, h3 + 1))) return rc;
  // k_imu was started on the second stream right after the Huber constant was known
  // (launch_imu_early): join it here
  if (ni && e->imu_early_pending) {
    BAE_HIP(hipStreamWaitEvent(e->stream, e->ev_imu_done, 0));
    e->imu_early_pending = false;
  }
  if ((rc = sum_small(e, ni, e->pp_err.p + nu + nb, h3 + 2))) return rc;
  if (e->st.n_pp_entries > 0) {
    hipLaunchKernelGGL(k_pp_scatter, dim3(e->st.Pact), dim3(256), 0, e->stream, e->pose_dim, e->st.ld,
                       e->st.ld, e->pp_ptr.p, e->pp_ent.p, e->pose_mask.p + e->st.P, e->pp_h.p,
                       e->pp_g.p, e->A.p, e->rhs_p.p, e->rhs_sc.p);
    BAE_HIP(hipGetLastError());
  }
  return 0;
}

// EvaluateResiduals for the pose-pose residuals (BundleAdjuster.cpp:190-256)
int launch_posepose_eval(Engine* e, double* h3) {  // h3: unary | binary | inertial error sums
  const Problem& pb = e->prob;
  const uint32_t nu = pb.num_unary, nb = pb.num_binary, ni = pb.num_imu;
  if (nu + nb + ni == 0 && !(e->sharded())) return 0;
  int rc;
  const double* state = e->pose_state[e->cur].p;
  if (nu) {
    hipLaunchKernelGGL(k_unary, dim3((nu + 63) / 64), dim3(64), 0, e->stream, (int)nu, 2, 0.0,
                       e->un_pose.p, e->un_t.p, e->un_cov_inv.p, e->un_rot.p, e->un_scale.p, state,
                       e->pp_err.p, e->pp_h.p, e->pp_g.p, e->pp_dz.p, e->pp_info.p, 0u, e->pp_err.p);
    BAE_HIP(hipGetLastError());
  }
  if ((rc = sum_small(e, nu, e->pp_err.p, h3))) return rc;
  if (nb) {
    hipLaunchKernelGGL(k_binary, dim3((nb + 63) / 64), dim3(64), 0, e->stream, (int)nb, 2, e->bin_p1.p,
                       e->bin_p2.p, e->bin_t.p, e->bin_cov_inv.p, e->bin_cov_inv_sqrt.p, e->bin_w.p,
                       e->bin_rot.p, state, e->pp_h.p, e->pp_g.p, e->pp_dz.p, e->pp_info.p, nu,
                       e->pp_err.p + nu);
    BAE_HIP(hipGetLastError());
  }
  if ((rc = sum_small(e, nb, e->pp_err.p + nu, h3 + 1))) return rc;
  if (ni) {
    hipLaunchKernelGGL(k_imu, dim3((ni + 63) / 64), dim3(64), 0, e->stream, (int)ni, 2, e->pose_dim, 0,
                       0.0, e->imu_p1.p, e->imu_p2.p, e->imu_ptr.p, e->imu_meas.p, e->imu_consts.p,
                       e->imu_consts.p + 3, e->pose_active.p, state, e->imu_cov_inv.p, e->pp_h.p,
                       e->pp_g.p, e->pp_dz.p, e->pp_info.p, nu + nb, e->pp_err.p + nu + nb, (double*)nullptr, 0,
                       (double*)nullptr, (uint8_t*)nullptr, (const double*)nullptr);
    BAE_HIP(hipGetLastError());
  }
  if ((rc = sum_small(e, ni, e->pp_err.p + nu + nb, h3 + 2))) return rc;
  return 0;
}

// dogleg denominator terms of the pose-pose residuals
int launch_posepose_jrhs(Engine* e, double* out) {
  const Problem& pb = e->prob;
  const uint32_t nres = pb.num_unary + pb.num_binary + pb.num_imu;
  *out = 0.0;
  if (nres == 0 && !(e->sharded())) return 0;
  if (nres) {
    hipLaunchKernelGGL(k_pp_jrhs, dim3((nres + 63) / 64), dim3(64), 0, e->stream, (int)nres, e->pose_dim,
                       e->pp_res_p1.p, e->pp_res_p2.p, e->pose_opt.p, e->pose_mask.p, e->pp_dz.p,
                       e->pp_info.p, e->rhs_p.p, e->pp_err.p);
    BAE_HIP(hipGetLastError());
  }
  return sum_small(e, nres, e->pp_err.p, out);
}

// The residual vectors of the inertial residuals at the current state (what ImuResidual::residual
// holds after EvaluateResiduals, BundleAdjuster.cpp:225-256), 15 doubles each (the first
// PoseSize are used), into the device buffer d_r15.
int launch_imu_residual_vectors(Engine* e, double* d_r15) {
  const Problem& pb = e->prob;
  const uint32_t nu = pb.num_unary, nb = pb.num_binary, ni = pb.num_imu;
  if (ni == 0) return 0;
  hipLaunchKernelGGL(k_imu, dim3((ni + 63) / 64), dim3(64), 0, e->stream, (int)ni, 2, e->pose_dim, 0, 0.0,
                     e->imu_p1.p, e->imu_p2.p, e->imu_ptr.p, e->imu_meas.p, e->imu_consts.p, e->imu_consts.p + 3,
                     e->pose_active.p, (const double*)e->pose_state[e->cur].p, e->imu_cov_inv.p, e->pp_h.p,
                     e->pp_g.p, e->pp_dz.p, e->pp_info.p, nu + nb, e->pp_err.p + nu + nb, d_r15, 0, (double*)nullptr,
                     (uint8_t*)nullptr, (const double*)nullptr);
  BAE_HIP(hipGetLastError());
  return 0;
}

}  // namespace bae
