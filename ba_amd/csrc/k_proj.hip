// gfx950 kernels of the projection path: pose preparation, residual evaluation,
// per-landmark linearisation (Jacobians, V, b_l, W, factor rows), back-substitution,
// state update, sensor<->world landmark transforms.
//
// Reference code replaced (see include/ba_hip.h for the per-call mapping):
//   parallel_algos.h:35-152        projection residual + Jacobians   -> k_linearize
//   BundleAdjuster.cpp:1355-1388   Huber weights                     -> k_linearize
//   BundleAdjuster.cpp:409-443     V, rhs_l, V^-1                    -> k_linearize
//   BundleAdjuster.cpp:452-460     W, W V^-1                         -> k_linearize
//   BundleAdjuster.cpp:144-187     EvaluateResiduals (projection)    -> k_residuals
//   BundleAdjuster.cpp:709-744     GetLandmarkDelta                  -> k_backsub
//   BundleAdjuster.cpp:21-140      ApplyUpdate                       -> k_apply_*
//   BundleAdjuster.cpp:288-296,672-678  x_s <-> x_w                  -> k_world_to_sensor/_back
#include "engine.h"
#include "dmath.h"

using namespace bad;

namespace bae {

__device__ __forceinline__ Rt load_rt(const double* __restrict__ p) {
  Rt t;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.R.m[i] = p[i];
  t.t = v3(p[9], p[10], p[11]);
  return t;
}
__device__ __forceinline__ void store_rt(double* p, const Rt& t) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p[i] = t.R.m[i];
  p[9] = t.t.x; p[10] = t.t.y; p[11] = t.t.z;
}

// ---------------------------------------------------------------------------------
// T_wp, and per camera T_ws = T_wp T_vs, T_sw = T_ws^-1 (PoseT::GetTsw, Types.h:61-70).
// The product renormalises the quaternion as Sophus' SO3 product does.
__global__ void k_pose_prep(int P, int C, const double* __restrict__ state,
                            const double* __restrict__ cam, double* __restrict__ twp,
                            double* __restrict__ tws, double* __restrict__ tsw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * C) return;
  const int p = i / C, c = i % C;
  const double* s = state + (size_t)p * kPoseState;
  Rt wp;
  wp.R = quat_to_rot(s[3], s[4], s[5], s[6]);
  wp.t = v3(s[0], s[1], s[2]);
  if (c == 0) store_rt(twp + (size_t)p * kRt, wp);
  const double* cm = cam + (size_t)c * kCamRec;
  // T_vs as quaternion is kept at cm[28..34] (t, q) to reproduce the normalised product
  double qv[4] = {cm[31], cm[32], cm[33], cm[34]};
  double qp[4] = {s[3], s[4], s[5], s[6]};
  double q[4];
  quat_mul(qp, qv, q);
  quat_normalize(q);
  Rt ws;
  ws.R = quat_to_rot(q[0], q[1], q[2], q[3]);
  ws.t = mul(wp.R, v3(cm[28], cm[29], cm[30])) + wp.t;
  store_rt(tws + (size_t)i * kRt, ws);
  store_rt(tsw + (size_t)i * kRt, inverse(ws));
}

int launch_pose_prep(Engine* e) {
  const int P = e->st.P, C = e->st.C;
  if (P * C == 0) return 0;
  hipLaunchKernelGGL(k_pose_prep, dim3((P * C + 255) / 256), dim3(256), 0, e->stream, P, C,
                     e->pose_state[e->cur].p, e->cam_eval_ptr(), e->twp.p, e->tws.p, e->tsw.p);
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// x_s = T_sw(ref) x_w / |.|  (BundleAdjuster.cpp:288-296) and back (:672-678)
__global__ void k_world_to_sensor(int L, int C, const double* __restrict__ xw,
                                  const uint32_t* __restrict__ ref_pose,
                                  const uint32_t* __restrict__ ref_cam,
                                  const double* __restrict__ tsw, double* __restrict__ xs) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const Rt T = load_rt(tsw + ((size_t)ref_pose[l] * C + ref_cam[l]) * kRt);
  const double* x = xw + (size_t)l * 4;
  const V3 p = mul(T.R, v3(x[0], x[1], x[2])) + T.t * x[3];
  const double len = sqrt(dot(p, p));
  double* o = xs + (size_t)l * 4;
  o[0] = p.x / len; o[1] = p.y / len; o[2] = p.z / len; o[3] = x[3] / len;
}
__global__ void k_sensor_to_world(int L, int C, const double* __restrict__ xs,
                                  const uint32_t* __restrict__ ref_pose,
                                  const uint32_t* __restrict__ ref_cam,
                                  const double* __restrict__ tws, double* __restrict__ xw) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const Rt T = load_rt(tws + ((size_t)ref_pose[l] * C + ref_cam[l]) * kRt);
  const double* x = xs + (size_t)l * 4;
  const V3 p = mul(T.R, v3(x[0], x[1], x[2])) + T.t * x[3];
  double* o = xw + (size_t)l * 4;
  o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = x[3];
}

int launch_begin_solve(Engine* e) {
  const int L = e->st.L;
  if (e->lm_dim != 1 || L == 0) return 0;
  hipLaunchKernelGGL(k_world_to_sensor, dim3((L + 255) / 256), dim3(256), 0, e->stream, L,
                     (int)e->st.C, e->lm_xw.p, e->lm_ref_pose.p, e->lm_ref_cam.p, e->tsw.p,
                     e->lm_x[e->cur].p);
  BAE_HIP(hipGetLastError());
  return 0;
}
int launch_end_solve(Engine* e) {
  const int L = e->st.L;
  if (e->lm_dim != 1 || L == 0) return 0;
  hipLaunchKernelGGL(k_sensor_to_world, dim3((L + 255) / 256), dim3(256), 0, e->stream, L,
                     (int)e->st.C, e->lm_x[e->cur].p, e->lm_ref_pose.p, e->lm_ref_cam.p,
                     e->tws.p, e->lm_xw.p);
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// One thread per observation (observations sorted by landmark: coalesced reads of the
// observation list, the landmark row is shared by neighbouring lanes, pose transforms
// are gathered through L2).  mode 0: err[a] = |r|^2 * orig_weight (input of the Huber
// median, parallel_algos.h:143-150).  mode 2: sum of |r|^2 over the conditioning residuals
// (SolutionSummary::cond_proj_error, BundleAdjuster.cpp:692-703).  mode 1: EvaluateResiduals — sum |r|^2 * weight
// and per-landmark outlier counts (BundleAdjuster.cpp:155-187); block partial sums go
// to `partials` and are added in a fixed order by sum_partials (deterministic).
// camera of one observation: intrinsics `ip` (the rig camera's or the measurement pose's) with the
// model of the rig camera `cp`.  FOV == false (no FovCamera in the rig): the pinhole as a literal model.
template <bool FOV>
__device__ __forceinline__ Cam load_cam(const double* __restrict__ ip, const double* __restrict__ cp) {
  Cam c = {ip[0], ip[1], ip[2], ip[3], 0.0, 0};
  if constexpr (FOV) { c.w = cp[35]; c.model = cp[36] != 0.0 ? 1 : 0; }
  return c;
}

template <int LM, bool FOV>
__global__ void k_residuals(int O, int C, int mode, double outlier_thr,
                            const double* __restrict__ obs_z,
                            const uint32_t* __restrict__ obs_pose,
                            const uint32_t* __restrict__ obs_cam,
                            const uint32_t* __restrict__ obs_lm,
                            const double* __restrict__ wts, const double* __restrict__ lm_x,
                            const uint32_t* __restrict__ lm_ref_pose,
                            const uint32_t* __restrict__ lm_ref_cam,
                            const double* __restrict__ cam, const double* __restrict__ pose_cam,
                            const double* __restrict__ tsw,
                            const double* __restrict__ tws, double* __restrict__ err,
                            uint32_t* __restrict__ lm_outliers, double* __restrict__ partials,
                            const uint8_t* __restrict__ cond, const double* __restrict__ w0,
                            double* __restrict__ err0) {
  __shared__ double red[256];
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  double val = 0.0;
  if (a < O) {
    const uint32_t l = obs_lm[a], pm = obs_pose[a], cm = obs_cam[a];
    const double* x = lm_x + (size_t)l * 4;
    const Rt t_sw_m = load_rt(tsw + ((size_t)pm * C + cm) * kRt);
    Rt t_ws_r = t_sw_m;
    if (LM == 1) t_ws_r = load_rt(tws + ((size_t)lm_ref_pose[l] * C + lm_ref_cam[l]) * kRt);
    const V3 P = proj_point<LM>(t_sw_m, t_ws_r, x);
    const double* cp = cam + (size_t)cm * kCamRec;
    // Options::use_per_pose_cam_params (parallel_algos.h:54-57): intrinsics of the measurement pose
    const double* ip = pose_cam ? pose_cam + (size_t)pm * 4 : cp;
    const Cam cc = load_cam<FOV>(ip, cp);
    double u, v;
    project(cc, P, &u, &v);
    const double r0 = obs_z[2 * (size_t)a] - u, r1 = obs_z[2 * (size_t)a + 1] - v;
    const double sq = r0 * r0 + r1 * r1;
    val = sq * wts[a];
    if (mode == 0) {
      err[a] = val;
    } else if (mode == 2) {
      val = cond[a] ? sq : 0.0;  // conditioning residuals: |residual|^2, unweighted (BundleAdjuster.cpp:700-703)
    } else {
      // EvaluateResiduals also leaves |r|^2 * original weight — the input of the NEXT linearisation's Huber median at
      // this very state (mode 0 computes the same product): that pass is then skipped (engine.hip: err_cache)
      if (err0) err0[a] = sq * w0[a];
      if (sqrt(sq) > outlier_thr) atomicAdd(&lm_outliers[l], 1u);
    }
  }
  if (mode >= 1) {
    red[threadIdx.x] = val;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
  }
}

// Debug / read-back tap: the residual vector z - pi of every observation at the CURRENT state
// (what ProjectionResidual::residual holds after the last EvaluateResiduals of a Solve(),
// BundleAdjuster.cpp:155-181).
template <int LM, bool FOV>
__global__ void k_residual_vectors(int O, int C, const double* __restrict__ obs_z,
                                   const uint32_t* __restrict__ obs_pose, const uint32_t* __restrict__ obs_cam,
                                   const uint32_t* __restrict__ obs_lm, const double* __restrict__ lm_x,
                                   const uint32_t* __restrict__ lm_ref_pose,
                                   const uint32_t* __restrict__ lm_ref_cam, const double* __restrict__ cam,
                                   const double* __restrict__ pose_cam,
                                   const double* __restrict__ tsw, const double* __restrict__ tws,
                                   double* __restrict__ r2) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= O) return;
  const uint32_t l = obs_lm[a], pm = obs_pose[a], cm = obs_cam[a];
  const double* x = lm_x + (size_t)l * 4;
  const Rt t_sw_m = load_rt(tsw + ((size_t)pm * C + cm) * kRt);
  Rt t_ws_r = t_sw_m;
  if (LM == 1) t_ws_r = load_rt(tws + ((size_t)lm_ref_pose[l] * C + lm_ref_cam[l]) * kRt);
  const V3 P = proj_point<LM>(t_sw_m, t_ws_r, x);
  const double* cp = cam + (size_t)cm * kCamRec;
  // Options::use_per_pose_cam_params (parallel_algos.h:54-57): intrinsics of the measurement pose
    const double* ip = pose_cam ? pose_cam + (size_t)pm * 4 : cp;
    const Cam cc = load_cam<FOV>(ip, cp);
  double u, v;
  project(cc, P, &u, &v);
  r2[2 * (size_t)a] = obs_z[2 * (size_t)a] - u;
  r2[2 * (size_t)a + 1] = obs_z[2 * (size_t)a + 1] - v;
}

int launch_residual_vectors(Engine* e, double* d_r2) {
  const int O = e->st.O;
  if (O == 0) return 0;
  int rc;
  if ((rc = launch_pose_prep(e))) return rc;
  const dim3 grid((O + 255) / 256), block(256);
#define BAE_ARGS                                                                              \
  O, (int)e->st.C, e->obs_z.p, e->obs_pose.p, e->obs_cam.p, e->obs_lm.p, e->lm_x[e->cur].p,   \
      e->lm_ref_pose.p, e->lm_ref_cam.p, e->cam.p, e->pose_cam_ptr(), e->tsw.p, e->tws.p, d_r2
  if (e->has_fov) {
    if (e->lm_dim == 1) hipLaunchKernelGGL((k_residual_vectors<1, true>), grid, block, 0, e->stream, BAE_ARGS);
    else hipLaunchKernelGGL((k_residual_vectors<3, true>), grid, block, 0, e->stream, BAE_ARGS);
  } else {
    if (e->lm_dim == 1) hipLaunchKernelGGL((k_residual_vectors<1, false>), grid, block, 0, e->stream, BAE_ARGS);
    else hipLaunchKernelGGL((k_residual_vectors<3, false>), grid, block, 0, e->stream, BAE_ARGS);
  }
#undef BAE_ARGS
  BAE_HIP(hipGetLastError());
  return 0;
}

int launch_residuals(Engine* e, int mode) {
  const int O = e->st.O;
  if (O == 0) return 0;
  const dim3 grid((O + 255) / 256), block(256);
  if (mode == 1) BAE_HIP(hipMemsetAsync(e->lm_outliers.p, 0, e->lm_outliers.bytes(), e->stream));
  const double* w = mode == 0 ? e->obs_w0.p : e->obs_w.p;
  // mode 1 at the state in buffer `cur`: the median input of a linearisation at that state rides along
  double* err0 = nullptr;
  if (mode == 1 && e->err_cache_on()) {
    BAE_HIP(e->obs_e_state[e->cur].alloc(std::max<size_t>((size_t)O, 1)));
    err0 = e->obs_e_state[e->cur].p;
    e->obs_e_valid[e->cur] = true;
  }
#define BAE_ARGS                                                                          \
  O, (int)e->st.C, mode, e->opt.projection_outlier_threshold, e->obs_z.p, e->obs_pose.p,  \
      e->obs_cam.p, e->obs_lm.p, w, e->lm_x[e->cur].p, e->lm_ref_pose.p, e->lm_ref_cam.p, \
      e->cam.p, e->pose_cam_ptr(), e->tsw.p, e->tws.p, e->obs_e.p, e->lm_outliers.p, e->partials.p,     \
      (const uint8_t*)e->obs_cond.p, (const double*)e->obs_w0.p, err0
  if (e->has_fov) {
    if (e->lm_dim == 1) hipLaunchKernelGGL((k_residuals<1, true>), grid, block, 0, e->stream, BAE_ARGS);
    else hipLaunchKernelGGL((k_residuals<3, true>), grid, block, 0, e->stream, BAE_ARGS);
  } else {
    if (e->lm_dim == 1) hipLaunchKernelGGL((k_residuals<1, false>), grid, block, 0, e->stream, BAE_ARGS);
    else hipLaunchKernelGGL((k_residuals<3, false>), grid, block, 0, e->stream, BAE_ARGS);
  }
#undef BAE_ARGS
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// Linearisation, ONE THREAD PER OBSERVATION (parallel_algos.h:35-152, BundleAdjuster.cpp:1374-1387,
// 409-460).  Observations are sorted by landmark; a wavefront owns a range of whole landmarks with
// at most 64 observations (Lists::wave_rng), so that
//   * the observation records (z, pose id, camera id, weight) are coalesced reads, the landmark row
//     and the reference-pose transforms are wave-level broadcasts, pose transforms come from the
//     L2-resident tables of k_pose_prep;
//   * V = sum w Jl^T Jl, b_l = sum w Jl^T r and the reference-pose block W_r = sum w Jr^T Jl of a
//     landmark are WAVEFRONT-LEVEL SEGMENTED REDUCTIONS over its lanes (shuffle scan, fixed tree:
//     bitwise reproducible), the total is broadcast back, every lane inverts V (1x1, or the 3x3
//     cofactor form) and forms its own W_m and -W_m V^-1 in registers;
//   * the factor rows of the wave (observation-major, structure.h) are staged in LDS and leave as
//     one contiguous span of 16-byte stores — no read-modify-write, no second pass, nothing scattered.
// A landmark with more than 64 observations gets a wave of its own, which walks it twice (sums,
// then rows).  The weighted error sum of BuildProblem (proj_error_, :1386) falls out as a per-wave
// partial.
template <int LM, int CAL = 0>
struct ObsLin {
  double r[2], jm[12], jr[12], jl[2 * LM];
  double w;     // robust weight
  double jk[CAL ? 12 : 1];  // dz_dtvs / dz_dcam_params, two rows of six (calibration instantiations)
};

// Everything one observation contributes, at the current state: residual, Jacobians with the
// columns of regularised parameters zeroed (BundleAdjuster.cpp:1622-1629), Huber weight.
template <int LM, int CAL = 0, bool FOV = false>
__device__ __forceinline__ void linearize_obs(uint32_t a, uint32_t l, int C, double c_huber, int use_robust,
                                              const double* __restrict__ obs_z, const uint32_t* __restrict__ obs_pose,
                                              const uint32_t* __restrict__ obs_cam, const double* __restrict__ obs_w0,
                                              const uint16_t* __restrict__ pose_mask, const double* __restrict__ lm_x,
                                              const uint32_t* __restrict__ lm_ref_pose,
                                              const uint32_t* __restrict__ lm_ref_cam, const double* __restrict__ cam,
                                              const double* __restrict__ pose_cam, const double* __restrict__ tsw,
                                              const double* __restrict__ tws, const double* __restrict__ twp,
                                              ObsLin<LM, CAL>* o, const int32_t* __restrict__ pose_opt = nullptr,
                                              const double* __restrict__ lm_zref = nullptr,
                                              const double* __restrict__ state = nullptr,
                                              const double* __restrict__ cam_cache = nullptr) {
  const uint32_t pm = obs_pose[a], cm = obs_cam[a];
  const double* cp = cam + (size_t)cm * kCamRec;
  // Options::use_per_pose_cam_params (parallel_algos.h:54-57): intrinsics of the measurement pose
  const double* ip = pose_cam ? pose_cam + (size_t)pm * 4 : cp;
  const Cam cc = load_cam<FOV>(ip, cp);
  double x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = lm_x[(size_t)l * 4 + i];
  const Rt t_sw_m = load_rt(tsw + ((size_t)pm * C + cm) * kRt);
  // camera constants: R_vs at cp[4..12], t_sv at cp[25..27] (see build_structure)
  M3 R_vs_m;
#pragma unroll
  for (int i = 0; i < 9; ++i) R_vs_m.m[i] = cp[4 + i];
  const V3 t_sv_m = v3(cp[25], cp[26], cp[27]);
  const double z[2] = {obs_z[2 * (size_t)a], obs_z[2 * (size_t)a + 1]};
  ProjJac<LM> J;
  uint16_t mask_r = 0;
  if constexpr (LM == 1) {
    const uint32_t rp = lm_ref_pose[l], rc = lm_ref_cam[l];
    const Rt t_ws_r = load_rt(tws + ((size_t)rp * C + rc) * kRt);
    const Rt t_wp_r = load_rt(twp + (size_t)rp * kRt);
    mask_r = pose_mask[rp];
    proj_linearize<1, CAL == 1>(cc, z, x, t_sw_m, R_vs_m, t_sv_m, t_ws_r, t_wp_r, pm == rp, &J, o->jk);
    if constexpr (CAL != 0) {
      // parallel_algos.h:88,114,120: the calibration Jacobians are only formed when one of the two poses is active
      const double keep = (pose_opt[pm] >= 0 || pose_opt[rp] >= 0) ? 1.0 : 0.0;
      if constexpr (CAL == 3) {
        // T_vs calibration between a rejected step and the next applied one: the tables (residual, J_l)
        // hold the T_vs before the step, the rig the step itself — the three Jacobian blocks from the
        // reference's chains with the two kept apart (dmath.h: proj_chain_two_tvs).  Rare path.
        proj_chain_two_tvs(cc, x, state + (size_t)pm * kPoseState, state + (size_t)rp * kPoseState,
                           cam + (size_t)cm * kCamRec + 28, cam_cache + (size_t)cm * kCamRec + 28, pm == rp, J.jm, J.jr, o->jk);
#pragma unroll
        for (int i = 0; i < 12; ++i) o->jk[i] *= keep;
      } else if constexpr (CAL == 1) {
#pragma unroll
        for (int i = 0; i < 12; ++i) o->jk[i] *= keep;
      } else {
        const double zr[2] = {lm_zref[2 * (size_t)l], lm_zref[2 * (size_t)l + 1]};
        proj_intrinsics_rows(cc, zr, x[3], t_sw_m, t_ws_r, pm == rp ? 0.0 : keep, o->jk);
      }
    }
  } else {
    proj_linearize<LM>(cc, z, x, t_sw_m, R_vs_m, t_sv_m, t_sw_m, t_sw_m, false, &J);
  }
  // Huber weight (BundleAdjuster.cpp:1374-1387)
  double w = obs_w0[a];
  const double en = sqrt((J.r[0] * J.r[0] + J.r[1] * J.r[1]) * w);
  if (use_robust && en > c_huber) w *= c_huber / en;
  o->w = w;
  o->r[0] = J.r[0]; o->r[1] = J.r[1];
  const uint16_t mask_m = pose_mask[pm];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const bool zm = (mask_m >> c) & 1u, zr = (mask_r >> c) & 1u;
    o->jm[c] = zm ? 0.0 : J.jm[c];
    o->jm[6 + c] = zm ? 0.0 : J.jm[6 + c];
    o->jr[c] = (LM != 1 || zr) ? 0.0 : J.jr[c];
    o->jr[6 + c] = (LM != 1 || zr) ? 0.0 : J.jr[6 + c];
  }
#pragma unroll
  for (int i = 0; i < 2 * LM; ++i) o->jl[i] = J.jl[i];
}

// V guard (BundleAdjuster.cpp:431-440) and inverse; Vs = the LM (LM + 1) / 2 unique entries
// (00 | 00 01 02 11 12 22), Vi row-major LM x LM
template <int LM>
__device__ __forceinline__ void invert_v(const double* Vs, double* Vi) {
  if constexpr (LM == 1) {
    double v = Vs[0];
    if (fabs(v) < 1e-6) v += 1e-6;
    Vi[0] = 1.0 / v;
  } else {
    double a = Vs[0], b = Vs[1], c = Vs[2], ee = Vs[3], f = Vs[4], i = Vs[5];
    const double nrm = a * a + ee * ee + i * i + 2.0 * (b * b + c * c + f * f);
    if (sqrt(nrm) < 1e-6) { a += 1e-6; ee += 1e-6; i += 1e-6; }
    const double d = b, g = c, h = f;  // symmetric
    // closed-form 3x3 inverse (cofactors)
    const double A00 = ee * i - f * h, A01 = c * h - b * i, A02 = b * f - c * ee;
    const double A10 = f * g - d * i, A11 = a * i - c * g, A12 = c * d - a * f;
    const double A20 = d * h - ee * g, A21 = b * g - a * h, A22 = a * ee - b * d;
    const double id = 1.0 / (a * A00 + b * A10 + c * A20);
    Vi[0] = A00 * id; Vi[1] = A01 * id; Vi[2] = A02 * id;
    Vi[3] = A10 * id; Vi[4] = A11 * id; Vi[5] = A12 * id;
    Vi[6] = A20 * id; Vi[7] = A21 * id; Vi[8] = A22 * id;
  }
}

// lane-local contributions to the landmark sums: [V unique | b_l | W_r (LM == 1)]
template <int LM, int CAL = 0> struct LmSums {
  static constexpr int NV = LM * (LM + 1) / 2, NE = NV + LM + (LM == 1 ? 6 : 0), N = NE + (CAL ? 6 : 0);
};
template <int LM, int CAL = 0>
__device__ __forceinline__ void obs_sums(const ObsLin<LM, CAL>& q, double* v) {
  int k = 0;
#pragma unroll
  for (int p = 0; p < LM; ++p)
#pragma unroll
    for (int c = p; c < LM; ++c) v[k++] = (q.jl[p] * q.jl[c] + q.jl[LM + p] * q.jl[LM + c]) * q.w;
#pragma unroll
  for (int p = 0; p < LM; ++p) v[k++] = (q.jl[p] * q.r[0] + q.jl[LM + p] * q.r[1]) * q.w;
  if constexpr (LM == 1) {
#pragma unroll
    for (int x = 0; x < 6; ++x) v[k++] = (q.jr[x] * q.jl[0] + q.jr[6 + x] * q.jl[1]) * q.w;
    if constexpr (CAL) {  // E_l = sum w J_l^T J_k (jt_kpr_ j_l_, BundleAdjuster.cpp:534-536)
#pragma unroll
      for (int x = 0; x < 6; ++x) v[k++] = (q.jk[x] * q.jl[0] + q.jk[6 + x] * q.jl[1]) * q.w;
    }
  }
}

// the R rows of one observation (structure.h): J_m (2), [J_r (2)], W_m (LM), -W_m V^-1 (LM)
template <int LM, int CAL = 0>
__device__ __forceinline__ void obs_rows(const ObsLin<LM, CAL>& q, const double* Vi, double* rows) {
  const double sw = sqrt(q.w);
#pragma unroll
  for (int i = 0; i < 12; ++i) rows[i] = q.jm[i] * sw;
  constexpr int WO = (LM == 1 ? 4 : 2) * 6;
  if constexpr (LM == 1) {
#pragma unroll
    for (int i = 0; i < 12; ++i) rows[12 + i] = q.jr[i] * sw;
  }
  double W[6 * LM];
#pragma unroll
  for (int k = 0; k < LM; ++k)
#pragma unroll
    for (int x = 0; x < 6; ++x) {
      W[k * 6 + x] = (q.jm[x] * q.jl[k] + q.jm[6 + x] * q.jl[LM + k]) * q.w;
      rows[WO + k * 6 + x] = W[k * 6 + x];
    }
#pragma unroll
  for (int c = 0; c < LM; ++c)
#pragma unroll
    for (int x = 0; x < 6; ++x) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < LM; ++k) s += W[k * 6 + x] * Vi[k * LM + c];
      rows[WO + (LM + c) * 6 + x] = -s;
    }
}

// landmark outputs: V^-1, b_l (also a scalar of the rhs lists), and for LM == 1 the reference
// pose's rows W_r, -W_r V^-1
template <int LM>
__device__ __forceinline__ void store_landmark(uint32_t l, uint32_t O, uint32_t lrow_base, const double* tot,
                                               const double* Vi, double* __restrict__ lm_vinv,
                                               double* __restrict__ lm_bl, double* __restrict__ scal,
                                               double* __restrict__ frow) {
  constexpr int NV = LmSums<LM>::NV;
#pragma unroll
  for (int i = 0; i < LM * LM; ++i) lm_vinv[(size_t)l * LM * LM + i] = Vi[i];
#pragma unroll
  for (int i = 0; i < LM; ++i) {
    lm_bl[(size_t)l * LM + i] = tot[NV + i];
    scal[2 * (size_t)O + (size_t)l * LM + i] = tot[NV + i];
  }
  if constexpr (LM == 1) {
    double* o = frow + ((size_t)lrow_base + 2 * (size_t)l) * kRow;
#pragma unroll
    for (int x = 0; x < 6; ++x) { o[x] = tot[NV + 1 + x]; o[6 + x] = -tot[NV + 1 + x] * Vi[0]; }
  }
}

// CAL (calibration instantiations, LM == 1): additionally the calibration rows `crow` (engine.h) —
// sqrt(w) dz_dtvs of the observation at 2a, 2a+1 and the landmark's E_l = sum w J_l^T J_k at 2O + l
// (six more components of the segmented sums).
template <int LM, int WAVES, bool BIG, bool STAGE, int CAL = 0, bool FOV = false>
__global__ void __launch_bounds__(64 * WAVES)
k_linearize(uint32_t n_chunks, int C, uint32_t O, uint32_t lrow_base, double c_huber, int use_robust,
            const uint2* __restrict__ wave_rng, const uint32_t* __restrict__ lm_ptr,
            const double* __restrict__ obs_z, const uint32_t* __restrict__ obs_pose,
            const uint32_t* __restrict__ obs_cam, const uint32_t* __restrict__ obs_lm,
            const double* __restrict__ obs_w0, const int32_t* __restrict__ lm_opt,
            const uint16_t* __restrict__ pose_mask, const double* __restrict__ lm_x,
            const uint32_t* __restrict__ lm_ref_pose, const uint32_t* __restrict__ lm_ref_cam,
            const double* __restrict__ cam, const double* __restrict__ pose_cam,
            const double* __restrict__ tsw, const double* __restrict__ tws, const double* __restrict__ twp,
            double* __restrict__ obs_w, double* __restrict__ frow, double* __restrict__ scal,
            double* __restrict__ lm_vinv, double* __restrict__ lm_bl, double* __restrict__ obs_jl,
            double* __restrict__ partials, const int32_t* __restrict__ pose_opt, double* __restrict__ crow,
            const double* __restrict__ lm_zref, const double* __restrict__ state,
            const double* __restrict__ cam_cache) {
  constexpr int R = LM == 1 ? 6 : 8, RD = R * 6;   // rows / doubles per observation
  constexpr int STRIDE = RD + 2;                    // LDS stride per lane (even: 16-byte reads; odd multiple of 2 banks)
  constexpr int NS = LmSums<LM, CAL>::N, NV = LmSums<LM, CAL>::NV, NE = LmSums<LM, CAL>::NE;
  // the calibration rows of one observation / of one landmark
  auto store_calib_obs = [&](uint32_t a, const ObsLin<LM, CAL>& q) {
    if constexpr (CAL) {
      const double sw = sqrt(q.w);
      double* o = crow + 2 * (size_t)a * kRow;
#pragma unroll
      for (int i = 0; i < 12; i += 2) *reinterpret_cast<double2*>(o + i) = make_double2(q.jk[i] * sw, q.jk[i + 1] * sw);
    }
  };
  auto store_calib_lm = [&](uint32_t l, const double* tot) {
    if constexpr (CAL) {
      double* o = crow + (2 * (size_t)O + l) * kRow;
#pragma unroll
      for (int i = 0; i < 6; i += 2) *reinterpret_cast<double2*>(o + i) = make_double2(tot[NE + i], tot[NE + i + 1]);
    }
  };
  __shared__ __attribute__((aligned(16))) double stage[(BIG || !STAGE) ? 1 : WAVES][(BIG || !STAGE) ? 2 : 64 * STRIDE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t chunk = blockIdx.x * WAVES + wave;
  if (chunk >= n_chunks) return;  // waves are independent: no block-level barrier below
  const uint2 rng = wave_rng[chunk];
  const uint32_t a0 = rng.x, a1 = rng.y, nobs = a1 - a0;
#define BAE_LIN_ARGS C, c_huber, use_robust, obs_z, obs_pose, obs_cam, obs_w0, pose_mask, lm_x, lm_ref_pose, \
                     lm_ref_cam, cam, pose_cam, tsw, tws, twp
  double err = 0.0;
  if constexpr (!BIG) {
    const bool valid = (uint32_t)lane < nobs;
    const uint32_t a = valid ? a0 + lane : a1 - 1;
    const uint32_t l = obs_lm[a];
    const bool lm_act = lm_opt[l] >= 0;
    const int s0 = (int)(lm_ptr[l] - a0), s1 = (int)(lm_ptr[l + 1] - 1 - a0);  // lanes of this landmark
    ObsLin<LM, CAL> q;
    linearize_obs<LM, CAL, FOV>(a, l, BAE_LIN_ARGS, &q, pose_opt, lm_zref, state, cam_cache);
    double v[NS];
    obs_sums<LM, CAL>(q, v);
    if (!valid || !lm_act) {
#pragma unroll
      for (int i = 0; i < NS; ++i) v[i] = 0.0;
    }
    // segmented inclusive scan over the lanes of a landmark, then the total from its last lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const bool take = lane - off >= s0;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const double up = __shfl_up(v[i], off, 64);
        if (take) v[i] += up;
      }
    }
    double tot[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) tot[i] = __shfl(v[i], s1, 64);
    double Vi[LM * LM];
    invert_v<LM>(tot, Vi);
    if constexpr (STAGE) {
      double* st = stage[wave] + lane * STRIDE;
      double rows[RD];
      obs_rows<LM, CAL>(q, Vi, rows);
#pragma unroll
      for (int i = 0; i < RD; i += 2) *reinterpret_cast<double2*>(st + i) = make_double2(rows[i], rows[i + 1]);
    } else if (valid) {
      // direct variant: every lane stores its own RD doubles (lane stride RD * 8 bytes); the L2 merges
      // the partial lines of the wave's contiguous span
      double rows[RD];
      obs_rows<LM, CAL>(q, Vi, rows);
      double* dst = frow + (size_t)a * RD;
#pragma unroll
      for (int i = 0; i < RD; i += 2) *reinterpret_cast<double2*>(dst + i) = make_double2(rows[i], rows[i + 1]);
    }
    if (valid) {
      const double sw = sqrt(q.w);
      obs_w[a] = q.w;
      *reinterpret_cast<double2*>(scal + 2 * (size_t)a) = make_double2(q.r[0] * sw, q.r[1] * sw);
#pragma unroll
      for (int i = 0; i < 2 * LM; ++i) obs_jl[(size_t)a * 2 * LM + i] = lm_act ? q.jl[i] * sw : 0.0;
      err = (q.r[0] * q.r[0] + q.r[1] * q.r[1]) * q.w;
      store_calib_obs(a, q);
      if (lane == s1 && lm_act) {
        store_landmark<LM>(l, O, lrow_base, tot, Vi, lm_vinv, lm_bl, scal, frow);
        store_calib_lm(l, tot);
      }
    }
    // LDS image -> one contiguous span of the factor rows (16 bytes per lane per store)
    if constexpr (STAGE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t total = nobs * RD;
    double* dst = frow + (size_t)a0 * RD;
    const double* img = stage[wave];
    for (uint32_t g = 2 * lane; g < total; g += 128) {
      const uint32_t ln = g / RD, k = g - ln * RD;
      // written once, read by the assembly kernels after 3 GB more have gone by: streaming stores (measured
      // 1.12 -> 1.05-1.09 ms at configs[3])
      typedef double d2_t __attribute__((ext_vector_type(2)));
      __builtin_nontemporal_store(*reinterpret_cast<const d2_t*>(img + ln * STRIDE + k), reinterpret_cast<d2_t*>(dst + g));
    }
    }
  } else {
    // one landmark with more than 64 observations: pass 1 sums, pass 2 rows (direct stores)
    (void)stage;
    const uint32_t l = obs_lm[a0];
    const bool lm_act = lm_opt[l] >= 0;
    double tot[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) tot[i] = 0.0;
    for (uint32_t a = a0 + lane; a < a1; a += 64) {
      ObsLin<LM, CAL> q;
      linearize_obs<LM, CAL, FOV>(a, l, BAE_LIN_ARGS, &q, pose_opt, lm_zref, state, cam_cache);
      double v[NS];
      obs_sums<LM, CAL>(q, v);
#pragma unroll
      for (int i = 0; i < NS; ++i) tot[i] += lm_act ? v[i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) tot[i] += __shfl_xor(tot[i], off, 64);
    }
    double Vi[LM * LM];
    invert_v<LM>(tot, Vi);
    for (uint32_t a = a0 + lane; a < a1; a += 64) {
      ObsLin<LM, CAL> q;
      linearize_obs<LM, CAL, FOV>(a, l, BAE_LIN_ARGS, &q, pose_opt, lm_zref, state, cam_cache);
      double rows[RD];
      obs_rows<LM, CAL>(q, Vi, rows);
      store_calib_obs(a, q);
      double* dst = frow + (size_t)a * RD;
#pragma unroll
      for (int i = 0; i < RD; i += 2) *reinterpret_cast<double2*>(dst + i) = make_double2(rows[i], rows[i + 1]);
      const double sw = sqrt(q.w);
      obs_w[a] = q.w;
      *reinterpret_cast<double2*>(scal + 2 * (size_t)a) = make_double2(q.r[0] * sw, q.r[1] * sw);
#pragma unroll
      for (int i = 0; i < 2 * LM; ++i) obs_jl[(size_t)a * 2 * LM + i] = lm_act ? q.jl[i] * sw : 0.0;
      err += (q.r[0] * q.r[0] + q.r[1] * q.r[1]) * q.w;
    }
    if (lane == 0 && lm_act) {
      store_landmark<LM>(l, O, lrow_base, tot, Vi, lm_vinv, lm_bl, scal, frow);
      store_calib_lm(l, tot);
    }
  }
#undef BAE_LIN_ARGS
  // weighted error of the wave (fixed tree) -> one partial per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) err += __shfl_xor(err, off, 64);
  if (lane == 0) partials[chunk] = err;
  (void)NV; (void)NE;
}

// one launch of k_linearize: the FOV instantiation when the rig holds a FovCamera
#define BAE_LIN_LAUNCH(LMv, WAVESv, BIGv, STAGEv, CALv, grid_, block_, shm_, stream_, ...)                        \
  do {                                                                                                          \
    if (e->has_fov) hipLaunchKernelGGL((k_linearize<LMv, WAVESv, BIGv, STAGEv, CALv, true>), grid_, block_, shm_, stream_, __VA_ARGS__); \
    else hipLaunchKernelGGL((k_linearize<LMv, WAVESv, BIGv, STAGEv, CALv, false>), grid_, block_, shm_, stream_, __VA_ARGS__);           \
  } while (0)

int launch_landmarks(Engine* e, double c_huber, int use_robust) {
  const Structure& st = e->st;
  if (st.n_chunks == 0 || st.O == 0) return 0;
  constexpr int WAVES = 2;
  // waves [0, n_small): ranges of whole landmarks with at most 64 observations; [n_small, n_chunks):
  // one landmark with more than 64 each (two-pass variant of the kernel)
  const uint32_t n_small = st.n_chunks - st.n_big_chunks;
#define BAE_ARGS(first, count)                                                                          \
  count, (int)st.C, st.O, st.lrow_base, c_huber, use_robust, e->wave_rng.p + (first), e->lm_ptr.p, e->obs_z.p, \
      e->obs_pose.p, e->obs_cam.p, e->obs_lm.p, e->obs_w0.p, e->lm_opt.p, e->pose_mask.p, e->lm_x[e->cur].p, \
      e->lm_ref_pose.p, e->lm_ref_cam.p, e->cam.p, e->pose_cam_ptr(), e->tsw.p, e->tws.p, e->twp.p,     \
      e->obs_w.p, e->frow.p, e->scal.p, e->lm_vinv.p, e->lm_bl.p, e->obs_jl.p, e->partials.p + (first),  \
      (const int32_t*)e->pose_opt.p, e->crow.p, (const double*)e->lm_zref.p,                           \
      (const double*)e->pose_state[e->cur].p, e->cam_eval_ptr()
  e->prof_begin(e->ev_landmarks);
  if (st.K) {  // calibration instantiations (LmSize 1): the CAL kernels, same launch shapes
    // T_vs calibration with the tables of k_pose_prep built for another T_vs than the rig's (after a
    // rejected step, until the next applied one): the chain variant (CAL 3), direct stores
    const bool two_tvs = e->calib_tvs && e->tvs_eval != e->prob.cam_tvs;
    if (two_tvs) {
      const dim3 grid((st.n_chunks + WAVES - 1) / WAVES), block(64 * WAVES);
      if (n_small)
        BAE_LIN_LAUNCH(1, WAVES, false, false, 3, dim3((n_small + WAVES - 1) / WAVES), block, 0, e->stream, BAE_ARGS(0, n_small));
      if (st.n_big_chunks)
        BAE_LIN_LAUNCH(1, WAVES, true, false, 3, dim3((st.n_big_chunks + WAVES - 1) / WAVES), block, 0, e->stream,
                           BAE_ARGS(n_small, st.n_big_chunks));
      (void)grid;
      e->prof_end(e->ev_landmarks);
      BAE_HIP(hipGetLastError());
      return 0;
    }
    if (n_small) {
      const dim3 grid((n_small + WAVES - 1) / WAVES), block(64 * WAVES);
      if (e->calib_tvs) BAE_LIN_LAUNCH(1, WAVES, false, true, 1, grid, block, 0, e->stream, BAE_ARGS(0, n_small));
      else BAE_LIN_LAUNCH(1, WAVES, false, true, 2, grid, block, 0, e->stream, BAE_ARGS(0, n_small));
    }
    if (st.n_big_chunks) {
      const dim3 grid((st.n_big_chunks + WAVES - 1) / WAVES), block(64 * WAVES);
      if (e->calib_tvs)
        BAE_LIN_LAUNCH(1, WAVES, true, false, 1, grid, block, 0, e->stream, BAE_ARGS(n_small, st.n_big_chunks));
      else
        BAE_LIN_LAUNCH(1, WAVES, true, false, 2, grid, block, 0, e->stream, BAE_ARGS(n_small, st.n_big_chunks));
    }
    e->prof_end(e->ev_landmarks);
    BAE_HIP(hipGetLastError());
    return 0;
  }
  if (n_small) {
    if (e->dbg_linearize_variant == 0) {
      const dim3 grid((n_small + WAVES - 1) / WAVES), block(64 * WAVES);
      if (e->lm_dim == 1) BAE_LIN_LAUNCH(1, WAVES, false, true, 0, grid, block, 0, e->stream, BAE_ARGS(0, n_small));
      else BAE_LIN_LAUNCH(3, WAVES, false, true, 0, grid, block, 0, e->stream, BAE_ARGS(0, n_small));
    } else {
      const dim3 grid((n_small + 3) / 4), block(256);
      if (e->lm_dim == 1) BAE_LIN_LAUNCH(1, 4, false, false, 0, grid, block, 0, e->stream, BAE_ARGS(0, n_small));
      else BAE_LIN_LAUNCH(3, 4, false, false, 0, grid, block, 0, e->stream, BAE_ARGS(0, n_small));
    }
  }
  if (st.n_big_chunks) {
    const dim3 grid((st.n_big_chunks + WAVES - 1) / WAVES), block(64 * WAVES);
    if (e->lm_dim == 1) BAE_LIN_LAUNCH(1, WAVES, true, false, 0, grid, block, 0, e->stream, BAE_ARGS(n_small, st.n_big_chunks));
    else BAE_LIN_LAUNCH(3, WAVES, true, false, 0, grid, block, 0, e->stream, BAE_ARGS(n_small, st.n_big_chunks));
  }
  e->prof_end(e->ev_landmarks);
#undef BAE_ARGS
#undef BAE_LIN_LAUNCH
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// delta_l = V^-1 (b_l - sum_inc W_inc^T delta_p[pose(inc)])   (BundleAdjuster.cpp:709-744).
// One thread per landmark walks its observations: the incidence of an observation is its W rows
// (structure.h) when the observation is listed and its measuring pose active; plus the reference
// pose's rows (LM == 1).
template <int LM>
__global__ void k_backsub(int L, int D, uint32_t lrow_base, const int32_t* __restrict__ lm_opt,
                          const int32_t* __restrict__ pose_opt, const uint32_t* __restrict__ lm_ptr,
                          const uint32_t* __restrict__ obs_pose, const uint32_t* __restrict__ lm_ref_pose,
                          const double* __restrict__ frow, const double* __restrict__ lm_vinv,
                          const double* __restrict__ lm_bl, const double* __restrict__ delta_p,
                          double* __restrict__ delta_l, const double* __restrict__ crow_lm,
                          const double* __restrict__ delta_k, int K) {
  constexpr int R = LM == 1 ? 6 : 8, WO = LM == 1 ? 4 : 2;
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int lo = lm_opt[l];
  if (lo < 0) return;
  double rhs[LM];
#pragma unroll
  for (int i = 0; i < LM; ++i) rhs[i] = lm_bl[(size_t)l * LM + i];
  const uint32_t rp = lm_ref_pose[l];
  bool any_listed = false;
  for (uint32_t a = lm_ptr[l]; a < lm_ptr[l + 1]; ++a) {
    const uint32_t pm = obs_pose[a];
    if (LM == 1 && pm == rp) continue;  // not listed (BundleAdjuster.h:489-497)
    any_listed = true;
    const int po = pose_opt[pm];
    if (po < 0) continue;
    const double* dp = delta_p + (size_t)po * D;
#pragma unroll
    for (int k = 0; k < LM; ++k) {
      const double* wr = frow + ((size_t)a * R + WO + k) * kRow;
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += wr[r] * dp[r];
      rhs[k] -= s;
    }
  }
  if (LM == 1 && any_listed && pose_opt[rp] >= 0) {
    const double* dp = delta_p + (size_t)pose_opt[rp] * D;
    const double* wr = frow + ((size_t)lrow_base + 2 * (size_t)l) * kRow;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += wr[r] * dp[r];
    rhs[0] -= s;
  }
  if (LM == 1 && crow_lm) {  // calibration: rhs_l -= (J_l^T J_k) delta_k (BundleAdjuster.cpp:729-733)
    const double* er = crow_lm + (size_t)l * kRow;
    double s = 0.0;
    for (int r = 0; r < K; ++r) s += er[r] * delta_k[r];
    rhs[0] -= s;
  }
#pragma unroll
  for (int a = 0; a < LM; ++a) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < LM; ++b) s += lm_vinv[(size_t)l * LM * LM + a * LM + b] * rhs[b];
    delta_l[(size_t)lo * LM + a] = s;
  }
}

int launch_backsub(Engine* e) {
  const int L = e->st.L;
  if (L == 0 || e->lm_dim == 0 || e->st.Lact == 0) return 0;
  const dim3 grid((L + 255) / 256), block(256);
#define BAE_ARGS                                                                                      \
  L, e->pose_dim, e->st.lrow_base, e->lm_opt.p, e->pose_opt.p, e->lm_ptr.p, e->obs_pose.p, e->lm_ref_pose.p, \
      e->frow.p, e->lm_vinv.p, e->lm_bl.p, e->gn_p.p, e->gn_l.p,                                    \
      (const double*)(e->st.K ? e->crow.p + 2 * (size_t)e->st.O * kRow : nullptr),                    \
      (const double*)(e->gn_p.p + e->st.np), (int)e->st.K
  if (e->lm_dim == 1) hipLaunchKernelGGL(k_backsub<1>, grid, block, 0, e->stream, BAE_ARGS);
  else hipLaunchKernelGGL(k_backsub<3>, grid, block, 0, e->stream, BAE_ARGS);
#undef BAE_ARGS
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// ApplyUpdate (BundleAdjuster.cpp:84-139): x <- x [+] (-delta).
__global__ void k_apply_poses(int P, int D, const int32_t* __restrict__ pose_opt,
                              const double* __restrict__ step_p, const double* __restrict__ in,
                              double* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const double* s = in + (size_t)p * kPoseState;
  double* o = out + (size_t)p * kPoseState;
  double st[kPoseState];
#pragma unroll
  for (int i = 0; i < kPoseState; ++i) st[i] = s[i];
  const int po = pose_opt[p];
  if (po >= 0) {
    const double* d = step_p + (size_t)po * D;
    // exp_decoupled(T, -delta): t += -d[0:3], R <- R exp(-d[3:6])   (Utils.h:364-369)
    st[0] -= d[0]; st[1] -= d[1]; st[2] -= d[2];
    double qe[4], q[4];
    so3_exp(v3(-d[3], -d[4], -d[5]), qe);
    quat_mul(st + 3, qe, q);
    quat_normalize(q);
    st[3] = q[0]; st[4] = q[1]; st[5] = q[2]; st[6] = q[3];
    if (D >= 9) { st[7] -= d[6]; st[8] -= d[7]; st[9] -= d[8]; }
    if (D >= 15) {
#pragma unroll
      for (int i = 0; i < 6; ++i) st[10 + i] -= d[9 + i];
    }
  }
#pragma unroll
  for (int i = 0; i < kPoseState; ++i) o[i] = st[i];
}

template <int LM>
__global__ void k_apply_landmarks(int L, const int32_t* __restrict__ lm_opt,
                                  const double* __restrict__ step_l,
                                  const double* __restrict__ in, const uint8_t* __restrict__ rel_in,
                                  double* __restrict__ out, uint8_t* __restrict__ rel_out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  double x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = in[(size_t)l * 4 + i];
  uint8_t rel = rel_in[l];
  const int lo = lm_opt[l];
  if (lo >= 0) {
    if (LM == 1) {
      const double d = step_l[lo];
      x[3] -= d;
      if (x[3] < 0) { x[3] += d; rel = 0; }  // BundleAdjuster.cpp:127-134
    } else {
#pragma unroll
      for (int i = 0; i < LM; ++i) x[i] -= step_l[(size_t)lo * LM + i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[(size_t)l * 4 + i] = x[i];
  rel_out[l] = rel;
}

// Intrinsics calibration, BundleAdjuster.cpp:57-68: after the camera parameters moved, the sensor-frame
// ray of EVERY landmark is re-derived from its reference pixel, x_s[0:3] = unit(Unproject(z_ref)) |x_s[0:3]|.
__global__ void k_reset_rays(int L, const double* __restrict__ cam, const double* __restrict__ zref,
                             double* __restrict__ x) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  double* o = x + (size_t)l * 4;
  const double norm = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
  double rx = (zref[2 * (size_t)l] - cam[2]) / cam[0], ry = (zref[2 * (size_t)l + 1] - cam[3]) / cam[1];
  if (cam[36] != 0.0) {  // FovCamera: the un-distorted ray
    double g, dg_dr, dg_dw;
    fov_factor_inv(cam[35], sqrt(rx * rx + ry * ry), &g, &dg_dr, &dg_dw);
    rx *= g; ry *= g;
  }
  const double s = norm / sqrt(rx * rx + ry * ry + 1.0);
  o[0] = rx * s; o[1] = ry * s; o[2] = s;
}
int launch_reset_rays(Engine* e) {
  const int L = e->st.L;
  if (L == 0) return 0;
  hipLaunchKernelGGL(k_reset_rays, dim3((L + 255) / 256), dim3(256), 0, e->stream, L, (const double*)e->cam.p,
                     (const double*)e->lm_zref.p, e->lm_x[e->cur].p);
  BAE_HIP(hipGetLastError());
  return 0;
}

int launch_apply_step(Engine* e) {
  const int P = e->st.P, L = e->st.L, nxt = 1 - e->cur;
  if (P > 0) {
    hipLaunchKernelGGL(k_apply_poses, dim3((P + 255) / 256), dim3(256), 0, e->stream, P,
                       e->pose_dim, e->pose_opt.p, e->step_p.p, e->pose_state[e->cur].p,
                       e->pose_state[nxt].p);
    BAE_HIP(hipGetLastError());
  }
  if (L > 0 && e->lm_dim > 0) {
    const dim3 grid((L + 255) / 256), block(256);
    if (e->lm_dim == 1)
      hipLaunchKernelGGL(k_apply_landmarks<1>, grid, block, 0, e->stream, L, e->lm_opt.p,
                         e->step_l.p, e->lm_x[e->cur].p, e->lm_reliable[e->cur].p,
                         e->lm_x[nxt].p, e->lm_reliable[nxt].p);
    else
      hipLaunchKernelGGL(k_apply_landmarks<3>, grid, block, 0, e->stream, L, e->lm_opt.p,
                         e->step_l.p, e->lm_x[e->cur].p, e->lm_reliable[e->cur].p,
                         e->lm_x[nxt].p, e->lm_reliable[nxt].p);
    BAE_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace bae
