// gfx950 kernels of the projection path: pose preparation, residual evaluation,
// per-landmark linearisation (Jacobians, V, b_l, W, factor rows), back-substitution,
// state update, sensor<->world landmark transforms.
//
// Reference code replaced (see include/ba_hip.h for the per-call mapping):
//   parallel_algos.h:35-152        projection residual + Jacobians   -> k_landmarks
//   BundleAdjuster.cpp:1355-1388   Huber weights                     -> k_landmarks
//   BundleAdjuster.cpp:409-443     V, rhs_l, V^-1                    -> k_landmarks
//   BundleAdjuster.cpp:452-460     W, W V^-1                         -> k_landmarks
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
  const double* cm = cam + (size_t)c * 35;
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
                     e->pose_state[e->cur].p, e->cam.p, e->twp.p, e->tws.p, e->tsw.p);
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
// median, parallel_algos.h:143-150).  mode 1: EvaluateResiduals — sum |r|^2 * weight
// and per-landmark outlier counts (BundleAdjuster.cpp:155-187); block partial sums go
// to `partials` and are added in a fixed order by sum_partials (deterministic).
template <int LM>
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
                            uint32_t* __restrict__ lm_outliers, double* __restrict__ partials) {
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
    const double* cp = cam + (size_t)cm * 35;
    // Options::use_per_pose_cam_params (parallel_algos.h:54-57): intrinsics of the measurement pose
    const double* ip = pose_cam ? pose_cam + (size_t)pm * 4 : cp;
    Cam cc = {ip[0], ip[1], ip[2], ip[3]};
    double u, v;
    project(cc, P, &u, &v);
    const double r0 = obs_z[2 * (size_t)a] - u, r1 = obs_z[2 * (size_t)a + 1] - v;
    const double sq = r0 * r0 + r1 * r1;
    val = sq * wts[a];
    if (mode == 0) {
      err[a] = val;
    } else if (sqrt(sq) > outlier_thr) {
      atomicAdd(&lm_outliers[l], 1u);
    }
  }
  if (mode == 1) {
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
template <int LM>
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
  const double* cp = cam + (size_t)cm * 35;
  // Options::use_per_pose_cam_params (parallel_algos.h:54-57): intrinsics of the measurement pose
    const double* ip = pose_cam ? pose_cam + (size_t)pm * 4 : cp;
    Cam cc = {ip[0], ip[1], ip[2], ip[3]};
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
  if (e->lm_dim == 1) hipLaunchKernelGGL(k_residual_vectors<1>, grid, block, 0, e->stream, BAE_ARGS);
  else hipLaunchKernelGGL(k_residual_vectors<3>, grid, block, 0, e->stream, BAE_ARGS);
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
#define BAE_ARGS                                                                          \
  O, (int)e->st.C, mode, e->opt.projection_outlier_threshold, e->obs_z.p, e->obs_pose.p,  \
      e->obs_cam.p, e->obs_lm.p, w, e->lm_x[e->cur].p, e->lm_ref_pose.p, e->lm_ref_cam.p, \
      e->cam.p, e->pose_cam_ptr(), e->tsw.p, e->tws.p, e->obs_e.p, e->lm_outliers.p, e->partials.p
  if (e->lm_dim == 1) hipLaunchKernelGGL(k_residuals<1>, grid, block, 0, e->stream, BAE_ARGS);
  else hipLaunchKernelGGL(k_residuals<3>, grid, block, 0, e->stream, BAE_ARGS);
#undef BAE_ARGS
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// Per-landmark linearisation.  One thread walks the observations of one landmark:
// residual, Jacobians (closed forms of dmath.h), Huber weight, then accumulates
//   V = sum w Jl^T Jl, b_l = sum w Jl^T r, W_inc = sum w Jp^T Jl      (per incidence)
// and emits the factor rows the gather kernels consume:
//   J rows     sqrt(w) * Jp (masked columns zeroed, BundleAdjuster.cpp:1622-1629,1636-1642)
//   W rows     columns of W_inc                      (jt_pr * j_l,   :452)
//   NWV rows   columns of -W_inc V^-1                (W V^-1,        :460)
// plus the scalars sqrt(w) r (r_pr_, :1384-1385) and b_l (rhs_l_, :429-430).
__device__ __forceinline__ void store_row(double* __restrict__ frow, int row, const double* v,
                                          double s) {
  double* o = frow + (size_t)row * kRow;
#pragma unroll
  for (int i = 0; i < 6; ++i) o[i] = v[i] * s;
}

template <int LM>
__global__ void __launch_bounds__(64) k_landmarks(int L, int C, int O, double c_huber, int use_robust,
                            const uint32_t* __restrict__ lm_ptr,
                            const double* __restrict__ obs_z,
                            const uint32_t* __restrict__ obs_pose,
                            const uint32_t* __restrict__ obs_cam,
                            const double* __restrict__ obs_w0,
                            const int32_t* __restrict__ obs_jrow_m,
                            const int32_t* __restrict__ obs_jrow_r,
                            const int32_t* __restrict__ obs_wrow_m,
                            const uint8_t* __restrict__ obs_first,
                            const int32_t* __restrict__ lm_wrow_r,
                            const uint32_t* __restrict__ linc_ptr,
                            const uint32_t* __restrict__ linc_row,
                            const int32_t* __restrict__ lm_opt,
                            const uint16_t* __restrict__ pose_mask,
                            const double* __restrict__ lm_x,
                            const uint32_t* __restrict__ lm_ref_pose,
                            const uint32_t* __restrict__ lm_ref_cam,
                            const double* __restrict__ cam, const double* __restrict__ pose_cam,
                            const double* __restrict__ tsw,
                            const double* __restrict__ tws, const double* __restrict__ twp,
                            double* __restrict__ obs_w, double* __restrict__ frow,
                            double* __restrict__ scal, double* __restrict__ lm_vinv,
                            double* __restrict__ lm_bl, double* __restrict__ obs_jl) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const uint32_t a0 = lm_ptr[l], a1 = lm_ptr[l + 1];
  const bool lm_act = lm_opt[l] >= 0;
  const uint32_t rp = lm_ref_pose[l], rc = lm_ref_cam[l];
  double x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = lm_x[(size_t)l * 4 + i];
  Rt t_ws_r, t_wp_r, t_vs_r;
  uint16_t mask_r = 0;
  if (LM == 1) {
    t_ws_r = load_rt(tws + ((size_t)rp * C + rc) * kRt);
    t_wp_r = load_rt(twp + (size_t)rp * kRt);
    t_vs_r = load_rt(cam + (size_t)rc * 35 + 4);
    mask_r = pose_mask[rp];
  }
  const int wrow_r = (LM == 1) ? lm_wrow_r[l] : -1;
  double V[LM * LM], bl[LM], Wr[6 * LM];
#pragma unroll
  for (int i = 0; i < LM * LM; ++i) V[i] = 0.0;
#pragma unroll
  for (int i = 0; i < LM; ++i) bl[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6 * LM; ++i) Wr[i] = 0.0;

  for (uint32_t a = a0; a < a1; ++a) {
    const uint32_t pm = obs_pose[a], cm = obs_cam[a];
    const double* cp = cam + (size_t)cm * 35;
    // Options::use_per_pose_cam_params (parallel_algos.h:54-57): intrinsics of the measurement pose
    const double* ip = pose_cam ? pose_cam + (size_t)pm * 4 : cp;
    Cam cc = {ip[0], ip[1], ip[2], ip[3]};
    const Rt t_sw_m = load_rt(tsw + ((size_t)pm * C + cm) * kRt);
    const Rt t_wp_m = load_rt(twp + (size_t)pm * kRt);
    const Rt t_sv_m = load_rt(cp + 16);
    if (LM == 3) { t_ws_r = t_sw_m; t_wp_r = t_wp_m; t_vs_r = t_sv_m; }
    const double z[2] = {obs_z[2 * (size_t)a], obs_z[2 * (size_t)a + 1]};
    ProjJac<LM> J;
    const bool same_pose = (LM == 1) && (pm == rp);
    proj_jacobians<LM>(cc, z, x, t_sw_m, t_ws_r, t_wp_m, t_sv_m, t_wp_r, t_vs_r, same_pose, &J);
    // Huber weight (BundleAdjuster.cpp:1374-1387)
    double w = obs_w0[a];
    const double md = (J.r[0] * J.r[0] + J.r[1] * J.r[1]) * w;
    const double en = sqrt(md);
    if (use_robust && en > c_huber) w *= c_huber / en;
    obs_w[a] = w;
    const double sw = sqrt(w);
    scal[2 * (size_t)a] = J.r[0] * sw;
    scal[2 * (size_t)a + 1] = J.r[1] * sw;
#pragma unroll
    for (int i = 0; i < 2 * LM; ++i) obs_jl[(size_t)a * 2 * LM + i] = lm_act ? J.jl[i] * sw : 0.0;
    // column masks of regularised parameters
    const uint16_t mask_m = pose_mask[pm];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (mask_m & (1u << c)) { J.jm[c] = 0.0; J.jm[6 + c] = 0.0; }
      if (LM == 1 && (mask_r & (1u << c))) { J.jr[c] = 0.0; J.jr[6 + c] = 0.0; }
    }
    const int jrm = obs_jrow_m[a], jrr = obs_jrow_r[a];
    if (jrm >= 0) { store_row(frow, jrm, J.jm, sw); store_row(frow, jrm + 1, J.jm + 6, sw); }
    if (LM == 1 && jrr >= 0) { store_row(frow, jrr, J.jr, sw); store_row(frow, jrr + 1, J.jr + 6, sw); }
    if (lm_act) {
#pragma unroll
      for (int p = 0; p < LM; ++p) {
#pragma unroll
        for (int q = 0; q < LM; ++q)
          V[p * LM + q] += (J.jl[p] * J.jl[q] + J.jl[LM + p] * J.jl[LM + q]) * w;
        bl[p] += (J.jl[p] * J.r[0] + J.jl[LM + p] * J.r[1]) * w;
      }
      const int wrm = obs_wrow_m[a];
      if (wrm >= 0) {
        const bool first = obs_first[a] != 0;
#pragma unroll
        for (int k = 0; k < LM; ++k) {
          double* o = frow + (size_t)(wrm + k) * kRow;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const double v = (J.jm[r] * J.jl[k] + J.jm[6 + r] * J.jl[LM + k]) * w;
            o[r] = first ? v : o[r] + v;
          }
        }
      }
      if (LM == 1 && wrow_r >= 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) Wr[r] += (J.jr[r] * J.jl[0] + J.jr[6 + r] * J.jl[1]) * w;
      }
    }
  }
  if (!lm_act) return;
  // V guard (BundleAdjuster.cpp:431-440) and inverse
  double Vi[LM * LM];
  if constexpr (LM == 1) {
    if (fabs(V[0]) < 1e-6) V[0] += 1e-6;
    Vi[0] = 1.0 / V[0];
  } else {
    double nrm = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) nrm += V[i] * V[i];
    if (sqrt(nrm) < 1e-6) { V[0] += 1e-6; V[4] += 1e-6; V[8] += 1e-6; }
    // closed-form 3x3 inverse (cofactors)
    const double a = V[0], b = V[1], c = V[2], d = V[3], ee = V[4], f = V[5], g = V[6],
                 h = V[7], i = V[8];
    const double A00 = ee * i - f * h, A01 = c * h - b * i, A02 = b * f - c * ee;
    const double A10 = f * g - d * i, A11 = a * i - c * g, A12 = c * d - a * f;
    const double A20 = d * h - ee * g, A21 = b * g - a * h, A22 = a * ee - b * d;
    const double id = 1.0 / (a * A00 + b * A10 + c * A20);
    Vi[0] = A00 * id; Vi[1] = A01 * id; Vi[2] = A02 * id;
    Vi[3] = A10 * id; Vi[4] = A11 * id; Vi[5] = A12 * id;
    Vi[6] = A20 * id; Vi[7] = A21 * id; Vi[8] = A22 * id;
  }
#pragma unroll
  for (int i = 0; i < LM * LM; ++i) lm_vinv[(size_t)l * LM * LM + i] = Vi[i];
#pragma unroll
  for (int i = 0; i < LM; ++i) {
    lm_bl[(size_t)l * LM + i] = bl[i];
    scal[2 * (size_t)O + (size_t)l * LM + i] = bl[i];
  }
  if (LM == 1 && wrow_r >= 0) {
    double* o = frow + (size_t)wrow_r * kRow;
#pragma unroll
    for (int r = 0; r < 6; ++r) o[r] = Wr[r];
  }
  // NWV rows = -W V^-1 for every incidence of the landmark
  for (uint32_t q = linc_ptr[l]; q < linc_ptr[l + 1]; ++q) {
    const uint32_t row = linc_row[q];
    double Wq[6 * LM];
#pragma unroll
    for (int k = 0; k < LM; ++k)
#pragma unroll
      for (int r = 0; r < 6; ++r) Wq[k * 6 + r] = frow[(size_t)(row + k) * kRow + r];
#pragma unroll
    for (int c = 0; c < LM; ++c) {
      double* o = frow + (size_t)(row + LM + c) * kRow;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < LM; ++k) s += Wq[k * 6 + r] * Vi[k * LM + c];
        o[r] = -s;
      }
    }
  }
}

int launch_landmarks(Engine* e, double c_huber, int use_robust) {
  const int L = e->st.L;
  if (L == 0 || e->st.O == 0) return 0;
  const dim3 grid((L + 63) / 64), block(64);
#define BAE_ARGS                                                                              \
  L, (int)e->st.C, (int)e->st.O, c_huber, use_robust, e->lm_ptr.p, e->obs_z.p, e->obs_pose.p, \
      e->obs_cam.p, e->obs_w0.p, e->obs_jrow_m.p, e->obs_jrow_r.p, e->obs_wrow_m.p,           \
      e->obs_first.p, e->lm_wrow_r.p, e->linc_ptr.p, e->linc_row.p, e->lm_opt.p,              \
      e->pose_mask.p, e->lm_x[e->cur].p, e->lm_ref_pose.p, e->lm_ref_cam.p, e->cam.p,         \
      e->pose_cam_ptr(), e->tsw.p, e->tws.p, e->twp.p, e->obs_w.p, e->frow.p, e->scal.p, e->lm_vinv.p, e->lm_bl.p,   \
      e->obs_jl.p
  e->prof_begin(e->ev_landmarks);
  if (e->lm_dim == 1) hipLaunchKernelGGL(k_landmarks<1>, grid, block, 0, e->stream, BAE_ARGS);
  else hipLaunchKernelGGL(k_landmarks<3>, grid, block, 0, e->stream, BAE_ARGS);
  e->prof_end(e->ev_landmarks);
#undef BAE_ARGS
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------
// delta_l = V^-1 (b_l - sum_inc W_inc^T delta_p[pose(inc)])   (BundleAdjuster.cpp:709-744)
template <int LM>
__global__ void k_backsub(int L, int D, const int32_t* __restrict__ lm_opt,
                          const uint32_t* __restrict__ linc_ptr,
                          const uint32_t* __restrict__ linc_row,
                          const uint32_t* __restrict__ linc_pose,
                          const double* __restrict__ frow, const double* __restrict__ lm_vinv,
                          const double* __restrict__ lm_bl, const double* __restrict__ delta_p,
                          double* __restrict__ delta_l) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int lo = lm_opt[l];
  if (lo < 0) return;
  double rhs[LM];
#pragma unroll
  for (int i = 0; i < LM; ++i) rhs[i] = lm_bl[(size_t)l * LM + i];
  for (uint32_t q = linc_ptr[l]; q < linc_ptr[l + 1]; ++q) {
    const double* dp = delta_p + (size_t)linc_pose[q] * D;
    const uint32_t row = linc_row[q];
#pragma unroll
    for (int k = 0; k < LM; ++k) {
      const double* wr = frow + (size_t)(row + k) * kRow;
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += wr[r] * dp[r];
      rhs[k] -= s;
    }
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
#define BAE_ARGS                                                                          \
  L, e->pose_dim, e->lm_opt.p, e->linc_ptr.p, e->linc_row.p, e->linc_pose.p, e->frow.p,   \
      e->lm_vinv.p, e->lm_bl.p, e->gn_p.p, e->gn_l.p
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
