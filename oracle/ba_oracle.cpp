// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (single-threaded C++17, no Eigen/Sophus/Calibu) of the
// Gauss-Newton path of arpg/ba:
//   /root/reference/src/BundleAdjuster.cpp           Solve, BuildProblem, SolveInternal,
//                                                    CalculateGn, GetLandmarkDelta,
//                                                    EvaluateResiduals, ApplyUpdate
//   /root/reference/include/ba/BundleAdjuster.h      Init / Add* / RegularizePose
//   /root/reference/include/ba/parallel_algos.h      projection + inertial residuals
//   /root/reference/include/ba/Types.h               data model, IMU integrator (oimu.h)
//   /root/reference/include/ba/Utils.h               Lie-group helpers (outils.h)
//   /root/reference/include/ba/SparseBlockMatrixOps.h  semantics of the block products
//                                                    (strides, upper-triangular rule)
// Every function cites the reference lines it follows.  The block-sparse containers
// are not re-created: their results (U, W, V^-1, S, rhs) are accumulated directly,
// so only the floating-point summation order differs from the reference.
//
// PARITY UNPINNED (see ba_oracle.h): no golden vectors exist in the reference; the
// restatement is pinned by FD-Jacobian and dense-algebra tests under tests/.
//
// Nothing here is used by the product path.
#include "ba_oracle.h"
#include "oimu.h"
#include <chrono>
#include <climits>
#include <cfloat>

using namespace orc;

namespace {

const int kTrustRegionAuto = -1;  // BundleAdjuster.h:30

enum OptimizationResult {  // BundleAdjuster.h:38-46
  Success, ErrorIncreased, ErrorChangeBelowThreshold, ParamChangeBelowThreshold,
  FactorizationError, SolverError
};

double now_s() {
  return std::chrono::duration<double>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

typedef Mat<15, 15> Mat15;
typedef Mat<15, 1> Vec15;

struct Camera { Pinhole model; SE3 t_vs; };

struct Pose {  // Types.h:41-71
  SE3 t_wp; Vec3 v_w; Vec6 b;
  std::vector<bool> param_mask;
  bool is_param_mask_used = false, is_active = true;
  uint32_t id = 0, opt_id = 0;
  double time = -1;
  double cam_params[4] = {0, 0, 0, 0};  // Types.h:46 (pinhole fx, fy, u0, v0); used with use_per_pose_cam_params
  bool has_cam_params = false;
  std::vector<int> proj_residuals, inertial_residuals, binary_residuals,
      unary_residuals, landmarks;
  std::vector<SE3> t_sw;
  const SE3& GetTsw(uint32_t cam_id, const std::vector<Camera>& rig) {
    while (t_sw.size() <= cam_id) t_sw.push_back((t_wp * rig[t_sw.size()].t_vs).inverse());
    return t_sw[cam_id];
  }
};

struct Landmark {  // Types.h:73-89
  Vec2 z_ref; Vec4 x_s, x_w;
  std::vector<int> proj_residuals;
  uint32_t num_outlier_residuals = 0, id = 0, opt_id = 0, ref_pose_id = 0, ref_cam_id = 0;
  bool is_active = true, is_reliable = true;
  Mat3 jtj;  // top-left LmDim x LmDim used
};

struct ProjectionResidual {  // Types.h:282-298
  Vec2 z, residual;
  uint32_t x_meas_id, x_ref_id, landmark_id, cam_id, residual_id, residual_offset;
  Mat<2, 3> dz_dlm;  // first LmDim columns used
  Mat<2, 6> dz_dx_meas, dz_dx_ref;
  Mat<2, 6> dz_dx_meas_raw, dz_dx_ref_raw;  // taps: before column masking
  Mat<2, 6> dz_dtvs;  // Types.h:295, d residual / d T_vs (DoTvs instantiations only)
  Mat<2, 5> dz_dcam_params;  // Types.h:294 (CalibSize instantiations; pinhole: 4 parameters, FOV camera: 5)
  double mahalanobis_distance = 0, weight = 1, orig_weight = 1;
  bool is_conditioning = false;
};

struct UnaryResidual {  // Types.h:255-266
  uint32_t pose_id, residual_id, residual_offset;
  SE3 t_wp;
  Mat<6, 6> dz_dx, cov_inv, cov_inv_sqrt;
  Vec6 residual;
  double mahalanobis_distance = 0, weight = 1, orig_weight = 1;
  bool use_rotation = true;
};

struct BinaryResidual {  // Types.h:268-280
  uint32_t x1_id, x2_id, residual_id, residual_offset;
  SE3 t_12;
  Mat<6, 6> dz_dx1, dz_dx2, cov_inv, cov_inv_sqrt;
  Vec6 residual;
  double mahalanobis_distance = 0, weight = 1, orig_weight = 1;
  bool use_rotation = true;
};

struct ImuResidual {  // Types.h:300-321 (kResSize = PoseSize; 15x15 storage)
  uint32_t pose1_id, pose2_id, residual_id, residual_offset;
  std::vector<ImuMeasurement> measurements;
  Mat15 dz_dx1, dz_dx2, cov_inv, cov_inv_sqrt;
  Mat<15, 6> dz_db;
  Vec15 residual;
  Mat<10, 6> dintegration_db;
  Mat<10, 10> c_integration;
  bool covariance_computed = false;  // Types.h:321
  double mahalanobis_distance = 0, weight = 1, orig_weight = 1;
};

struct Delta { std::vector<double> delta_p, delta_l, delta_k; };  // BundleAdjuster.h:157-162

double sq_norm(const std::vector<double>& v) {
  double s = 0;
  for (double x : v) s += x * x;
  return s;
}

SE3 se3_from7(const double* p) {
  SE3 t;
  t.t[0] = p[0]; t.t[1] = p[1]; t.t[2] = p[2];
  t.r = SO3::raw(Quat(p[3], p[4], p[5], p[6]));
  return t;
}
void se3_to7(const SE3& t, double* p) {
  p[0] = t.t[0]; p[1] = t.t[1]; p[2] = t.t[2];
  p[3] = t.r.q.x; p[4] = t.r.q.y; p[5] = t.r.q.z; p[6] = t.r.q.w;
}

// LDL^T (no pivoting) of the symmetric matrix whose UPPER triangle is given in the
// row-major n x n array s, then solve.  Stands in for Eigen::LDLT<Upper> /
// SimplicialLDLT<Upper> (BundleAdjuster.cpp:752-799): both read the upper triangle
// only; any exact factorisation agrees to rounding for the SPD systems of this path.
// Blocked right-looking so the CPU baseline is not needlessly slow.
// g_ldlt_threads > 1: the "best-effort CPU" mode of BASELINE.md §3 (ii) — the rows of a trailing
// update (97 % of the flops) are independent, so they are dealt to OpenMP threads; every element is
// computed by the same instruction sequence as with one thread (bitwise identical results).
// 1 (default) is the reference-faithful mode: Eigen's LDLT is single-threaded.
static int g_ldlt_threads = 1;
bool ldlt_solve_upper(uint32_t n, const double* s, const double* rhs, double* x) {
  if (n == 0) return true;
  const int nth = g_ldlt_threads;
  (void)nth;
  std::vector<double> A((size_t)n * n);  // lower, row-major: A[i][j] = s[j][i], j<=i
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j <= i; ++j) A[(size_t)i * n + j] = s[(size_t)j * n + i];
  std::vector<double> d(n);
  const uint32_t nb = 64;
  std::vector<double> Wp;  // W = L21 * D for the current panel, row-major m x nb
  bool ok = true;
  for (uint32_t k0 = 0; k0 < n; k0 += nb) {
    const uint32_t kb = std::min(nb, n - k0);
    // unblocked LDL^T of the diagonal block, applied to the whole panel below
    for (uint32_t k = k0; k < k0 + kb; ++k) {
      double dk = A[(size_t)k * n + k];
      for (uint32_t p = k0; p < k; ++p) {
        const double l = A[(size_t)k * n + p];
        dk -= l * l * d[p];
      }
      d[k] = dk;
      if (dk == 0.0 || std::isnan(dk)) ok = false;
      const double inv = 1.0 / dk;
      for (uint32_t i = k + 1; i < n; ++i) {
        double v = A[(size_t)i * n + k];
        const double* li = &A[(size_t)i * n + k0];
        const double* lk = &A[(size_t)k * n + k0];
        for (uint32_t p = 0; p < k - k0; ++p) v -= li[p] * lk[p] * d[k0 + p];
        A[(size_t)i * n + k] = v * inv;
      }
    }
    // trailing update A22 -= L21 D L21^T (lower part)
    const uint32_t r0 = k0 + kb;
    if (r0 >= n) break;
    const uint32_t m = n - r0;
    Wp.assign((size_t)m * kb, 0.0);
    for (uint32_t i = 0; i < m; ++i)
      for (uint32_t p = 0; p < kb; ++p)
        Wp[(size_t)i * kb + p] = A[(size_t)(r0 + i) * n + k0 + p] * d[k0 + p];
#pragma omp parallel for schedule(dynamic, 16) num_threads(nth) if (nth > 1)
    for (uint32_t i = 0; i < m; ++i) {
      const double* li = &A[(size_t)(r0 + i) * n + k0];
      double* ai = &A[(size_t)(r0 + i) * n + r0];
      uint32_t j = 0;
      for (; j + 4 <= i + 1; j += 4) {
        const double* w0 = &Wp[(size_t)(j + 0) * kb];
        const double* w1 = &Wp[(size_t)(j + 1) * kb];
        const double* w2 = &Wp[(size_t)(j + 2) * kb];
        const double* w3 = &Wp[(size_t)(j + 3) * kb];
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (uint32_t p = 0; p < kb; ++p) {
          const double l = li[p];
          s0 += l * w0[p]; s1 += l * w1[p]; s2 += l * w2[p]; s3 += l * w3[p];
        }
        ai[j] -= s0; ai[j + 1] -= s1; ai[j + 2] -= s2; ai[j + 3] -= s3;
      }
      for (; j <= i; ++j) {
        const double* w0 = &Wp[(size_t)j * kb];
        double s0 = 0;
        for (uint32_t p = 0; p < kb; ++p) s0 += li[p] * w0[p];
        ai[j] -= s0;
      }
    }
  }
  // solve L y = b ; z = D^-1 y ; L^T x = z
  for (uint32_t i = 0; i < n; ++i) {
    double v = rhs[i];
    const double* li = &A[(size_t)i * n];
    for (uint32_t j = 0; j < i; ++j) v -= li[j] * x[j];
    x[i] = v;
  }
  for (uint32_t i = 0; i < n; ++i) x[i] /= d[i];
  for (uint32_t ii = n; ii-- > 0;) {
    const double xi = x[ii];
    const double* li = &A[(size_t)ii * n];
    for (uint32_t j = 0; j < ii; ++j) x[j] -= li[j] * xi;
  }
  for (uint32_t i = 0; i < n; ++i)
    if (std::isnan(x[i]) || std::isinf(x[i])) ok = false;
  return ok;
}

}  // namespace

// ---------------------------------------------------------------------------
struct orc_ba {
  const int kLmDim, kPoseDim;
  const bool kVelInState, kBiasInState;
  // BundleAdjuster.h:121-134: kCalibDim = CalibSize + (DoTvs ? 6 : 0), kTvsOffset = CalibSize.
  // CalibSize: 0, or 4 = the parameters (fx, fy, u0, v0) of the pinhole model this restatement has
  // (the reference instantiates 5 for Calibu's FOV camera, which is not in its tree).  Both at once
  // is refused: the reference's T_vs block does setZero() on the j_kpr_ entry it shares with the
  // intrinsics (BundleAdjuster.cpp:1775-1783 after :1759-1766) and so wipes them.
  const bool kTvsInCalib;
  const int kCamParamsDim;   // CalibSize
  const bool kCamParamsInCalib;
  const int kCalibDim;
  // the calibration Jacobian of one residual as the reference lays j_kpr_ out: intrinsics first,
  // T_vs at kTvsOffset = CalibSize
  double jk(const ProjectionResidual& res, int r, int c) const {
    return c < kCamParamsDim ? res.dz_dcam_params(r, c) : res.dz_dtvs(r, c - kCamParamsDim);
  }
  uint32_t num_total_params() const { return num_active_poses_ * kPoseDim + kCalibDim; }
  static const int kPrPoseDim = 6;
  int imu_res_size() const { return kPoseDim; }  // ImuResidualT<S, kPoseDim, kPoseDim>

  orc_options options_;
  orc_summary summary_;
  orc_timers timers_;
  double trust_region_size_ = kTrustRegionAuto;
  uint32_t root_pose_id_ = 0, num_active_poses_ = 0, num_active_landmarks_ = 0;
  uint32_t proj_residual_offset = 0, binary_residual_offset_ = 0,
           unary_residual_offset_ = 0, inertial_residual_offset_ = 0;
  bool is_param_mask_used_ = false;
  bool use_per_pose_cam_params_ = false;  // Options::use_per_pose_cam_params (BundleAdjuster.h:96)
  bool calculate_inertial_covariance_once_ = false;  // Options::calculate_inertial_covariance_once (:106)
  // parallel_algos.h:54-57, BundleAdjuster.cpp:162-165: the camera model takes the intrinsics of
  // the MEASUREMENT pose for the duration of one residual (and is restored afterwards)
  Pinhole cam_for(const ProjectionResidual& res) const {
    Pinhole c = rig_[res.cam_id].model;
    if (use_per_pose_cam_params_) {
      const Pose& p = poses_[res.x_meas_id];
      c.fx = p.cam_params[0]; c.fy = p.cam_params[1]; c.u0 = p.cam_params[2]; c.v0 = p.cam_params[3];
    }
    return c;
  }
  double proj_error_ = 0, binary_error_ = 0, unary_error_ = 0, inertial_error_ = 0;

  Vec3 g_vec;     // imu_.g_vec
  Vec6 imu_r;     // imu_.r diagonal
  Vec6 imu_r_b;   // imu_.r_b
  std::vector<Camera> rig_;
  std::vector<Pose> poses_;
  std::vector<Landmark> landmarks_;
  std::vector<ProjectionResidual> proj_residuals_;
  std::vector<UnaryResidual> unary_residuals_;
  std::vector<BinaryResidual> binary_residuals_;
  std::vector<ImuResidual> inertial_residuals_;
  std::vector<uint32_t> conditioning_proj_residuals_, conditioning_inertial_residuals_;

  // linear system of the current iteration
  std::vector<double> r_pr_, r_pp_, r_u_, r_i_;
  std::vector<double> rhs_p_, rhs_l_, rhs_p_sc_, rhs_k_;
  std::vector<double> s_;                 // dense (n + kCalibDim)^2, row-major
  std::vector<double> jt_l_j_kpr_;        // per active landmark LmDim x kCalibDim (:532-536)
  std::vector<double> vi_;                // per active landmark LmDim x LmDim
  // W = jt_pr * j_l: per active landmark, list of (pose opt_id, 6 x LmDim block)
  struct WBlock { uint32_t pose; Mat<6, 3> w; };
  std::vector<std::vector<WBlock>> w_;
  Delta last_delta_;

  orc_ba(int lm, int pd, bool do_tvs = false, int calib_size = 0)
      : kLmDim(lm), kPoseDim(pd), kVelInState(pd >= 9), kBiasInState(pd >= 15),
        kTvsInCalib(do_tvs), kCamParamsDim(calib_size), kCamParamsInCalib(calib_size > 0),
        kCalibDim(calib_size + (do_tvs ? 6 : 0)) {
    orc_default_options(&options_);
    memset(&summary_, 0, sizeof(summary_));
    memset(&timers_, 0, sizeof(timers_));
    // ImuCalibrationT ctor (Types.h:114-138): g = (0,0) -> g_vec = GetGravityVector
    g_vec[0] = 0; g_vec[1] = 0; g_vec[2] = -9.8007;
    Init(options_);
  }

  // BundleAdjuster.h:177-237
  void Init(const orc_options& o) {
    options_ = o;
    trust_region_size_ = options_.trust_region_size;
    root_pose_id_ = 0;
    num_active_poses_ = num_active_landmarks_ = 0;
    proj_residual_offset = binary_residual_offset_ = unary_residual_offset_ =
        inertial_residual_offset_ = 0;
    for (int i = 0; i < 3; ++i) {
      imu_r[i] = powi(o.gyro_sigma, 2);
      imu_r[3 + i] = powi(o.accel_sigma, 2);
      imu_r_b[i] = powi(o.gyro_bias_sigma, 2);
      imu_r_b[3 + i] = powi(o.accel_bias_sigma, 2);
    }
    rig_.clear(); poses_.clear(); proj_residuals_.clear(); binary_residuals_.clear();
    unary_residuals_.clear(); inertial_residuals_.clear(); landmarks_.clear();
    conditioning_inertial_residuals_.clear(); conditioning_proj_residuals_.clear();
  }

  // BundleAdjuster.h:292-323
  uint32_t AddPose(const SE3& t_wv, const Vec3& v_w, const Vec6& b, bool is_active,
                   double time) {
    Pose pose;
    pose.time = time; pose.t_wp = t_wv; pose.v_w = v_w; pose.b = b;
    pose.is_active = is_active; pose.is_param_mask_used = false;
    pose.id = poses_.size();
    if (is_active) { pose.opt_id = num_active_poses_; num_active_poses_++; }
    else pose.opt_id = UINT_MAX;
    poses_.push_back(pose);
    return pose.id;
  }

  // BundleAdjuster.h:326-367
  uint32_t AddLandmark(const Vec4& x_w, uint32_t ref_pose_id, uint32_t ref_cam_id,
                       bool is_active) {
    assert(ref_pose_id < poses_.size());
    Landmark lm;
    lm.x_w = x_w; lm.ref_pose_id = ref_pose_id; lm.ref_cam_id = ref_cam_id;
    lm.is_active = is_active; lm.is_reliable = true; lm.id = landmarks_.size();
    poses_[ref_pose_id].landmarks.push_back(lm.id);
    if (is_active) { lm.opt_id = num_active_landmarks_; num_active_landmarks_++; }
    else lm.opt_id = UINT_MAX;
    landmarks_.push_back(lm);
    return lm.id;
  }

  // BundleAdjuster.h:377-407
  uint32_t AddUnaryConstraint(uint32_t pose_id, const SE3& t_wv, Mat<6, 6> covariance,
                              bool use_rotation) {
    assert(pose_id < poses_.size());
    UnaryResidual r;
    r.orig_weight = 1.0; r.pose_id = pose_id; r.residual_id = unary_residuals_.size();
    r.residual_offset = unary_residual_offset_; r.t_wp = t_wv; r.use_rotation = use_rotation;
    if (!use_rotation) covariance(3, 3) = covariance(4, 4) = covariance(5, 5) = 1.0;
    r.cov_inv = inverse(covariance);
    r.cov_inv_sqrt = sqrt_spd(r.cov_inv);
    unary_residuals_.push_back(r);
    unary_residual_offset_ += 6;
    poses_[pose_id].unary_residuals.push_back(r.residual_id);
    return r.residual_id;
  }

  // BundleAdjuster.h:425-456
  uint32_t AddBinaryConstraint(uint32_t p1, uint32_t p2, const SE3& t_12,
                               const Mat<6, 6>& covariance, double weight,
                               bool use_rotation) {
    assert(p1 < poses_.size() && p2 < poses_.size());
    BinaryResidual r;
    r.orig_weight = weight; r.x1_id = p1; r.x2_id = p2;
    r.residual_id = binary_residuals_.size(); r.residual_offset = binary_residual_offset_;
    r.t_12 = t_12; r.cov_inv = inverse(covariance); r.cov_inv_sqrt = sqrt_spd(r.cov_inv);
    r.use_rotation = use_rotation;
    binary_residuals_.push_back(r);
    binary_residual_offset_ += 6;
    poses_[p1].binary_residuals.push_back(r.residual_id);
    poses_[p2].binary_residuals.push_back(r.residual_id);
    return r.residual_id;
  }

  // BundleAdjuster.h:459-513
  uint32_t AddProjectionResidual(const Vec2& z, uint32_t meas_pose_id, uint32_t landmark_id,
                                 uint32_t cam_id, double weight) {
    assert(landmark_id < landmarks_.size() && meas_pose_id < poses_.size());
    ProjectionResidual r;
    r.orig_weight = weight; r.landmark_id = landmark_id; r.x_meas_id = meas_pose_id;
    r.x_ref_id = landmarks_[landmark_id].ref_pose_id; r.z = z; r.cam_id = cam_id;
    r.residual_id = proj_residuals_.size(); r.residual_offset = proj_residual_offset;
    Landmark& lm = landmarks_[landmark_id];
    if (meas_pose_id == r.x_ref_id && cam_id == lm.ref_cam_id) lm.z_ref = z;
    const uint32_t res_id = r.residual_id;
    const bool diff_poses = meas_pose_id != r.x_ref_id;
    if (diff_poses || cam_id != lm.ref_cam_id || kLmDim != 1) {
      lm.proj_residuals.push_back(res_id);
      if (diff_poses || kLmDim != 1) {
        poses_[meas_pose_id].proj_residuals.push_back(res_id);
        if (kLmDim == 1) poses_[r.x_ref_id].proj_residuals.push_back(res_id);
      }
    } else {
      return (uint32_t)-1;  // observation from the privileged frame is rejected
    }
    proj_residuals_.push_back(r);
    proj_residual_offset += 2;
    // Quirk Q1 (BundleAdjuster.h:503-510): is_conditioning is set on the local copy
    // AFTER push_back, so the stored flag stays false; only the id list is filled.
    if (!poses_[r.x_ref_id].is_active && poses_[r.x_meas_id].is_active)
      conditioning_proj_residuals_.push_back(r.residual_id);
    return r.residual_id;
  }

  // BundleAdjuster.h:516-546
  uint32_t AddImuResidual(uint32_t p1, uint32_t p2, const std::vector<ImuMeasurement>& m,
                          double weight) {
    assert(p1 < poses_.size() && p2 < poses_.size());
    ImuResidual r;
    r.orig_weight = weight; r.pose1_id = p1; r.pose2_id = p2; r.measurements = m;
    r.residual_id = inertial_residuals_.size(); r.residual_offset = inertial_residual_offset_;
    inertial_residuals_.push_back(r);
    inertial_residual_offset_ += imu_res_size();
    poses_[p1].inertial_residuals.push_back(r.residual_id);
    poses_[p2].inertial_residuals.push_back(r.residual_id);
    if (!poses_[p1].is_active && poses_[p2].is_active)
      conditioning_inertial_residuals_.push_back(r.residual_id);
    return r.residual_id;
  }

  // BundleAdjuster.h:634-652
  uint32_t GetGravityRegularizationDimension(uint32_t pose_id) {
    const Mat3 rot = poses_[pose_id].t_wp.rotationMatrix();
    double max_dot = 0; uint32_t max_dim = 0;
    for (uint32_t i = 0; i < 3; ++i) {
      double dot = 0;
      for (int r = 0; r < 3; ++r) dot += rot(r, i) * g_vec[r];
      dot = std::fabs(dot);
      if (dot > max_dot) { max_dot = dot; max_dim = i; }
    }
    return max_dim + 3;
  }

  // BundleAdjuster.h:608-631 (Quirk Q8: rotation masks indices 2,4,5)
  void RegularizePose(uint32_t pose_id, bool translation, bool gravity, bool bias,
                      bool rotation) {
    Pose& pose = poses_[pose_id];
    pose.is_param_mask_used = true;
    pose.param_mask.assign(kPoseDim, true);
    if (translation) pose.param_mask[0] = pose.param_mask[1] = pose.param_mask[2] = false;
    if (rotation) pose.param_mask[2] = pose.param_mask[4] = pose.param_mask[5] = false;
    if (gravity) pose.param_mask[GetGravityRegularizationDimension(pose_id)] = false;
    if (bias && kBiasInState)
      for (int i = 9; i < 15; ++i) pose.param_mask[i] = false;
  }

  // ------------------------------------------------------------------------
  // parallel_algos.h:35-152  (one observation)
  void ProjectionResidualJacobian(ProjectionResidual& res) {
    Landmark& lm = landmarks_[res.landmark_id];
    Pose& pose = poses_[res.x_meas_id];
    Pose& ref_pose = poses_[res.x_ref_id];
    const Pinhole cam = cam_for(res);
    const SE3& t_vs_m = rig_[res.cam_id].t_vs;
    const SE3& t_vs_r = rig_[lm.ref_cam_id].t_vs;
    const SE3 t_sw_m = pose.GetTsw(res.cam_id, rig_);
    const SE3 t_ws_r = ref_pose.GetTsw(lm.ref_cam_id, rig_).inverse();
    Vec3 xs3, xw3;
    for (int i = 0; i < 3; ++i) { xs3[i] = lm.x_s[i]; xw3[i] = lm.x_w[i]; }

    const Vec2 p = kLmDim == 3 ? cam.Transfer3d(t_sw_m, xw3, lm.x_w[3])
                               : cam.Transfer3d(t_sw_m * t_ws_r, xs3, lm.x_s[3]);
    res.residual = res.z - p;

    const Vec4 x_s_m = kLmDim == 1 ? MultHomogeneous(t_sw_m * t_ws_r, lm.x_s)
                                   : MultHomogeneous(t_sw_m, lm.x_w);
    Vec3 xsm3; for (int i = 0; i < 3; ++i) xsm3[i] = x_s_m[i];
    const Mat<2, 4> dt_dp_m = cam.dTransfer3d_dray(SE3(), xsm3, x_s_m[3]);
    const Mat<2, 4> dt_dp_s = kLmDim == 3 ? dt_dp_m * t_sw_m.matrix()
                                          : dt_dp_m * (t_sw_m * t_ws_r).matrix();
    if (lm.is_active) {
      res.dz_dlm = Mat<2, 3>::Zero();
      for (int c = 0; c < kLmDim; ++c)
        for (int r = 0; r < 2; ++r)
          res.dz_dlm(r, c) = -dt_dp_s(r, (kLmDim == 3 ? 0 : 3) + c);
    }
    // Quirk Q13: the LmDim == 3 branch of the reference is never
    // instantiated and is broken — it reads the never initialised x_s
    // (parallel_algos.h:94) and zeroes the Jacobian when the measurement comes from
    // the (for a world-frame point meaningless) reference pose.  The evident intent is
    // implemented instead: the world point x_w, and no meas==ref special case.
    const bool diff_poses = res.x_ref_id != res.x_meas_id || kLmDim == 3;
    if (pose.is_active || ref_pose.is_active) {
      if (diff_poses) {
        const Vec4 pw = kLmDim == 1 ? (Vec4)(t_ws_r.matrix() * lm.x_s) : lm.x_w;
        res.dz_dx_meas = -(dt_dp_m * dt_x_dt(t_sw_m, pw) * dt1_t2_dt2(t_vs_m.inverse()) *
                           dinv_exp_decoupled_dx(pose.t_wp));
      } else {
        res.dz_dx_meas = Mat<2, 6>::Zero();
      }
      if (kLmDim == 1) {
        if (diff_poses) {
          res.dz_dx_ref = -(dt_dp_m *
                            dt_x_dt(t_sw_m * ref_pose.t_wp, (Vec4)(t_vs_r.matrix() * lm.x_s)) *
                            dt1_t2_dt2(t_sw_m) * dexp_decoupled_dx(ref_pose.t_wp));
        } else {
          res.dz_dx_ref = Mat<2, 6>::Zero();
        }
        if (kCamParamsInCalib)  // parallel_algos.h:114-118: at the pixel z_ref, with x_s(3) as it stands
          res.dz_dcam_params = -cam.dTransfer_dparams(t_sw_m * t_ws_r, lm.z_ref, lm.x_s[3]);
        if (kTvsInCalib) {  // parallel_algos.h:120-131, total derivative of the transfer
          const SE3 t_pm_pr = pose.t_wp.inverse() * ref_pose.t_wp;
          res.dz_dtvs =
              -(dt_dp_m * dt_x_dt(t_sw_m * t_ws_r, lm.x_s) *
                (dt1_t2_dt2(t_vs_m.inverse()) * dt1_t2_dt2(t_pm_pr) * dexp_decoupled_dx(t_vs_r) +
                 dt1_t2_dt1(t_vs_m.inverse(), t_pm_pr * t_vs_r) * dinv_exp_decoupled_dx(t_vs_m)));
        }
      }
    }
    res.dz_dx_meas_raw = res.dz_dx_meas;
    res.dz_dx_ref_raw = res.dz_dx_ref;
    res.weight = res.orig_weight;
    res.mahalanobis_distance = res.residual.squaredNorm() * res.weight;
  }

  // parallel_algos.h:178-358  (one inertial residual)
  void InertialResidualJacobian(ImuResidual& res) {
    const Vec3 gravity = g_vec;
    const Pose& pose1 = poses_[res.pose1_id];
    const Pose& pose2 = poses_[res.pose2_id];
    // parallel_algos.h:189-205: with Options::calculate_inertial_covariance_once the integration
    // covariance and the bias Jacobian of the FIRST linearisation of this residual are kept
    const bool compute_covariance = !calculate_inertial_covariance_once_ || !res.covariance_computed;
    if (compute_covariance) res.c_integration = Mat<10, 10>::Zero();
    ImuPose start; start.t_wp = pose1.t_wp; start.v_w = pose1.v_w; start.time = pose1.time;
    Vec3 bg, ba;
    for (int i = 0; i < 3; ++i) { bg[i] = pose1.b[i]; ba[i] = pose1.b[3 + i]; }
    const ImuPose imu_pose = IntegrateResidual(start, res.measurements, bg, ba, gravity,
                                               compute_covariance ? &res.dintegration_db : nullptr, nullptr,
                                               compute_covariance ? &res.c_integration : nullptr,
                                               compute_covariance ? &imu_r : nullptr);
    res.covariance_computed = true;
    const double total_dt = res.measurements.back().time - res.measurements.front().time;
    const SE3& t_w1 = pose1.t_wp;
    const SE3& t_w2 = pose2.t_wp;
    SE3 t_12_0 = imu_pose.t_wp;
    t_12_0.translation() = t_12_0.translation() -
        (gravity * (-0.5 * powi(total_dt, 2)) + pose1.v_w * total_dt);
    t_12_0 = pose1.t_wp.inverse() * t_12_0;
    Vec3 v_12_0 = imu_pose.v_w - pose1.v_w;
    v_12_0 = v_12_0 + gravity * total_dt;
    v_12_0 = pose1.t_wp.so3().inverse() * v_12_0;

    const int RS = imu_res_size();
    res.residual = Vec15::Zero();
    res.dz_dx1 = Mat15::Zero(); res.dz_dx2 = Mat15::Zero(); res.dz_db = Mat<15, 6>::Zero();

    res.dz_dx1.setBlock<3, 3>(0, 6, Mat3::Identity() * total_dt);
    for (int ii = 0; ii < 3; ++ii) {
      const Vec3 col = t_w1.so3().matrix() * SO3::generator(ii) * v_12_0;
      res.dz_dx1.setBlock<3, 1>(6, 3 + ii, col);
    }
    res.dz_dx1.setBlock<3, 3>(6, 6, Mat3::Identity());
    const Mat<6, 6> b00 = dLog_decoupled_dt1(imu_pose.t_wp, t_w2) *
                          dt1_t2_dt1(t_w1, t_12_0) * dexp_decoupled_dx(t_w1);
    res.dz_dx1.setBlock<6, 6>(0, 0, b00);
    const Mat<6, 6> c00 = dlog_decoupled_dt2(imu_pose.t_wp, t_w2) * dexp_decoupled_dx(t_w2);
    res.dz_dx2.setBlock<6, 6>(0, 0, c00);
    res.dz_dx2.setBlock<3, 3>(6, 6, -Mat3::Identity());

    res.weight = res.orig_weight;
    const Vec6 lg = log_decoupled(imu_pose.t_wp, t_w2);
    for (int i = 0; i < 6; ++i) res.residual[i] = lg[i];
    for (int i = 0; i < 3; ++i) res.residual[6 + i] = imu_pose.v_w[i] - pose2.v_w[i];

    const Mat<6, 7> dlogt1t2_dt1 = dLog_decoupled_dt1(imu_pose.t_wp, t_w2);
    Mat<9, 10> dse3t1t2v_dt1;
    dse3t1t2v_dt1.setBlock<6, 7>(0, 0, dlogt1t2_dt1);
    dse3t1t2v_dt1.setBlock<3, 3>(6, 7, Mat3::Identity());

    Mat15 cov = Mat15::Zero();
    for (int i = 0; i < RS; ++i) cov(i, i) = 1.0;
    if (kBiasInState)
      for (int i = 0; i < 6; ++i) cov(9 + i, 9 + i) = imu_r_b[i] * total_dt;
    const Mat<9, 9> c9 = dse3t1t2v_dt1 * res.c_integration * dse3t1t2v_dt1.T();
    cov.setBlock<9, 9>(0, 0, c9);
    // inverse of the RS x RS leading block
    for (int i = RS; i < 15; ++i) cov(i, i) = 1.0;  // pad: keeps the inverse block-exact
    res.cov_inv = inverse(cov);
    for (int i = RS; i < 15; ++i) res.cov_inv(i, i) = 0.0;

    if (kBiasInState) {
      const Mat<6, 6> tb = dlogt1t2_dt1 * res.dintegration_db.block<7, 6>(0, 0);
      res.dz_db.setBlock<6, 6>(0, 0, tb);
      res.dz_db.setBlock<3, 6>(6, 0, res.dintegration_db.block<3, 6>(7, 0));
      res.dz_db.setBlock<6, 6>(9, 0, Mat<6, 6>::Identity());
      res.dz_dx1.setBlock<15, 6>(0, 9, res.dz_db);
      res.dz_dx2.setBlock<6, 6>(9, 9, -Mat<6, 6>::Identity());
      for (int i = 0; i < 6; ++i) res.residual[9 + i] = pose1.b[i] - pose2.b[i];
    }
    const Vec15 ci = res.cov_inv * res.residual;
    double md = 0;
    for (int i = 0; i < 15; ++i) md += res.residual[i] * ci[i];
    res.mahalanobis_distance = md;
  }

  // std::nth_element at floor(0.5 N): the (upper) median (Quirk Q10).
  static double median_upper(std::vector<double> e) {
    auto it = e.begin() + (size_t)std::floor(e.size() * 0.5);
    std::nth_element(e.begin(), it, e.end());
    return *it;
  }

  // BundleAdjuster.cpp:1166-1803
  void BuildProblem() {
    const uint32_t num_proj_res = proj_residuals_.size();
    const uint32_t num_bin_res = binary_residuals_.size();
    const uint32_t num_un_res = unary_residuals_.size();
    const uint32_t num_im_res = inertial_residuals_.size();
    const int RS = imu_res_size();
    r_pr_.assign((size_t)num_proj_res * 2, 0.0);
    r_pp_.assign((size_t)num_bin_res * 6, 0.0);
    r_u_.assign((size_t)num_un_res * 6, 0.0);
    r_i_.assign((size_t)num_im_res * RS, 0.0);

    is_param_mask_used_ = false;
    // :1240-1259 (Quirk Q7: the loop breaks at the first inactive pose)
    bool are_all_active = true;
    for (Pose& pose : poses_) {
      for (size_t ii = 0; ii < rig_.size(); ++ii) pose.GetTsw(ii, rig_);
      if (!pose.is_active) { are_all_active = false; break; }
      if (pose.proj_residuals.empty() && pose.binary_residuals.empty() &&
          pose.unary_residuals.empty() && pose.inertial_residuals.empty()) {
        pose.is_param_mask_used = true;
        pose.param_mask.assign(kPoseDim, false);
      }
    }
    // :1263-1279
    if (kVelInState) {
      for (Pose& pose : poses_) {
        if (pose.inertial_residuals.empty() && pose.is_active) {
          pose.is_param_mask_used = true;
          pose.param_mask.assign(kPoseDim, true);
          pose.param_mask[6] = pose.param_mask[7] = pose.param_mask[8] = false;
          if (kBiasInState)
            for (int i = 9; i < 15; ++i) pose.param_mask[i] = false;
        }
      }
    }
    // :1285-1330
    if (are_all_active && num_un_res == 0 && options_.enable_auto_regularization) {
      Pose& root_pose = poses_[root_pose_id_];
      root_pose.is_param_mask_used = true;
      root_pose.param_mask.assign(kPoseDim, true);
      root_pose.param_mask[0] = root_pose.param_mask[1] = root_pose.param_mask[2] = false;
      if (kBiasInState && options_.regularize_biases_in_batch)
        for (int i = 9; i < 15; ++i) root_pose.param_mask[i] = false;
      if (!kVelInState) {
        root_pose.param_mask[3] = root_pose.param_mask[4] = root_pose.param_mask[5] = false;
      } else {
        const uint32_t reg_dim = GetGravityRegularizationDimension(root_pose_id_);
        root_pose.param_mask[reg_dim] = false;
      }
    }

    // :1338-1350  projection residuals, serially
    double t0 = now_s();
    proj_error_ = 0;
    std::vector<double> errors;
    errors.reserve(num_proj_res);
    for (ProjectionResidual& res : proj_residuals_) {
      ProjectionResidualJacobian(res);
      errors.push_back(res.mahalanobis_distance);  // is_conditioning is always false (Q1)
    }
    const std::vector<double> proj_errors = errors;
    // :1355-1388  median -> Huber
    if (!errors.empty()) {
      const double sigma = std::sqrt(median_upper(errors));
      const double c_huber = 1.2107 * sigma;
      for (ProjectionResidual& res : proj_residuals_) {
        const double e = std::sqrt(res.mahalanobis_distance);
        const bool use_robust = options_.use_robust_norm_for_proj_residuals;
        const bool is_outlier = e > c_huber;
        res.weight *= (is_outlier && use_robust ? c_huber / e : 1.0);
        res.mahalanobis_distance = res.residual.squaredNorm() * res.weight;
        const double sw = std::sqrt(res.weight);
        r_pr_[res.residual_offset] = res.residual[0] * sw;
        r_pr_[res.residual_offset + 1] = res.residual[1] * sw;
        proj_error_ += res.mahalanobis_distance;
      }
    }
    timers_.j_evaluation_proj += now_s() - t0;

    // :1392-1428  binary residuals
    binary_error_ = 0;
    for (BinaryResidual& res : binary_residuals_) {
      const SE3& t_w1 = poses_[res.x1_id].t_wp;
      const SE3& t_w2 = poses_[res.x2_id].t_wp;
      const SE3 t_1w = t_w1.inverse();
      const SE3 t_12 = t_1w * t_w2;
      res.residual = res.cov_inv_sqrt * log_decoupled(t_12, res.t_12);
      const Mat<6, 7> dlog_dt1 = dLog_decoupled_dt1(t_12, res.t_12);
      res.dz_dx1 = dlog_dt1 * dt1_t2_dt1(t_1w, t_w2) * dinv_exp_decoupled_dx(t_w1);
      res.dz_dx2 = dlog_dt1 * dt1_t2_dt2(t_1w) * dexp_decoupled_dx(t_w2);
      if (!res.use_rotation) {
        for (int i = 3; i < 6; ++i) {
          res.residual[i] = 0;
          for (int c = 0; c < 6; ++c) { res.dz_dx1(i, c) = 0; res.dz_dx2(i, c) = 0; }
        }
      }
      res.weight = res.orig_weight;
      for (int i = 0; i < 6; ++i) r_pp_[res.residual_offset + i] = res.residual[i];
      // Quirk Q5: the whitened residual is weighted by cov_inv once more.
      const Vec6 ci = res.cov_inv * res.residual;
      double md = 0;
      for (int i = 0; i < 6; ++i) md += res.residual[i] * ci[i];
      res.mahalanobis_distance = md;
      binary_error_ += res.mahalanobis_distance * res.weight;
    }

    // :1431-1483  unary residuals
    unary_error_ = 0;
    errors.clear();
    for (UnaryResidual& res : unary_residuals_) {
      const SE3& t_wp = poses_[res.pose_id].t_wp;
      res.dz_dx = dlog_decoupled_dx(t_wp, res.t_wp);
      res.residual = log_decoupled(t_wp, res.t_wp);
      if (!res.use_rotation) {
        for (int i = 3; i < 6; ++i) {
          res.residual[i] = 0;
          for (int c = 0; c < 6; ++c) res.dz_dx(i, c) = 0;
        }
      }
      res.weight = res.orig_weight;
      const Vec6 ci = res.cov_inv * res.residual;
      double md = 0;
      for (int i = 0; i < 6; ++i) md += res.residual[i] * ci[i];
      res.mahalanobis_distance = md;
      errors.push_back(md);
    }
    if (!errors.empty()) {
      const double sigma = std::sqrt(median_upper(errors));
      const double c_huber = 1.2107 * sigma;
      for (UnaryResidual& res : unary_residuals_) {
        const double e = std::sqrt(res.mahalanobis_distance);
        const double weight = (e > c_huber) ? c_huber / e : 1.0;
        res.cov_inv = res.cov_inv * weight;  // Quirk Q4: compounds across iterations
        res.cov_inv_sqrt = sqrt_spd(res.cov_inv);
        const Vec6 std_form = res.cov_inv_sqrt * res.residual;
        for (int i = 0; i < 6; ++i) r_u_[res.residual_offset + i] = std_form[i];
        res.mahalanobis_distance = std_form.squaredNorm();
        unary_error_ += res.mahalanobis_distance;
      }
    }

    // :1486-1541  inertial residuals
    inertial_error_ = 0;
    for (ImuResidual& res : inertial_residuals_) InertialResidualJacobian(res);
    // Quirk Q2 (:1497): the weighting loop is driven by the PROJECTION error list.
    // With no projection residuals the reference skips the loop and then reads an
    // uninitialised cov_inv_sqrt (undefined behaviour).  That case cannot be
    // reproduced; the loop is run whenever inertial residuals exist, which equals
    // the reference's result whenever it is defined.
    errors = proj_errors;
    if (!inertial_residuals_.empty()) {
      double c_huber = 0;
      if (!errors.empty()) c_huber = 1.2107 * std::sqrt(median_upper(errors));
      for (ImuResidual& res : inertial_residuals_) {
        const bool use_robust =
            options_.use_robust_norm_for_inertial_residuals && !errors.empty();
        const bool is_cond = !poses_[res.pose1_id].is_active && poses_[res.pose2_id].is_active;
        const double e = std::sqrt(res.mahalanobis_distance);
        const double weight = ((e > c_huber) && !is_cond && use_robust) ? c_huber / e : 1.0;
        res.cov_inv = res.cov_inv * weight;
        res.cov_inv_sqrt = sqrt_spd(res.cov_inv);
        const Vec15 std_form = res.cov_inv_sqrt * res.residual;
        for (int i = 0; i < RS; ++i) r_i_[res.residual_offset + i] = std_form[i];
        res.mahalanobis_distance = std_form.squaredNorm();
        inertial_error_ += res.mahalanobis_distance;
      }
    }

    // :1552-1802  "insertion": column masking.  The sparse containers are not
    // built; masking is applied in place to the per-residual Jacobians exactly as
    // the reference does through its non-const references (:1620-1629 etc.).
    for (Pose& pose : poses_) {
      if (!pose.is_active) continue;
      if (!pose.is_param_mask_used) continue;
      for (int id : pose.proj_residuals) {
        ProjectionResidual& res = proj_residuals_[id];
        Mat<2, 6>& dz_dx = res.x_meas_id == pose.id ? res.dz_dx_meas : res.dz_dx_ref;
        is_param_mask_used_ = true;
        for (int ii = 0; ii < kPrPoseDim; ++ii)
          if (!pose.param_mask[ii]) dz_dx.setColZero(ii);
      }
      for (int id : pose.binary_residuals) {
        BinaryResidual& res = binary_residuals_[id];
        Mat<6, 6>& dz = res.x1_id == pose.id ? res.dz_dx1 : res.dz_dx2;
        is_param_mask_used_ = true;
        for (int ii = 0; ii < 6; ++ii)
          if (!pose.param_mask[ii]) dz.setColZero(ii);
      }
      for (int id : pose.unary_residuals) {
        UnaryResidual& res = unary_residuals_[id];
        is_param_mask_used_ = true;
        for (int ii = 0; ii < 6; ++ii)
          if (!pose.param_mask[ii]) res.dz_dx.setColZero(ii);
      }
      // inertial: the reference masks a COPY (:1695-1705), so the masked columns
      // are re-derived where the blocks are consumed (imu_block()).
      if (!pose.inertial_residuals.empty()) is_param_mask_used_ = true;
    }
  }

  // Masked inertial Jacobian of `res` w.r.t. `pose` (BundleAdjuster.cpp:1693-1705).
  Mat15 imu_block(const ImuResidual& res, const Pose& pose) const {
    Mat15 dz = res.pose1_id == pose.id ? res.dz_dx1 : res.dz_dx2;
    if (pose.is_param_mask_used)
      for (int ii = 0; ii < kPoseDim; ++ii)
        if (!pose.param_mask[ii]) dz.setColZero(ii);
    return dz;
  }

  // Add a (rows x cols) block into the dense s_ at pose-block (bi,bj), honouring the
  // upper-triangular rule of SparseBlockProduct (SparseBlockMatrixOps.h:236-238):
  // block (i,j) is kept iff i <= j; the mirror block is added too when
  // use_triangular_matrices is off.
  template <int N>
  void s_add_pair(uint32_t bi, uint32_t bj, const Mat<N, N>& blk_ij, double sign) {
    const uint32_t n = num_total_params();
    const int lim = N < kPoseDim ? N : kPoseDim;
    auto put = [&](uint32_t ri, uint32_t cj, bool transpose) {
      for (int r = 0; r < lim; ++r)
        for (int c = 0; c < lim; ++c)
          s_[(size_t)(ri * kPoseDim + r) * n + cj * kPoseDim + c] +=
              sign * (transpose ? blk_ij(c, r) : blk_ij(r, c));
    };
    if (bi <= bj) put(bi, bj, false);
    else if (!options_.use_triangular_matrices) put(bi, bj, false);
    if (bi != bj) {
      // mirror contribution J_j^T J_i = (J_i^T J_j)^T
      if (bj <= bi) put(bj, bi, true);
      else if (!options_.use_triangular_matrices) put(bj, bi, true);
    }
  }

  // BundleAdjuster.cpp:298-636 (one iteration up to, not including, SolveInternal)
  void LineariseAndReduce() {
    double t0 = now_s();
    BuildProblem();
    timers_.build_problem += now_s() - t0;

    const uint32_t num_poses = num_active_poses_;
    const uint32_t n = num_poses * kPoseDim;
    const uint32_t num_lm = num_active_landmarks_;
    const int D = kPoseDim, L = kLmDim, RS = imu_res_size();
    const int K = kCalibDim;
    const uint32_t nt = n + K;  // :316-322
    rhs_p_.assign(n, 0.0);
    rhs_k_.assign(K, 0.0);
    rhs_p_sc_.assign(nt, 0.0);
    s_.assign((size_t)nt * nt, 0.0);
    jt_l_j_kpr_.assign((size_t)num_lm * L * K, 0.0);
    vi_.assign((size_t)num_lm * L * L, 0.0);
    rhs_l_.assign((size_t)num_lm * L, 0.0);
    w_.assign(num_lm, std::vector<WBlock>());

    // ---- :327-406  U = sum J^T J, rhs_p = sum J^T r ------------------------
    t0 = now_s();
    if (!proj_residuals_.empty() && num_poses > 0) {
      for (const ProjectionResidual& res : proj_residuals_) {
        const Landmark& lm = landmarks_[res.landmark_id];
        // Jacobian blocks were inserted only through pose.proj_residuals lists of
        // ACTIVE poses (:1613-1643); a residual is in those lists iff it passed the
        // diff_poses test of AddProjectionResidual.
        const bool listed = (res.x_meas_id != res.x_ref_id) || kLmDim != 1;
        if (!listed) continue;
        const Pose& pm = poses_[res.x_meas_id];
        const Pose& pr = poses_[res.x_ref_id];
        const double w = res.weight;  // sqrt(w) on J and on J^T
        const bool has_m = pm.is_active;
        const bool has_r = (kLmDim == 1) && pr.is_active;
        Vec2 rw;  // r_pr_ entry = r sqrt(w); J^T carries another sqrt(w)
        rw[0] = r_pr_[res.residual_offset]; rw[1] = r_pr_[res.residual_offset + 1];
        const double sw = std::sqrt(w);
        if (has_m) {
          const Mat<6, 2> jt = res.dz_dx_meas.T() * sw;
          const Mat<6, 6> jj = jt * (res.dz_dx_meas * sw);
          s_add_pair<6>(pm.opt_id, pm.opt_id, jj, 1.0);
          const Mat<6, 1> g = jt * rw;
          for (int i = 0; i < 6; ++i) rhs_p_[pm.opt_id * D + i] += g[i];
        }
        if (has_r) {
          const Mat<6, 2> jt = res.dz_dx_ref.T() * sw;
          const Mat<6, 6> jj = jt * (res.dz_dx_ref * sw);
          s_add_pair<6>(pr.opt_id, pr.opt_id, jj, 1.0);
          const Mat<6, 1> g = jt * rw;
          for (int i = 0; i < 6; ++i) rhs_p_[pr.opt_id * D + i] += g[i];
        }
        if (has_m && has_r) {
          const Mat<6, 6> jmr = (res.dz_dx_meas.T() * sw) * (res.dz_dx_ref * sw);
          s_add_pair<6>(pm.opt_id, pr.opt_id, jmr, 1.0);
        }
        (void)lm;
      }
    }
    // binary (:357-371): J = S^-1/2 dz, J^T = dz^T S^-1/2 weight
    for (const BinaryResidual& res : binary_residuals_) {
      const Pose& p1 = poses_[res.x1_id];
      const Pose& p2 = poses_[res.x2_id];
      Vec6 rr; for (int i = 0; i < 6; ++i) rr[i] = r_pp_[res.residual_offset + i];
      const Mat<6, 6> j1 = res.cov_inv_sqrt * res.dz_dx1, j2 = res.cov_inv_sqrt * res.dz_dx2;
      const Mat<6, 6> jt1 = res.dz_dx1.T() * res.cov_inv_sqrt * res.weight;
      const Mat<6, 6> jt2 = res.dz_dx2.T() * res.cov_inv_sqrt * res.weight;
      if (p1.is_active) {
        s_add_pair<6>(p1.opt_id, p1.opt_id, jt1 * j1, 1.0);
        const Vec6 g = jt1 * rr;
        for (int i = 0; i < 6; ++i) rhs_p_[p1.opt_id * D + i] += g[i];
      }
      if (p2.is_active) {
        s_add_pair<6>(p2.opt_id, p2.opt_id, jt2 * j2, 1.0);
        const Vec6 g = jt2 * rr;
        for (int i = 0; i < 6; ++i) rhs_p_[p2.opt_id * D + i] += g[i];
      }
      if (p1.is_active && p2.is_active && res.x1_id != res.x2_id)
        s_add_pair<6>(p1.opt_id, p2.opt_id, jt1 * j2, 1.0);
    }
    // unary (:374-386)
    for (const UnaryResidual& res : unary_residuals_) {
      const Pose& p = poses_[res.pose_id];
      if (!p.is_active) continue;
      Vec6 rr; for (int i = 0; i < 6; ++i) rr[i] = r_u_[res.residual_offset + i];
      const Mat<6, 6> j = res.cov_inv_sqrt * res.dz_dx;
      const Mat<6, 6> jt = res.dz_dx.T() * res.cov_inv_sqrt;
      s_add_pair<6>(p.opt_id, p.opt_id, jt * j, 1.0);
      const Vec6 g = jt * rr;
      for (int i = 0; i < 6; ++i) rhs_p_[p.opt_id * D + i] += g[i];
    }
    // inertial (:389-401)
    for (const ImuResidual& res : inertial_residuals_) {
      const Pose& p1 = poses_[res.pose1_id];
      const Pose& p2 = poses_[res.pose2_id];
      Vec15 rr; for (int i = 0; i < RS; ++i) rr[i] = r_i_[res.residual_offset + i];
      const Mat15 d1 = imu_block(res, p1), d2 = imu_block(res, p2);
      const Mat15 j1 = res.cov_inv_sqrt * d1, j2 = res.cov_inv_sqrt * d2;
      const Mat15 jt1 = d1.T() * res.cov_inv_sqrt, jt2 = d2.T() * res.cov_inv_sqrt;
      if (p1.is_active) {
        s_add_pair<15>(p1.opt_id, p1.opt_id, jt1 * j1, 1.0);
        const Vec15 g = jt1 * rr;
        for (int i = 0; i < D; ++i) rhs_p_[p1.opt_id * D + i] += g[i];
      }
      if (p2.is_active) {
        s_add_pair<15>(p2.opt_id, p2.opt_id, jt2 * j2, 1.0);
        const Vec15 g = jt2 * rr;
        for (int i = 0; i < D; ++i) rhs_p_[p2.opt_id * D + i] += g[i];
      }
      if (p1.is_active && p2.is_active && res.pose1_id != res.pose2_id)
        s_add_pair<15>(p1.opt_id, p2.opt_id, jt1 * j2, 1.0);
    }
    timers_.jtj += now_s() - t0;

    // ---- :408-491  Schur complement ---------------------------------------
    t0 = now_s();
    std::copy(rhs_p_.begin(), rhs_p_.end(), rhs_p_sc_.begin());
    // ---- :493-529  calibration border: S_kk = Jk^T Jk, S_pk = Jp^T Jk, rhs_k = Jk^T r.
    // j_kpr_ holds sqrt(w) dz_dtvs of EVERY residual (:1769-1783); jt_pr only the blocks of
    // listed residuals at active poses.
    if (K > 0) {
      for (const ProjectionResidual& res : proj_residuals_) {
        const double w = res.weight, sw = std::sqrt(w);
        for (int a = 0; a < K; ++a) {
          for (int b = 0; b < K; ++b)
            s_[(size_t)(n + a) * nt + n + b] +=
                (jk(res, 0, a) * jk(res, 0, b) + jk(res, 1, a) * jk(res, 1, b)) * w;
          rhs_k_[a] += jk(res, 0, a) * sw * r_pr_[res.residual_offset] +
                       jk(res, 1, a) * sw * r_pr_[res.residual_offset + 1];
        }
        const bool listed = (res.x_meas_id != res.x_ref_id) || kLmDim != 1;
        if (!listed || num_poses == 0) continue;
        auto border = [&](const Pose& p, const Mat<2, 6>& jp) {
          if (!p.is_active) return;
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < K; ++c) {
              const double v = (jp(0, r) * jk(res, 0, c) + jp(1, r) * jk(res, 1, c)) * w;
              s_[(size_t)(p.opt_id * D + r) * nt + n + c] += v;
              if (!options_.use_triangular_matrices) s_[(size_t)(n + c) * nt + p.opt_id * D + r] += v;
            }
        };
        border(poses_[res.x_meas_id], res.dz_dx_meas);
        if (kLmDim == 1) border(poses_[res.x_ref_id], res.dz_dx_ref);
      }
      for (int a = 0; a < K; ++a) rhs_p_sc_[n + a] = rhs_k_[a];  // :526-528
    }
    if (L > 0 && num_lm > 0) {
      for (Landmark& lm : landmarks_) {
        if (!lm.is_active) continue;
        lm.jtj = Mat3::Zero();
        Vec3 jtr_l;
        for (int id : lm.proj_residuals) {
          const ProjectionResidual& res = proj_residuals_[id];
          const double sw = std::sqrt(res.weight);
          for (int a = 0; a < L; ++a) {
            for (int b = 0; b < L; ++b)
              lm.jtj(a, b) += (res.dz_dlm(0, a) * res.dz_dlm(0, b) +
                               res.dz_dlm(1, a) * res.dz_dlm(1, b)) * res.weight;
            jtr_l[a] += res.dz_dlm(0, a) * sw * r_pr_[res.residual_offset] +
                        res.dz_dlm(1, a) * sw * r_pr_[res.residual_offset + 1];
          }
        }
        for (int a = 0; a < L; ++a) rhs_l_[lm.opt_id * L + a] = jtr_l[a];
        if (K > 0)  // jt_kpr_ * j_l_ (:534-536), transposed: L x K per landmark
          for (int id : lm.proj_residuals) {
            const ProjectionResidual& res = proj_residuals_[id];
            for (int a = 0; a < L; ++a)
              for (int c = 0; c < K; ++c)
                jt_l_j_kpr_[((size_t)lm.opt_id * L + a) * K + c] +=
                    (res.dz_dlm(0, a) * jk(res, 0, c) + res.dz_dlm(1, a) * jk(res, 1, c)) * res.weight;
          }
        // Quirk Q11 (:431-440)
        if (L == 1) {
          if (std::fabs(lm.jtj(0, 0)) < 1e-6) lm.jtj(0, 0) += 1e-6;
          vi_[lm.opt_id] = 1.0 / lm.jtj(0, 0);
        } else {
          if (lm.jtj.norm() < 1e-6)
            for (int a = 0; a < 3; ++a) lm.jtj(a, a) += 1e-6;
          const Mat3 inv = inverse(lm.jtj);
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) vi_[(size_t)lm.opt_id * 9 + a * 3 + b] = inv(a, b);
        }
      }
      if (num_poses > 0) {
        // W = jt_pr * j_l (:452): per landmark, one 6 x L block per incident ACTIVE pose
        for (const Landmark& lm : landmarks_) {
          if (!lm.is_active) continue;
          std::vector<WBlock>& wl = w_[lm.opt_id];
          auto acc = [&](uint32_t opt_id, const Mat<2, 6>& jp, const ProjectionResidual& res) {
            WBlock* wb = nullptr;
            for (WBlock& b : wl) if (b.pose == opt_id) { wb = &b; break; }
            if (!wb) { wl.push_back(WBlock()); wb = &wl.back(); wb->pose = opt_id; }
            for (int r = 0; r < 6; ++r)
              for (int c = 0; c < L; ++c)
                wb->w(r, c) += (jp(0, r) * res.dz_dlm(0, c) + jp(1, r) * res.dz_dlm(1, c)) *
                               res.weight;
          };
          for (int id : lm.proj_residuals) {
            const ProjectionResidual& res = proj_residuals_[id];
            const bool listed = (res.x_meas_id != res.x_ref_id) || kLmDim != 1;
            if (!listed) continue;
            if (poses_[res.x_meas_id].is_active)
              acc(poses_[res.x_meas_id].opt_id, res.dz_dx_meas, res);
            if (kLmDim == 1 && poses_[res.x_ref_id].is_active)
              acc(poses_[res.x_ref_id].opt_id, res.dz_dx_ref, res);
          }
          std::sort(wl.begin(), wl.end(),
                    [](const WBlock& a, const WBlock& b) { return a.pose < b.pose; });
          // W V^-1 (:460), W V^-1 W^T (:468-470), S = U - ... (:473-477),
          // rhs_p_sc = rhs_p - W V^-1 rhs_l (:480-484)
          std::vector<Mat<6, 3>> wvi(wl.size());
          for (size_t a = 0; a < wl.size(); ++a) {
            for (int r = 0; r < 6; ++r)
              for (int c = 0; c < L; ++c) {
                double s = 0;
                for (int k = 0; k < L; ++k)
                  s += wl[a].w(r, k) * vi_[(size_t)lm.opt_id * L * L + k * L + c];
                wvi[a](r, c) = s;
              }
            for (int r = 0; r < 6; ++r) {
              double s = 0;
              for (int c = 0; c < L; ++c) s += wvi[a](r, c) * rhs_l_[lm.opt_id * L + c];
              rhs_p_sc_[wl[a].pose * D + r] -= s;
            }
          }
          for (size_t a = 0; a < wl.size(); ++a)
            for (size_t b = 0; b < wl.size(); ++b) {
              if (wl[a].pose > wl[b].pose && options_.use_triangular_matrices) continue;
              Mat<6, 6> blk;
              for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                  double s = 0;
                  for (int k = 0; k < L; ++k) s += wvi[a](r, k) * wl[b].w(c, k);
                  blk(r, c) = s;
                }
              const uint32_t nn = nt;
              for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c)
                  s_[(size_t)(wl[a].pose * D + r) * nn + wl[b].pose * D + c] -= blk(r, c);
            }
          // :538-556  S_pk -= (W V^-1) (Jl^T Jk)
          for (size_t a = 0; a < wl.size() && K > 0; ++a)
            for (int r = 0; r < 6; ++r)
              for (int c = 0; c < K; ++c) {
                double s = 0;
                for (int k = 0; k < L; ++k)
                  s += wvi[a](r, k) * jt_l_j_kpr_[((size_t)lm.opt_id * L + k) * K + c];
                s_[(size_t)(wl[a].pose * D + r) * nt + n + c] -= s;
                if (!options_.use_triangular_matrices)
                  s_[(size_t)(n + c) * nt + wl[a].pose * D + r] -= s;
              }
        }
      }
      // :558-582  S_kk -= (Jk^T Jl) V^-1 (Jl^T Jk), rhs_k_sc = rhs_k - (Jk^T Jl) V^-1 b_l
      if (K > 0) {
        for (const Landmark& lm : landmarks_) {
          if (!lm.is_active) continue;
          const double* e = &jt_l_j_kpr_[(size_t)lm.opt_id * L * K];
          for (int a = 0; a < K; ++a) {
            double ev[3] = {0, 0, 0};  // row a of (Jk^T Jl) V^-1
            for (int c = 0; c < L; ++c)
              for (int k = 0; k < L; ++k)
                ev[c] += e[k * K + a] * vi_[(size_t)lm.opt_id * L * L + k * L + c];
            for (int b = 0; b < K; ++b) {
              double s = 0;
              for (int c = 0; c < L; ++c) s += ev[c] * e[c * K + b];
              s_[(size_t)(n + a) * nt + n + b] -= s;
            }
            double s = 0;
            for (int c = 0; c < L; ++c) s += ev[c] * rhs_l_[lm.opt_id * L + c];
            rhs_p_sc_[n + a] -= s;
          }
        }
      }
    }
    timers_.schur_complement += now_s() - t0;

    // ---- :587-598  masked parameters (Quirk Q12: overwrite, not add) --------
    if (is_param_mask_used_) {
      for (const Pose& pose : poses_) {
        if (pose.is_active && pose.is_param_mask_used) {
          for (uint32_t ii = 0; ii < pose.param_mask.size(); ++ii) {
            if (!pose.param_mask[ii]) {
              const size_t idx = (size_t)pose.opt_id * D + ii;
              s_[idx * nt + idx] = 1e6;
            }
          }
        }
      }
    }
  }

  // BundleAdjuster.cpp:748-833
  void CalculateGn(const std::vector<double>& rhs_p, Delta& delta) {
    summary_.result = Success;
    const uint32_t n = rhs_p.size();
    delta.delta_p.assign(n, 0.0);
    delta.delta_k.clear();
    if (n == 0) return;
    double t0 = now_s();
    const bool ok = ldlt_solve_upper(n, s_.data(), rhs_p.data(), delta.delta_p.data());
    timers_.solve += now_s() - t0;
    if (!ok) summary_.result = FactorizationError;
    if (kCalibDim) {  // :766-769: the solution is [delta_p ; delta_k]
      delta.delta_k.assign(delta.delta_p.end() - kCalibDim, delta.delta_p.end());
      delta.delta_p.resize(n - kCalibDim);
    }
  }

  // BundleAdjuster.cpp:709-744
  void GetLandmarkDelta(const Delta& delta, uint32_t num_poses, uint32_t num_lm,
                        std::vector<double>& delta_l) {
    double t0 = now_s();
    const int L = kLmDim, D = kPoseDim;
    if (num_lm > 0 && L > 0) {
      delta_l.assign((size_t)num_lm * L, 0.0);
      std::vector<double> rhs_l_sc = rhs_l_;
      if (num_poses > 0) {
        for (uint32_t l = 0; l < num_lm; ++l)
          for (const WBlock& b : w_[l])
            for (int c = 0; c < L; ++c) {
              double s = 0;
              for (int r = 0; r < 6; ++r) s += b.w(r, c) * delta.delta_p[b.pose * D + r];
              rhs_l_sc[l * L + c] -= s;
            }
        if (kCalibDim && !delta.delta_k.empty())  // :729-733
          for (uint32_t l = 0; l < num_lm; ++l)
            for (int c = 0; c < L; ++c) {
              double s = 0;
              for (int k = 0; k < kCalibDim; ++k)
                s += jt_l_j_kpr_[((size_t)l * L + c) * kCalibDim + k] * delta.delta_k[k];
              rhs_l_sc[l * L + c] -= s;
            }
      }
      for (uint32_t l = 0; l < num_lm; ++l)
        for (int a = 0; a < L; ++a) {
          double s = 0;
          for (int b = 0; b < L; ++b)
            s += vi_[(size_t)l * L * L + a * L + b] * rhs_l_sc[l * L + b];
          delta_l[l * L + a] = s;
        }
    } else {
      delta_l.clear();
    }
    timers_.back_substitution += now_s() - t0;
  }

  // BundleAdjuster.cpp:21-140
  void ApplyUpdate(const Delta& delta, bool do_rollback, double damping = 1.0) {
    double t0 = now_s();
    // Quirk Q6: sum of the two norms
    summary_.delta_norm = std::sqrt(sq_norm(delta.delta_l)) + std::sqrt(sq_norm(delta.delta_p));
    const double coef = (do_rollback ? -1.0 : 1.0) * damping;
    const int D = kPoseDim, L = kLmDim;
    // :72-83: the extrinsics of camera 0 take -delta_k; no `coef` here (neither the rollback sign
    // nor the damping reaches T_vs), and the copies SolveInternal restores on a rejected step do
    // not include the rig — a rejected step's T_vs update stays.
    // :46-69: the intrinsics of camera 0 take -delta_k (no `coef` either); with inverse-depth
    // landmarks every x_s ray is re-derived from its reference pixel, keeping its length.
    if (kCamParamsInCalib && !delta.delta_k.empty() && !rig_.empty()) {
      Pinhole& m = rig_[0].model;
      m.fx -= delta.delta_k[0]; m.fy -= delta.delta_k[1]; m.u0 -= delta.delta_k[2]; m.v0 -= delta.delta_k[3];
      if (kCamParamsDim == 5) m.w -= delta.delta_k[4];
      if (kLmDim == 1)
        for (Landmark& lm : landmarks_) {
          const double norm = std::sqrt(lm.x_s[0] * lm.x_s[0] + lm.x_s[1] * lm.x_s[1] + lm.x_s[2] * lm.x_s[2]);
          Vec3 ray = m.Unproject(lm.z_ref);
          const double len = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
          for (int i = 0; i < 3; ++i) lm.x_s[i] = ray[i] / len * norm;
        }
    }
    if (kTvsInCalib && !delta.delta_k.empty() && !rig_.empty()) {
      Vec6 d;
      for (int i = 0; i < 6; ++i) d[i] = -delta.delta_k[kCamParamsDim + i];
      rig_[0].t_vs = exp_decoupled(rig_[0].t_vs, d);
    }
    for (Pose& pose : poses_) {
      if (pose.is_active) {
        const uint32_t p_offset = pose.opt_id * D;
        Vec6 p_update;
        for (int i = 0; i < 6; ++i) p_update[i] = -delta.delta_p[p_offset + i] * coef;
        pose.t_wp = exp_decoupled(pose.t_wp, p_update);
        if (kVelInState)
          for (int i = 0; i < 3; ++i) pose.v_w[i] -= delta.delta_p[p_offset + 6 + i] * coef;
        if (kBiasInState)
          for (int i = 0; i < 6; ++i) pose.b[i] -= delta.delta_p[p_offset + 9 + i] * coef;
      }
      pose.t_sw.clear();
    }
    for (Landmark& lm : landmarks_) {
      if (!lm.is_active) continue;
      if (L == 1) {
        const double lm_delta = delta.delta_l[lm.opt_id] * coef;
        lm.x_s[3] -= lm_delta;
        if (lm.x_s[3] < 0) {  // Quirk Q9
          lm.x_s[3] += lm_delta;
          lm.is_reliable = false;
        }
      } else if (L == 3) {
        for (int a = 0; a < 3; ++a) lm.x_w[a] -= delta.delta_l[lm.opt_id * 3 + a] * coef;
      }
    }
    timers_.apply_update += now_s() - t0;
  }

  // BundleAdjuster.cpp:144-274
  void EvaluateResiduals(double* proj_error, double* binary_error, double* unary_error,
                         double* inertial_error) {
    double t0 = now_s();
    if (proj_error) {
      for (Landmark& lm : landmarks_) lm.num_outlier_residuals = 0;
      *proj_error = 0;
      for (ProjectionResidual& res : proj_residuals_) {
        Landmark& lm = landmarks_[res.landmark_id];
        Pose& pose = poses_[res.x_meas_id];
        Pose& ref_pose = poses_[res.x_ref_id];
        const SE3 t_sw_m = pose.GetTsw(res.cam_id, rig_);
        const SE3 t_ws_r = ref_pose.GetTsw(lm.ref_cam_id, rig_).inverse();
        const Pinhole cam = cam_for(res);
        Vec3 xs3, xw3;
        for (int i = 0; i < 3; ++i) { xs3[i] = lm.x_s[i]; xw3[i] = lm.x_w[i]; }
        const Vec2 p = kLmDim == 3 ? cam.Transfer3d(t_sw_m, xw3, lm.x_w[3])
                                   : cam.Transfer3d(t_sw_m * t_ws_r, xs3, lm.x_s[3]);
        res.residual = res.z - p;
        res.mahalanobis_distance = res.residual.squaredNorm() * res.weight;
        *proj_error += res.mahalanobis_distance;
        if (res.residual.norm() > options_.projection_outlier_threshold)
          lm.num_outlier_residuals++;
      }
    }
    if (unary_error) {
      *unary_error = 0;
      for (UnaryResidual& res : unary_residuals_) {
        res.residual = log_decoupled(poses_[res.pose_id].t_wp, res.t_wp);
        if (!res.use_rotation) for (int i = 3; i < 6; ++i) res.residual[i] = 0;
        const Vec6 ci = res.cov_inv * res.residual;
        double md = 0;
        for (int i = 0; i < 6; ++i) md += res.residual[i] * ci[i];
        res.mahalanobis_distance = md;
        *unary_error += md;
      }
    }
    if (binary_error) {
      *binary_error = 0;
      for (BinaryResidual& res : binary_residuals_) {
        res.residual = log_decoupled(poses_[res.x1_id].t_wp.inverse() * poses_[res.x2_id].t_wp,
                                     res.t_12);
        if (!res.use_rotation) for (int i = 3; i < 6; ++i) res.residual[i] = 0;
        res.mahalanobis_distance = res.residual.squaredNorm() * res.weight;
        *binary_error += res.mahalanobis_distance;
      }
    }
    if (inertial_error) {
      *inertial_error = 0;
      for (ImuResidual& res : inertial_residuals_) {
        const Pose& pose1 = poses_[res.pose1_id];
        const Pose& pose2 = poses_[res.pose2_id];
        ImuPose start; start.t_wp = pose1.t_wp; start.v_w = pose1.v_w; start.time = pose1.time;
        Vec3 bg, ba;
        for (int i = 0; i < 3; ++i) { bg[i] = pose1.b[i]; ba[i] = pose1.b[3 + i]; }
        const ImuPose imu_pose = IntegrateResidual(start, res.measurements, bg, ba, g_vec);
        res.residual = Vec15::Zero();
        const Vec6 lg = log_decoupled(imu_pose.t_wp, pose2.t_wp);
        for (int i = 0; i < 6; ++i) res.residual[i] = lg[i];
        for (int i = 0; i < 3; ++i) res.residual[6 + i] = imu_pose.v_w[i] - pose2.v_w[i];
        if (kBiasInState)
          for (int i = 0; i < 6; ++i) res.residual[9 + i] = pose1.b[i] - pose2.b[i];
        const Vec15 ci = res.cov_inv * res.residual;
        double md = 0;
        for (int i = 0; i < 15; ++i) md += res.residual[i] * ci[i];
        res.mahalanobis_distance = md;
        *inertial_error += md;
      }
    }
    timers_.evaluate_residuals += now_s() - t0;
  }

  // J * g products of the dogleg steepest-descent step (BundleAdjuster.cpp:858-925).
  double SteepestDescentDenominator() {
    const int D = kPoseDim, L = kLmDim, RS = imu_res_size();
    double denom = 0;
    // (j_p_rhs_p + j_l_rhs_l).squaredNorm()
    for (const ProjectionResidual& res : proj_residuals_) {
      double v[2] = {0, 0};
      const bool listed = (res.x_meas_id != res.x_ref_id) || kLmDim != 1;
      const double sw = std::sqrt(res.weight);
      if (kCalibDim && num_active_poses_ > 0) {  // j_kp_rhs_k (:883-886): its own squared norm
        double u[2] = {0, 0};
        for (int r = 0; r < 2; ++r)
          for (int c = 0; c < kCalibDim; ++c) u[r] += jk(res, r, c) * sw * rhs_k_[c];
        denom += u[0] * u[0] + u[1] * u[1];
      }
      if (num_active_poses_ > 0 && listed) {
        const Pose& pm = poses_[res.x_meas_id];
        const Pose& pr = poses_[res.x_ref_id];
        if (pm.is_active)
          for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 6; ++c)
              v[r] += res.dz_dx_meas(r, c) * sw * rhs_p_[pm.opt_id * D + c];
        if (kLmDim == 1 && pr.is_active)
          for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 6; ++c)
              v[r] += res.dz_dx_ref(r, c) * sw * rhs_p_[pr.opt_id * D + c];
      }
      const Landmark& lm = landmarks_[res.landmark_id];
      if (num_active_landmarks_ > 0 && lm.is_active && L > 0)
        for (int r = 0; r < 2; ++r)
          for (int c = 0; c < L; ++c)
            v[r] += res.dz_dlm(r, c) * sw * rhs_l_[lm.opt_id * L + c];
      denom += v[0] * v[0] + v[1] * v[1];
    }
    if (num_active_poses_ > 0) {
      for (const BinaryResidual& res : binary_residuals_) {
        Vec6 v;
        const Pose& p1 = poses_[res.x1_id];
        const Pose& p2 = poses_[res.x2_id];
        Vec6 g1, g2;
        if (p1.is_active) for (int i = 0; i < 6; ++i) g1[i] = rhs_p_[p1.opt_id * D + i];
        if (p2.is_active) for (int i = 0; i < 6; ++i) g2[i] = rhs_p_[p2.opt_id * D + i];
        if (p1.is_active) v += res.cov_inv_sqrt * res.dz_dx1 * g1;
        if (p2.is_active) v += res.cov_inv_sqrt * res.dz_dx2 * g2;
        denom += v.squaredNorm();
      }
      for (const UnaryResidual& res : unary_residuals_) {
        const Pose& p = poses_[res.pose_id];
        if (!p.is_active) continue;
        Vec6 g; for (int i = 0; i < 6; ++i) g[i] = rhs_p_[p.opt_id * D + i];
        denom += (res.cov_inv_sqrt * res.dz_dx * g).squaredNorm();
      }
      for (const ImuResidual& res : inertial_residuals_) {
        const Pose& p1 = poses_[res.pose1_id];
        const Pose& p2 = poses_[res.pose2_id];
        Vec15 v, g1, g2;
        if (p1.is_active) for (int i = 0; i < D; ++i) g1[i] = rhs_p_[p1.opt_id * D + i];
        if (p2.is_active) for (int i = 0; i < D; ++i) g2[i] = rhs_p_[p2.opt_id * D + i];
        if (p1.is_active) v += res.cov_inv_sqrt * imu_block(res, p1) * g1;
        if (p2.is_active) v += res.cov_inv_sqrt * imu_block(res, p2) * g2;
        double s = 0;
        for (int i = 0; i < RS; ++i) s += v[i] * v[i];
        denom += s;
      }
    }
    return denom;
  }

  struct Snapshot {
    std::vector<SE3> t; std::vector<Vec3> v; std::vector<Vec6> b;
    std::vector<Vec4> xs, xw; std::vector<bool> reliable;
    std::vector<std::vector<SE3>> t_sw;
    Pinhole cam0;  // params_backup (:1025-1028, 1099-1102): the intrinsics ARE restored, T_vs is not
  };
  // The reference deep-copies landmarks_, poses_, imu_ (:1022-1028, :1096-1102);
  // only the fields ApplyUpdate can change need restoring.
  Snapshot TakeSnapshot() const {
    Snapshot s;
    for (const Pose& p : poses_) {
      s.t.push_back(p.t_wp); s.v.push_back(p.v_w); s.b.push_back(p.b);
      if (kTvsInCalib) s.t_sw.push_back(p.t_sw);
    }
    for (const Landmark& l : landmarks_) {
      s.xs.push_back(l.x_s); s.xw.push_back(l.x_w); s.reliable.push_back(l.is_reliable);
    }
    if (!rig_.empty()) s.cam0 = rig_[0].model;
    return s;
  }
  void Restore(const Snapshot& s) {
    for (size_t i = 0; i < poses_.size(); ++i) {
      poses_[i].t_wp = s.t[i]; poses_[i].v_w = s.v[i]; poses_[i].b = s.b[i];
      // the copies were taken before ApplyUpdate cleared t_sw; restoring the poses
      // restores those caches too, which equals recomputing them from t_wp — unless T_vs is a
      // parameter: the rig is not part of the copies, so the restored caches are those of the
      // T_vs BEFORE the rejected step while rig_[0].t_vs keeps the step (a reference quirk the
      // restatement keeps: the next evaluation reads the caches, the next Jacobian both).
      if (kTvsInCalib) poses_[i].t_sw = s.t_sw[i];
      else poses_[i].t_sw.clear();
    }
    for (size_t i = 0; i < landmarks_.size(); ++i) {
      landmarks_[i].x_s = s.xs[i]; landmarks_[i].x_w = s.xw[i];
      landmarks_[i].is_reliable = s.reliable[i];
    }
    if (!rig_.empty()) rig_[0].model = s.cam0;  // :1066, :1147
  }

  // BundleAdjuster.cpp:838-1161
  bool SolveInternal(const std::vector<double>& rhs_p_sc, double gn_damping,
                     bool error_increase_allowed, bool use_dogleg) {
    bool gn_computed = false;
    Delta delta_sd, delta_dl, delta_gn;
    double proj_error, binary_error, unary_error, inertial_error;
    if (use_dogleg) {
      const double numerator = sq_norm(rhs_p_) + sq_norm(rhs_l_) + sq_norm(rhs_k_);
      const double denominator = SteepestDescentDenominator();
      const double factor = numerator / denominator;
      delta_sd.delta_p = rhs_p_; delta_sd.delta_l = rhs_l_; delta_sd.delta_k = rhs_k_;
      for (double& x : delta_sd.delta_p) x *= factor;
      for (double& x : delta_sd.delta_l) x *= factor;
      for (double& x : delta_sd.delta_k) x *= factor;
      // :927-929: the steepest-descent norm leaves delta_k out
      const double delta_sd_norm =
          std::sqrt(sq_norm(delta_sd.delta_p) + sq_norm(delta_sd.delta_l));
      uint32_t iteration_count = 0;
      while (1) {
        iteration_count++;
        if (iteration_count > options_.dogleg_max_inner_iterations) break;
        if (delta_sd_norm > trust_region_size_ && trust_region_size_ != kTrustRegionAuto) {
          const double f = trust_region_size_ / delta_sd_norm;
          delta_dl = delta_sd;
          for (double& x : delta_dl.delta_p) x *= f;
          for (double& x : delta_dl.delta_l) x *= f;
          for (double& x : delta_dl.delta_k) x *= f;
        } else {
          if (!gn_computed) {
            if (num_active_poses_ > 0) {
              CalculateGn(rhs_p_sc, delta_gn);
              if (summary_.result == SolverError || summary_.result == FactorizationError)
                return false;
            }
            GetLandmarkDelta(delta_gn, num_active_poses_, num_active_landmarks_,
                             delta_gn.delta_l);
            gn_computed = true;
          }
          if (delta_gn.delta_k.size() != delta_sd.delta_k.size())  // no active pose: no GN step
            delta_gn.delta_k.assign(delta_sd.delta_k.size(), 0.0);
          const double delta_gn_norm = std::sqrt(sq_norm(delta_gn.delta_p) + sq_norm(delta_gn.delta_k) +
                                                 sq_norm(delta_gn.delta_l));
          const bool delta_gn_good = !std::isnan(delta_gn_norm) && !std::isinf(delta_gn_norm);
          if (delta_gn_good && trust_region_size_ == kTrustRegionAuto)
            trust_region_size_ = delta_gn_norm;
          if (delta_gn_good && delta_gn_norm <= trust_region_size_) {
            delta_dl = delta_gn;
          } else {
            Delta diff = delta_gn;
            for (size_t i = 0; i < diff.delta_p.size(); ++i) diff.delta_p[i] -= delta_sd.delta_p[i];
            for (size_t i = 0; i < diff.delta_l.size(); ++i) diff.delta_l[i] -= delta_sd.delta_l[i];
            for (size_t i = 0; i < diff.delta_k.size(); ++i) diff.delta_k[i] -= delta_sd.delta_k[i];
            const double a = sq_norm(diff.delta_p) + sq_norm(diff.delta_l) + sq_norm(diff.delta_k);
            double dot = 0;
            for (size_t i = 0; i < diff.delta_p.size(); ++i) dot += diff.delta_p[i] * delta_sd.delta_p[i];
            for (size_t i = 0; i < diff.delta_k.size(); ++i) dot += diff.delta_k[i] * delta_sd.delta_k[i];
            for (size_t i = 0; i < diff.delta_l.size(); ++i) dot += diff.delta_l[i] * delta_sd.delta_l[i];
            const double b = 2 * dot;
            const double c = (sq_norm(delta_sd.delta_p) + sq_norm(delta_sd.delta_k) +
                              sq_norm(delta_sd.delta_l)) -
                             trust_region_size_ * trust_region_size_;
            double beta = 0;
            // Quirk Q3 (:1006-1013): -(b*b), not -b.
            if (b * b > 4 * a * c && a > 1e-10)
              beta = (-(b * b) + std::sqrt(b * b - 4 * a * c)) / (2 * a);
            delta_dl = delta_sd;
            for (size_t i = 0; i < diff.delta_p.size(); ++i) delta_dl.delta_p[i] += beta * diff.delta_p[i];
            for (size_t i = 0; i < diff.delta_l.size(); ++i) delta_dl.delta_l[i] += beta * diff.delta_l[i];
            for (size_t i = 0; i < diff.delta_k.size(); ++i) delta_dl.delta_k[i] += beta * diff.delta_k[i];
          }
        }
        const Snapshot snap = TakeSnapshot();
        EvaluateResiduals(&proj_error, &binary_error, &unary_error, &inertial_error);
        summary_.pre_solve_norm = proj_error + inertial_error + binary_error + unary_error;
        if (options_.apply_results) ApplyUpdate(delta_dl, false);
        EvaluateResiduals(&proj_error, &binary_error, &unary_error, &inertial_error);
        summary_.post_solve_norm = proj_error + inertial_error + binary_error + unary_error;
        last_delta_ = delta_dl;
        if (summary_.post_solve_norm > summary_.pre_solve_norm) {
          if (options_.apply_results) Restore(snap);
          trust_region_size_ /= 2;
        } else {
          proj_error_ = proj_error; unary_error_ = unary_error;
          binary_error_ = binary_error; inertial_error_ = inertial_error;
          trust_region_size_ *= 2;
          break;
        }
      }
    } else {
      Delta delta;
      delta.delta_p.assign(num_active_poses_ * kPoseDim, 0.0);
      if (num_active_poses_ > 0) {
        CalculateGn(rhs_p_sc, delta);
        if (summary_.result == SolverError || summary_.result == FactorizationError)
          return false;
      }
      const Snapshot snap = TakeSnapshot();
      GetLandmarkDelta(delta, num_active_poses_, num_active_landmarks_, delta.delta_l);
      for (double& x : delta.delta_l) x *= gn_damping;
      for (double& x : delta.delta_k) x *= gn_damping;
      for (double& x : delta.delta_p) x *= gn_damping;
      EvaluateResiduals(&proj_error, &binary_error, &unary_error, &inertial_error);
      const double prev_error = proj_error + inertial_error + binary_error + unary_error;
      if (options_.apply_results) ApplyUpdate(delta, false);
      last_delta_ = delta;
      EvaluateResiduals(&proj_error, &binary_error, &unary_error, &inertial_error);
      const double postError = proj_error + inertial_error + binary_error + unary_error;
      // taps only: the GN branch of the reference never writes these two fields
      gn_prev_error_ = prev_error; gn_post_error_ = postError;
      if (postError > prev_error && !error_increase_allowed) {
        if (options_.apply_results) Restore(snap);
        summary_.result = ErrorIncreased;
        return false;
      } else {
        proj_error_ = proj_error; unary_error_ = unary_error;
        binary_error_ = binary_error; inertial_error_ = inertial_error;
      }
    }
    return true;
  }
  double gn_prev_error_ = 0, gn_post_error_ = 0;

  // BundleAdjuster.cpp:278-705
  void Solve(uint32_t max_iter, double gn_damping, bool error_increase_allowed) {
    if (proj_residuals_.empty() && binary_residuals_.empty() && unary_residuals_.empty() &&
        inertial_residuals_.empty())
      return;
    memset(&timers_, 0, sizeof(timers_));
    const double t_total = now_s();
    summary_.iterations_run = 0;
    if (kLmDim == 1) {  // :288-296
      for (Landmark& lm : landmarks_) {
        lm.x_s = MultHomogeneous(poses_[lm.ref_pose_id].GetTsw(lm.ref_cam_id, rig_), lm.x_w);
        const double length = std::sqrt(lm.x_s[0] * lm.x_s[0] + lm.x_s[1] * lm.x_s[1] +
                                        lm.x_s[2] * lm.x_s[2]);
        lm.x_s = lm.x_s * (1.0 / length);
      }
    }
    for (uint32_t kk = 0; kk < max_iter; ++kk) {
      LineariseAndReduce();
      summary_.iterations_run++;
      if (!SolveInternal(rhs_p_sc_, gn_damping, error_increase_allowed,
                         options_.use_dogleg != 0))
        break;
      // :648-661  exit tests (Quirk Q14).  In the GN branch pre/post_solve_norm are
      // never assigned by the reference (they keep their previous values).
      if ((std::fabs(summary_.post_solve_norm - summary_.pre_solve_norm) /
           summary_.pre_solve_norm) < options_.error_change_threshold) {
        summary_.result = ErrorChangeBelowThreshold;
        break;
      }
      if (summary_.delta_norm < options_.param_change_threshold) {
        summary_.result = ParamChangeBelowThreshold;
        break;
      }
    }
    if (kLmDim == 1) {  // :672-678
      for (Landmark& lm : landmarks_)
        lm.x_w = MultHomogeneous(
            poses_[lm.ref_pose_id].GetTsw(lm.ref_cam_id, rig_).inverse(), lm.x_s);
    }
    // :680-704
    summary_.cond_inertial_error = 0; summary_.cond_proj_error = 0;
    summary_.num_cond_inertial_residuals = conditioning_inertial_residuals_.size();
    summary_.num_inertial_residuals = inertial_residuals_.size();
    summary_.inertial_error = inertial_error_;
    for (uint32_t id : conditioning_inertial_residuals_)
      summary_.cond_inertial_error += inertial_residuals_[id].mahalanobis_distance;
    summary_.num_cond_proj_residuals = conditioning_proj_residuals_.size();
    summary_.num_proj_residuals = proj_residuals_.size();
    summary_.proj_error = proj_error_;
    for (uint32_t id : conditioning_proj_residuals_) {
      const ProjectionResidual& res = proj_residuals_[id];
      summary_.cond_proj_error += res.mahalanobis_distance / res.weight;
    }
    summary_.unary_error = unary_error_; summary_.binary_error = binary_error_;
    summary_.trust_region_size = trust_region_size_;
    timers_.total = now_s() - t_total;
  }
};

// ---------------------------------------------------------------------------
extern "C" {

void orc_default_options(orc_options* o) {
  o->trust_region_size = kTrustRegionAuto;
  o->gyro_sigma = 5.3088444e-5;          // IMU_GYRO_SIGMA        Types.h:33
  o->gyro_bias_sigma = 1.4125375e-4;     // IMU_GYRO_BIAS_SIGMA   Types.h:34
  o->accel_sigma = 0.001883649;          // IMU_ACCEL_SIGMA       Types.h:35
  o->accel_bias_sigma = 1.2589254e-2;    // IMU_ACCEL_BIAS_SIGMA  Types.h:36
  o->projection_outlier_threshold = 1.0;
  o->error_change_threshold = 0.01;
  o->param_change_threshold = 1e-3;
  o->dogleg_max_inner_iterations = 100;
  o->apply_results = 1; o->use_dogleg = 1; o->use_triangular_matrices = 1;
  o->use_sparse_solver = 1;
  o->regularize_biases_in_batch = 1; o->enable_auto_regularization = 1;
  o->use_robust_norm_for_proj_residuals = 1; o->use_robust_norm_for_inertial_residuals = 0;
}

orc_ba* orc_create(int lm_dim, int pose_dim) { return orc_create_calib(lm_dim, pose_dim, 0, 0); }
orc_ba* orc_create_calib(int lm_dim, int pose_dim, int calib_size, int do_tvs) {
  if (!(lm_dim == 0 || lm_dim == 1 || lm_dim == 3)) return nullptr;
  if (!(pose_dim == 6 || pose_dim == 9 || pose_dim == 15)) return nullptr;
  // the pinhole model has four parameters, the FOV model five (camera 0 must be of that model
  // when Solve() runs: checked there, as the reference's fixed-size assignment at parallel_algos.h:115 would)
  if (calib_size != 0 && calib_size != 4 && calib_size != 5) return nullptr;
  if (calib_size != 0 && do_tvs) return nullptr;           // the reference wipes the intrinsics then (see orc_ba)
  if ((do_tvs || calib_size) && lm_dim != 1) return nullptr;  // both Jacobians exist for LmSize 1 only (parallel_algos.h:102)
  return new orc_ba(lm_dim, pose_dim, do_tvs != 0, calib_size);
}
void orc_destroy(orc_ba* h) { delete h; }
void orc_init(orc_ba* h, const orc_options* o) { h->Init(*o); }
void orc_set_gravity(orc_ba* h, const double g[3]) { for (int i = 0; i < 3; ++i) h->g_vec[i] = g[i]; }

uint32_t orc_add_camera(orc_ba* h, const double params[4], const double t_vs[7]) {
  Camera c;
  c.model.fx = params[0]; c.model.fy = params[1]; c.model.u0 = params[2]; c.model.v0 = params[3];
  c.t_vs = se3_from7(t_vs);
  h->rig_.push_back(c);
  return h->rig_.size();  // BundleAdjuster.h:259-263 returns NumCams()
}
void orc_set_camera_fov(orc_ba* h, uint32_t cam_id, double w) {
  h->rig_[cam_id].model.model = 1;
  h->rig_[cam_id].model.w = w;
}
double orc_get_camera_fov(const orc_ba* h, uint32_t cam_id) { return h->rig_[cam_id].model.w; }
uint32_t orc_add_pose(orc_ba* h, const double t_wp[7], const double v_w[3], const double b[6],
                      int is_active, double time) {
  Vec3 v; Vec6 bb;
  if (v_w) for (int i = 0; i < 3; ++i) v[i] = v_w[i];
  if (b) for (int i = 0; i < 6; ++i) bb[i] = b[i];
  return h->AddPose(se3_from7(t_wp), v, bb, is_active != 0, time);
}
void orc_set_pose_cam_params(orc_ba* h, uint32_t pose_id, const double params4[4]) {
  if (pose_id >= h->poses_.size()) return;
  for (int i = 0; i < 4; ++i) h->poses_[pose_id].cam_params[i] = params4[i];
  h->poses_[pose_id].has_cam_params = true;
}
void orc_set_imu_noise(orc_ba* h, const double r6[6], const double rb6[6]) {  // SetImuCalibration: imu_.r, imu_.r_b
  for (int i = 0; i < 6; ++i) { h->imu_r[i] = r6[i]; h->imu_r_b[i] = rb6[i]; }
}
void orc_set_calculate_inertial_covariance_once(orc_ba* h, int on) { h->calculate_inertial_covariance_once_ = on != 0; }
int orc_set_use_per_pose_cam_params(orc_ba* h, int on) {
  if (on)
    for (const auto& p : h->poses_)
      if (!p.has_cam_params) return 1;  // the reference would SetParams an empty vector
  h->use_per_pose_cam_params_ = on != 0;
  return 0;
}
uint32_t orc_add_landmark(orc_ba* h, const double x_w[4], uint32_t ref_pose_id,
                          uint32_t ref_cam_id, int is_active) {
  Vec4 x; for (int i = 0; i < 4; ++i) x[i] = x_w[i];
  return h->AddLandmark(x, ref_pose_id, ref_cam_id, is_active != 0);
}
uint32_t orc_add_projection_residual(orc_ba* h, const double z[2], uint32_t meas_pose_id,
                                     uint32_t landmark_id, uint32_t cam_id, double weight) {
  Vec2 zz; zz[0] = z[0]; zz[1] = z[1];
  return h->AddProjectionResidual(zz, meas_pose_id, landmark_id, cam_id, weight);
}
static Mat<6, 6> cov_from(const double* cov) {
  Mat<6, 6> c = Mat<6, 6>::Identity();
  if (cov) for (int i = 0; i < 36; ++i) c.a[i] = cov[i];
  return c;
}
uint32_t orc_add_unary_constraint(orc_ba* h, uint32_t pose_id, const double t_wv[7],
                                  const double cov[36], int use_rotation) {
  return h->AddUnaryConstraint(pose_id, se3_from7(t_wv), cov_from(cov), use_rotation != 0);
}
uint32_t orc_add_binary_constraint(orc_ba* h, uint32_t p1, uint32_t p2, const double t_12[7],
                                   const double cov[36], double weight, int use_rotation) {
  return h->AddBinaryConstraint(p1, p2, se3_from7(t_12), cov_from(cov), weight,
                                use_rotation != 0);
}
static std::vector<ImuMeasurement> meas_from(const double* meas, uint32_t n) {
  std::vector<ImuMeasurement> m(n);
  for (uint32_t i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) { m[i].w[k] = meas[i * 7 + k]; m[i].a[k] = meas[i * 7 + 3 + k]; }
    m[i].time = meas[i * 7 + 6];
  }
  return m;
}
uint32_t orc_add_imu_residual(orc_ba* h, uint32_t p1, uint32_t p2, const double* meas,
                              uint32_t n, double weight) {
  return h->AddImuResidual(p1, p2, meas_from(meas, n), weight);
}
void orc_regularize_pose(orc_ba* h, uint32_t pose_id, int translation, int gravity, int bias,
                         int rotation) {
  h->RegularizePose(pose_id, translation != 0, gravity != 0, bias != 0, rotation != 0);
}
void orc_set_root_pose_id(orc_ba* h, uint32_t id) { h->root_pose_id_ = id; }

void orc_add_poses(orc_ba* h, uint32_t n, const double* t_wp, const double* v_w, const double* b,
                   const uint8_t* is_active, const double* time) {
  for (uint32_t i = 0; i < n; ++i)
    orc_add_pose(h, t_wp + 7 * i, v_w ? v_w + 3 * i : nullptr, b ? b + 6 * i : nullptr,
                 is_active ? is_active[i] : 1, time ? time[i] : -1);
}
void orc_add_landmarks(orc_ba* h, uint32_t n, const double* x_w, const uint32_t* ref_pose_id,
                       const uint32_t* ref_cam_id, const uint8_t* is_active) {
  for (uint32_t i = 0; i < n; ++i)
    orc_add_landmark(h, x_w + 4 * i, ref_pose_id[i], ref_cam_id ? ref_cam_id[i] : 0,
                     is_active ? is_active[i] : 1);
}
void orc_add_projection_residuals(orc_ba* h, uint32_t n, const double* z,
                                  const uint32_t* meas_pose_id, const uint32_t* landmark_id,
                                  const uint32_t* cam_id, const double* weight, uint32_t* out_ids) {
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t id = orc_add_projection_residual(h, z + 2 * i, meas_pose_id[i], landmark_id[i],
                                                    cam_id ? cam_id[i] : 0, weight ? weight[i] : 1.0);
    if (out_ids) out_ids[i] = id;
  }
}

void orc_solve(orc_ba* h, uint32_t max_iter, double gn_damping, int error_increase_allowed) {
  h->Solve(max_iter, gn_damping, error_increase_allowed != 0);
}

uint32_t orc_num_poses(const orc_ba* h) { return h->poses_.size(); }
uint32_t orc_num_landmarks(const orc_ba* h) { return h->landmarks_.size(); }
uint32_t orc_num_proj_residuals(const orc_ba* h) { return h->proj_residuals_.size(); }
void orc_get_pose(const orc_ba* h, uint32_t id, double t_wp[7], double v_w[3], double b[6]) {
  const Pose& p = h->poses_[id];
  se3_to7(p.t_wp, t_wp);
  if (v_w) for (int i = 0; i < 3; ++i) v_w[i] = p.v_w[i];
  if (b) for (int i = 0; i < 6; ++i) b[i] = p.b[i];
}
void orc_get_poses(const orc_ba* h, double* t_wp, double* v_w, double* b) {
  for (uint32_t i = 0; i < h->poses_.size(); ++i)
    orc_get_pose(h, i, t_wp + 7 * i, v_w ? v_w + 3 * i : nullptr, b ? b + 6 * i : nullptr);
}
void orc_get_landmark(const orc_ba* h, uint32_t id, double x_w[4]) {
  for (int i = 0; i < 4; ++i) x_w[i] = h->landmarks_[id].x_w[i];
}
void orc_get_landmarks(const orc_ba* h, double* x_w) {
  for (uint32_t i = 0; i < h->landmarks_.size(); ++i) orc_get_landmark(h, i, x_w + 4 * i);
}
int orc_is_landmark_reliable(const orc_ba* h, uint32_t id) { return h->landmarks_[id].is_reliable; }
double orc_landmark_outlier_ratio(const orc_ba* h, uint32_t id) {
  // BundleAdjuster.cpp:1805-1812
  const Landmark& l = h->landmarks_[id];
  return l.proj_residuals.empty() ? 0 : (double)l.num_outlier_residuals / l.proj_residuals.size();
}
void orc_get_summary(const orc_ba* h, orc_summary* s) { *s = h->summary_; }

uint32_t orc_num_pose_params(const orc_ba* h) { return h->num_active_poses_ * h->kPoseDim; }
uint32_t orc_num_lm_params(const orc_ba* h) { return h->num_active_landmarks_ * h->kLmDim; }
void orc_get_S(const orc_ba* h, double* s) { memcpy(s, h->s_.data(), h->s_.size() * sizeof(double)); }
void orc_get_rhs(const orc_ba* h, double* r) { memcpy(r, h->rhs_p_sc_.data(), h->rhs_p_sc_.size() * 8); }
void orc_get_rhs_p(const orc_ba* h, double* r) { memcpy(r, h->rhs_p_.data(), h->rhs_p_.size() * 8); }
void orc_get_rhs_l(const orc_ba* h, double* r) { memcpy(r, h->rhs_l_.data(), h->rhs_l_.size() * 8); }
void orc_get_delta_p(const orc_ba* h, double* d) {
  memcpy(d, h->last_delta_.delta_p.data(), h->last_delta_.delta_p.size() * 8);
}
void orc_get_delta_l(const orc_ba* h, double* d) {
  memcpy(d, h->last_delta_.delta_l.data(), h->last_delta_.delta_l.size() * 8);
}
uint32_t orc_num_calib_params(const orc_ba* h) { return h->kCalibDim; }
void orc_get_delta_k(const orc_ba* h, double* d) {
  memcpy(d, h->last_delta_.delta_k.data(), h->last_delta_.delta_k.size() * 8);
}
void orc_get_rhs_k(const orc_ba* h, double* r) { memcpy(r, h->rhs_k_.data(), h->rhs_k_.size() * 8); }
void orc_get_camera_pose(const orc_ba* h, uint32_t cam_id, double t_vs[7]) { se3_to7(h->rig_[cam_id].t_vs, t_vs); }
// SolutionSummary::calibration_marginals (BundleAdjuster.cpp:771-784): solves with the unit vectors
// of the calibration unknowns, bottom-right block.  Formed on request from the s_ of the last
// iteration (the reference forms it inside CalculateGn when the option is set: same matrix).
int orc_get_calibration_marginals(const orc_ba* h, double* cov) {
  const uint32_t K = h->kCalibDim, nt = h->num_active_poses_ * h->kPoseDim + K;
  if (K == 0 || h->s_.size() != (size_t)nt * nt) return 0;
  std::vector<double> unit(nt), x(nt);
  for (uint32_t i = 0; i < K; ++i) {
    std::fill(unit.begin(), unit.end(), 0.0);
    unit[nt - K + i] = 1.0;
    ldlt_solve_upper(nt, h->s_.data(), unit.data(), x.data());
    for (uint32_t r = 0; r < K; ++r) cov[(size_t)r * K + i] = x[nt - K + r];
  }
  return (int)K;
}
void orc_get_camera_params(const orc_ba* h, uint32_t cam_id, double p[4]) {
  const Pinhole& m = h->rig_[cam_id].model;
  p[0] = m.fx; p[1] = m.fy; p[2] = m.u0; p[3] = m.v0;
}
void orc_get_proj_calib_jacobians(const orc_ba* h, double* j_k) {  // 2 x kCalibDim per residual id
  const int K = h->kCalibDim;
  for (size_t i = 0; i < h->proj_residuals_.size(); ++i)
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < K; ++c) j_k[(i * 2 + r) * K + c] = h->jk(h->proj_residuals_[i], r, c);
}
void orc_math_transfer(const double params[4], const double t_ba[7], const double pix[2], double rho, double out[2],
                       double jac8[8]) {
  Pinhole m; m.fx = params[0]; m.fy = params[1]; m.u0 = params[2]; m.v0 = params[3];
  const SE3 t = se3_from7(t_ba);
  Vec2 px; px[0] = pix[0]; px[1] = pix[1];
  const Vec2 p = m.Transfer3d(t, m.Unproject(px), rho);
  out[0] = p[0]; out[1] = p[1];
  if (jac8) {
    const Mat<2, 5> J = m.dTransfer_dparams(t, px, rho);
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 4; ++c) jac8[4 * r + c] = J(r, c);
  }
}
void orc_math_transfer_fov(const double params[5], const double t_ba[7], const double pix[2], double rho,
                           double out[2], double jac10[10]) {
  Pinhole m; m.fx = params[0]; m.fy = params[1]; m.u0 = params[2]; m.v0 = params[3]; m.w = params[4]; m.model = 1;
  const SE3 t = se3_from7(t_ba);
  Vec2 px; px[0] = pix[0]; px[1] = pix[1];
  const Vec2 p = m.Transfer3d(t, m.Unproject(px), rho);
  out[0] = p[0]; out[1] = p[1];
  if (jac10) { const Mat<2, 5> J = m.dTransfer_dparams(t, px, rho); memcpy(jac10, J.a, 10 * sizeof(double)); }
}
void orc_math_fov_project(const double params[5], const double P[3], double pix[2], double dpix_dP[6], double ray[3]) {
  Pinhole m; m.fx = params[0]; m.fy = params[1]; m.u0 = params[2]; m.v0 = params[3]; m.w = params[4]; m.model = 1;
  Vec3 X; X[0] = P[0]; X[1] = P[1]; X[2] = P[2];
  const Vec2 p = m.Project(X);
  pix[0] = p[0]; pix[1] = p[1];
  if (dpix_dP) { const Mat<2, 3> d = m.dProject_dP(X); memcpy(dpix_dP, d.a, 6 * sizeof(double)); }
  if (ray) { const Vec3 r = m.Unproject(p); ray[0] = r[0]; ray[1] = r[1]; ray[2] = r[2]; }
}
void orc_get_proj_tvs_jacobians(const orc_ba* h, double* j_tvs) {
  for (size_t i = 0; i < h->proj_residuals_.size(); ++i)
    memcpy(j_tvs + 12 * i, h->proj_residuals_[i].dz_dtvs.a, 12 * 8);
}
void orc_get_proj_weights(const orc_ba* h, double* w) {
  for (size_t i = 0; i < h->proj_residuals_.size(); ++i) w[i] = h->proj_residuals_[i].weight;
}
void orc_get_proj_residuals(const orc_ba* h, double* r2) {
  for (size_t i = 0; i < h->proj_residuals_.size(); ++i) {
    r2[2 * i] = h->proj_residuals_[i].residual[0];
    r2[2 * i + 1] = h->proj_residuals_[i].residual[1];
  }
}
void orc_get_imu_residuals(const orc_ba* h, double* r15) {
  for (size_t i = 0; i < h->inertial_residuals_.size(); ++i)
    for (int k = 0; k < 15; ++k) r15[15 * i + k] = h->inertial_residuals_[i].residual[k];
}
uint32_t orc_num_imu_residuals(const orc_ba* h) { return (uint32_t)h->inertial_residuals_.size(); }
void orc_get_proj_jacobians(const orc_ba* h, double* j_meas, double* j_ref, double* j_lm) {
  const int L = h->kLmDim;
  for (size_t i = 0; i < h->proj_residuals_.size(); ++i) {
    const ProjectionResidual& r = h->proj_residuals_[i];
    if (j_meas) memcpy(j_meas + 12 * i, r.dz_dx_meas_raw.a, 12 * 8);
    if (j_ref) memcpy(j_ref + 12 * i, r.dz_dx_ref_raw.a, 12 * 8);
    if (j_lm)
      for (int rr = 0; rr < 2; ++rr)
        for (int c = 0; c < L; ++c) j_lm[(size_t)i * 2 * L + rr * L + c] = r.dz_dlm(rr, c);
  }
}
void orc_get_imu_jacobians(const orc_ba* h, uint32_t id, double* dz_dx1, double* dz_dx2,
                           double* cov_inv, double* residual) {
  const ImuResidual& r = h->inertial_residuals_[id];
  if (dz_dx1) memcpy(dz_dx1, r.dz_dx1.a, 225 * 8);
  if (dz_dx2) memcpy(dz_dx2, r.dz_dx2.a, 225 * 8);
  if (cov_inv) memcpy(cov_inv, r.cov_inv.a, 225 * 8);
  if (residual) memcpy(residual, r.residual.a, 15 * 8);
}
void orc_get_binary_jacobians(const orc_ba* h, uint32_t id, double* dz_dx1, double* dz_dx2,
                              double* residual) {
  const BinaryResidual& r = h->binary_residuals_[id];
  if (dz_dx1) memcpy(dz_dx1, r.dz_dx1.a, 36 * 8);
  if (dz_dx2) memcpy(dz_dx2, r.dz_dx2.a, 36 * 8);
  if (residual) memcpy(residual, r.residual.a, 6 * 8);
}
void orc_get_unary_jacobian(const orc_ba* h, uint32_t id, double* dz_dx, double* residual) {
  const UnaryResidual& r = h->unary_residuals_[id];
  if (dz_dx) memcpy(dz_dx, r.dz_dx.a, 36 * 8);
  if (residual) memcpy(residual, r.residual.a, 6 * 8);
}
void orc_get_timers(const orc_ba* h, orc_timers* t) { *t = h->timers_; }

// ---- stand-alone math taps -------------------------------------------------
void orc_math_dlog_dq(const double q[4], double out[12]) {
  const Mat<3, 4> J = dlog_dq(Quat(q[0], q[1], q[2], q[3]));
  memcpy(out, J.a, 12 * 8);
}
void orc_math_so3_log(const double q[4], double out[3]) {
  const Vec3 w = SO3::raw(Quat(q[0], q[1], q[2], q[3])).log();
  for (int i = 0; i < 3; ++i) out[i] = w[i];
}
void orc_math_so3_exp(const double w[3], double q[4]) {
  Vec3 v; for (int i = 0; i < 3; ++i) v[i] = w[i];
  const SO3 r = SO3::exp(v);
  q[0] = r.q.x; q[1] = r.q.y; q[2] = r.q.z; q[3] = r.q.w;
}
void orc_math_exp_decoupled(const double t[7], const double x[6], double out[7]) {
  Vec6 v; for (int i = 0; i < 6; ++i) v[i] = x[i];
  se3_to7(exp_decoupled(se3_from7(t), v), out);
}
void orc_math_log_decoupled(const double a[7], const double b[7], double out[6]) {
  const Vec6 v = log_decoupled(se3_from7(a), se3_from7(b));
  for (int i = 0; i < 6; ++i) out[i] = v[i];
}
void orc_math_se3_mul(const double a[7], const double b[7], double out[7]) {
  se3_to7(se3_from7(a) * se3_from7(b), out);
}
void orc_math_se3_inv(const double a[7], double out[7]) { se3_to7(se3_from7(a).inverse(), out); }
void orc_set_num_threads(int n) { g_ldlt_threads = n < 1 ? 1 : n; }
int orc_get_num_threads(void) { return g_ldlt_threads; }
void orc_math_dense_solve_upper(uint32_t n, const double* s, const double* rhs, double* x) {
  ldlt_solve_upper(n, s, rhs, x);
}
void orc_math_integrate(const double pose_t[7], const double v[3], const double* meas, uint32_t n,
                        const double bg[3], const double ba[3], const double g[3],
                        const double r6[6], double out_t[7], double out_v[3],
                        double* dpose_db, double* c) {
  ImuPose start; start.t_wp = se3_from7(pose_t); start.time = 0;
  Vec3 vbg, vba, vg; Vec6 r;
  for (int i = 0; i < 3; ++i) { start.v_w[i] = v[i]; vbg[i] = bg[i]; vba[i] = ba[i]; vg[i] = g[i]; }
  if (r6) for (int i = 0; i < 6; ++i) r[i] = r6[i];
  Mat<10, 6> db; Mat<10, 10> cc;
  const bool jac = dpose_db != nullptr || c != nullptr;
  const ImuPose out = IntegrateResidual(start, meas_from(meas, n), vbg, vba, vg,
                                        jac ? &db : nullptr, nullptr, jac ? &cc : nullptr,
                                        jac ? &r : nullptr);
  se3_to7(out.t_wp, out_t);
  for (int i = 0; i < 3; ++i) out_v[i] = out.v_w[i];
  if (dpose_db) memcpy(dpose_db, db.a, 60 * 8);
  if (c) memcpy(c, cc.a, 100 * 8);
}
// the same with dpose_dpose (10 x 10 over the start state; Types.h:716-718)
void orc_math_integrate_jacobians(const double pose_t[7], const double v[3], const double* meas, uint32_t n,
                                  const double bg[3], const double ba[3], const double g[3], const double r6[6],
                                  double* dpose_db, double* dpose_dpose, double* c) {
  ImuPose start; start.t_wp = se3_from7(pose_t); start.time = 0;
  Vec3 vbg, vba, vg; Vec6 r;
  for (int i = 0; i < 3; ++i) { start.v_w[i] = v[i]; vbg[i] = bg[i]; vba[i] = ba[i]; vg[i] = g[i]; }
  for (int i = 0; i < 6; ++i) r[i] = r6[i];
  Mat<10, 6> db; Mat<10, 10> dd, cc;
  IntegrateResidual(start, meas_from(meas, n), vbg, vba, vg, &db, &dd, &cc, &r);
  memcpy(dpose_db, db.a, 60 * 8);
  memcpy(dpose_dpose, dd.a, 100 * 8);
  memcpy(c, cc.a, 100 * 8);
}

// The Utils.h helpers by op code (the numbering of ba_hip_lie, include/ba_hip.h) — test tap
int orc_math_lie(int op, const double* a, const double* b, double* out) {
  auto quat = [](const double* p) { Quat q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; };
  auto vec3 = [](const double* p) { Vec3 v; v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; return v; };
  auto vec4 = [](const double* p) { Vec4 v; for (int i = 0; i < 4; ++i) v[i] = p[i]; return v; };
#define ORC_PUT(M, N) { const auto m_ = (M); memcpy(out, m_.a, (N) * sizeof(double)); return (N); }
  switch (op) {
    case 1: ORC_PUT(dlog_dq(quat(a)), 12)
    case 2: ORC_PUT(dq_exp_dw(vec3(a)), 12)
    case 3: ORC_PUT(dq1q2_dq1(quat(a)), 16)
    case 4: ORC_PUT(dq1q2_dq2(quat(a)), 16)
    case 5: ORC_PUT(dqx_dq(quat(a), vec3(b)), 12)
    case 6: ORC_PUT(quat(a).matrix(), 9)
    case 7: ORC_PUT(log_decoupled(se3_from7(a), se3_from7(b)), 6)
    case 8: { Vec6 x; for (int i = 0; i < 6; ++i) x[i] = b[i]; se3_to7(exp_decoupled(se3_from7(a), x), out); return 7; }
    case 9: ORC_PUT(dlog_decoupled_dx(se3_from7(a), se3_from7(b)), 36)
    case 10: ORC_PUT(dLog_decoupled_dt1(se3_from7(a), se3_from7(b)), 42)
    case 11: ORC_PUT(dlog_decoupled_dt2(se3_from7(a), se3_from7(b)), 42)
    case 12: ORC_PUT(dexp_decoupled_dx(se3_from7(a)), 42)
    case 13: ORC_PUT(dinv_exp_decoupled_dx(se3_from7(a)), 42)
    case 14: ORC_PUT(dt_x_dt(se3_from7(a), vec4(b)), 28)
    case 15: ORC_PUT(dt1_t2_dt1(se3_from7(a), se3_from7(b)), 49)
    case 16: ORC_PUT(dt1_t2_dt2(se3_from7(a)), 49)
    case 17: ORC_PUT(MultHomogeneous(se3_from7(a), vec4(b)), 4)
  }
#undef ORC_PUT
  return -1;
}

}  // extern "C"
