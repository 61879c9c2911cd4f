// ba::BundleAdjuster<> — MI355X-native host layer.
//
// Source-compatible with the public API of the reference class template
// (/root/reference/include/ba/BundleAdjuster.h:111-763: Init / AddCamera / AddPose /
// AddLandmark / AddUnaryConstraint / AddBinaryConstraint / AddProjectionResidual /
// AddImuResidual / Solve / Get* / RegularizePose / options()), but built differently:
// the problem graph is kept as flat structure-of-arrays ready for upload, and Solve()
// (reference: /root/reference/src/BundleAdjuster.cpp:278-705) drives the gfx950 engine
// through the C-ABI of include/ba_hip.h — one short sequence of calls per Gauss-Newton
// or dogleg iteration, only scalars returning to the host.  There is no CPU solver in
// this class: without a usable MI355X Solve() reports SolverError and prints the
// engine's message.
//
// Supported instantiations: LmSize in {0,1,3}, PoseSize in {6,9,15}; with LmSize 1 the
// self-calibration instantiations (reference :121-134, BundleAdjuster.cpp:46-83, 493-583): DoTvs —
// the extrinsics T_vs of camera 0 as six more unknowns — or CalibSize = 4 / 5 — the parameters of camera 0:
// (fx, fy, u0, v0) of a LinearCamera, (fx, fy, u0, v0, w) of a FovCamera (the reference's
// SelfCalBundleAdjuster; CalibSize must equal the camera's parameter count, as the fixed-size assignment
// at parallel_algos.h:115-118 demands).  Not both at once: the reference's own T_vs block wipes the
// intrinsics columns in that case (BundleAdjuster.cpp:1775-1783).
#pragma once
#include <algorithm>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>
#include <vector>

#include "../ba_hip.h"
#include "HostMath.h"
#include "Types.h"

namespace ba {

constexpr int kTrustRegionAuto = -1;  // reference BundleAdjuster.h:30

enum OptimizationResult {  // reference BundleAdjuster.h:38-46
  Success,
  ErrorIncreased,
  ErrorChangeBelowThreshold,
  ParamChangeBelowThreshold,
  FactorizationError,
  SolverError
};

template <typename Scalar = double>
struct SolutionSummary {  // reference BundleAdjuster.h:48-70
  uint32_t num_proj_residuals = 0;
  uint32_t num_inertial_residuals = 0;
  uint32_t num_cond_proj_residuals = 0;
  uint32_t num_cond_inertial_residuals = 0;
  Scalar cond_proj_error = 0;
  Scalar cond_inertial_error = 0;
  Scalar proj_error_ = 0;
  Scalar inertial_error = 0;
  Scalar delta_norm = 0;
  Scalar pre_solve_norm = 0;
  Scalar post_solve_norm = 0;
  MatX calibration_marginals;  // kCalibDim x kCalibDim block of S^-1 (Options::calculate_calibration_marginals)
  OptimizationResult result = Success;
  bool IsResultGood() const { return (result != SolverError) && (result != FactorizationError); }
};

template <typename Scalar = double>
struct Options {  // reference BundleAdjuster.h:72-107, same names and defaults
  Scalar trust_region_size = kTrustRegionAuto;
  Scalar gyro_sigma = IMU_GYRO_SIGMA;
  Scalar accel_sigma = IMU_ACCEL_SIGMA;
  Scalar gyro_bias_sigma = IMU_GYRO_BIAS_SIGMA;
  Scalar accel_bias_sigma = IMU_ACCEL_BIAS_SIGMA;
  Scalar projection_outlier_threshold = 1.0;
  Scalar error_change_threshold = 0.01;
  Scalar param_change_threshold = 1e-3;
  uint32_t dogleg_max_inner_iterations = 100;
  bool apply_results = true;
  bool use_dogleg = true;
  bool use_triangular_matrices = true;
  bool use_sparse_solver = true;  // accepted; the engine always factorises densely
  bool write_reduced_camera_matrix = false;
  bool keep_reduced_system = false;  // extension: keep S readable (GetReducedSystem taps) without writing files
  // extension (0 = off = reference behaviour): pivots below this fraction of their original
  // diagonal entry make Solve() report FactorizationError instead of applying an arbitrary step
  // on a rank-deficient reduced system (include/ba_hip.h: ba_hip_options::pivot_rel_tolerance)
  Scalar factorization_pivot_tolerance = 0;
  bool calculate_calibration_marginals = false;
  bool use_per_pose_cam_params = false;
  bool regularize_biases_in_batch = true;
  bool enable_auto_regularization = true;
  bool use_robust_norm_for_proj_residuals = true;
  bool use_robust_norm_for_inertial_residuals = false;
  bool calculate_inertial_covariance_once = false;
  // engine placement (not in the reference): HIP device ordinal
  int device = 0;
};

template <typename Scalar = double, int LmSize = 1, int PoseSize = 6, int CalibSize = 0,
          bool DoTvs = false>
class BundleAdjuster {
  static_assert(std::is_same<Scalar, double>::value, "the engine computes in FP64 (REAL_TYPE=double)");
  static_assert(LmSize == 0 || LmSize == 1 || LmSize == 3, "LmSize must be 0, 1 or 3");
  static_assert(PoseSize == 6 || PoseSize == 9 || PoseSize == 15, "PoseSize must be 6, 9 or 15");
  static_assert(CalibSize == 0 || CalibSize == 4 || CalibSize == 5,
                "CalibSize: 0, the 4 pinhole parameters (fx, fy, u0, v0) or the 5 of a FovCamera (fx, fy, u0, v0, w)");
  static_assert(!(CalibSize > 0 && DoTvs), "CalibSize > 0 with DoTvs: the reference wipes the intrinsics columns (BundleAdjuster.cpp:1775-1783)");
  static_assert((!DoTvs && CalibSize == 0) || LmSize == 1, "calibration exists for inverse-depth landmarks only (parallel_algos.h:102-131)");

 public:
  int debug_level_threshold = 0;
  int debug_level = 0;

  static constexpr uint32_t kPrPoseDim = 6;
  static constexpr uint32_t kLmDim = LmSize;
  static constexpr uint32_t kPoseDim = PoseSize;
  static constexpr uint32_t kCalibDim = CalibSize + (DoTvs ? 6 : 0);  // reference :123-124
  static constexpr bool kTvsInCalib = DoTvs;
  static constexpr bool kCamParamsInCalib = CalibSize > 0;
  static constexpr uint32_t kTvsOffset = CalibSize;
  static constexpr bool kVelInState = (kPoseDim >= 9);
  static constexpr bool kBiasInState = (kPoseDim >= 15);
  static constexpr bool kGravityInCalib = false;

  typedef PoseT<Scalar> Pose;
  typedef LandmarkT<Scalar, LmSize> Landmark;
  typedef ProjectionResidualT<Scalar, LmSize> ProjectionResidual;
  typedef ImuMeasurementT<Scalar> ImuMeasurement;
  typedef UnaryResidualT<Scalar> UnaryResidual;    // reference :140
  typedef BinaryResidualT<Scalar> BinaryResidual;  // reference :141
  typedef ImuResidualT<Scalar, kPoseDim, kPoseDim> ImuResidual;  // reference :142
  typedef ImuPoseT<Scalar> ImuPose;  // reference :144
  typedef ImuCalibrationT<Scalar> ImuCalibration;
  typedef ba::Vector2t Vector2t;
  typedef ba::Vector3t Vector3t;
  typedef ba::Vector4t Vector4t;
  typedef ba::Vector6t Vector6t;
  typedef ba::Vector7t Vector7t;
  typedef ba::Vector9t Vector9t;
  typedef std::vector<Scalar> VectorXt;  // reference :152 (Eigen dynamic vector: per-pose camera parameters)
  typedef ba::MatX MatrixXt;             // reference :153
  typedef ba::Matrix3t Matrix3t;
  typedef ba::SE3 SE3t;

  struct Delta {  // reference BundleAdjuster.h:157-162 (filled by GetLastStep)
    std::vector<Scalar> delta_p, delta_k, delta_l;
  };

  BundleAdjuster() { Init(Options<Scalar>()); }
  ~BundleAdjuster() { ReleaseEngine(); }
  BundleAdjuster(const BundleAdjuster&) = delete;
  BundleAdjuster& operator=(const BundleAdjuster&) = delete;

  // reference BundleAdjuster.h:177-237
  void Init(const Options<Scalar>& options, uint32_t num_poses = 0, uint32_t num_measurements = 0,
            uint32_t num_landmarks = 0, const SE3t& t_vs = SE3t()) {
    options_ = options;
    trust_region_size_ = options_.trust_region_size;
    root_pose_id_ = 0;
    num_active_poses_ = 0;
    num_active_landmarks_ = 0;
    imu_.t_vs = t_vs;
    for (int i = 0; i < 3; ++i) {
      imu_.r[i] = options.gyro_sigma * options.gyro_sigma;
      imu_.r[3 + i] = options.accel_sigma * options.accel_sigma;
      imu_.r_b[i] = options.gyro_bias_sigma * options.gyro_bias_sigma;
      imu_.r_b[3 + i] = options.accel_bias_sigma * options.accel_bias_sigma;
    }
    rig_.reset(new Rig<Scalar>());
    poses_.clear(); landmarks_.clear();
    poses_.reserve(std::max(1u, num_poses)); landmarks_.reserve(std::max(1u, num_landmarks));
    pr_z_.clear(); pr_pose_.clear(); pr_lm_.clear(); pr_cam_.clear(); pr_w_.clear();
    proj_view_dirty_ = true; imu_view_dirty_ = true; uploaded_once_ = false;
    imu_cov_reset_ = true;  // a new problem: forget the frozen inertial covariances
    pr_z_.reserve(2 * (size_t)std::max(1u, num_measurements));
    un_pose_.clear(); un_t_.clear(); un_cov_inv_.clear(); un_rot_.clear();
    bin_p1_.clear(); bin_p2_.clear(); bin_t_.clear(); bin_cov_inv_.clear(); bin_cov_inv_sqrt_.clear();
    bin_w_.clear(); bin_rot_.clear();
    imu_p1_.clear(); imu_p2_.clear(); imu_ptr_.assign(1, 0); imu_meas_.clear(); imu_w_.clear();
    conditioning_proj_residuals_.clear(); conditioning_inertial_residuals_.clear();
    proj_error_ = binary_error_ = unary_error_ = inertial_error_ = 0;
    summary_ = SolutionSummary<Scalar>();
    structure_dirty_ = true;
    host_state_stale_ = false;  // a new problem: whatever the engine still holds is obsolete
    last_step_stale_ = false;
    uploaded_poses_ = uploaded_landmarks_ = 0;
  }

  void SetGravity(const Vector3t& g) { imu_.g_vec = g; }          // reference :243-252
  Vector3t GetGravity() const { return imu_.g_vec; }              // reference :254-256

  // reference :259-263 — returns NumCams() after insertion
  uint32_t AddCamera(std::shared_ptr<CameraInterface<Scalar>> cam) {
    rig_->AddCamera(cam);
    structure_dirty_ = true;
    return rig_->NumCams();
  }

  // reference :267-274
  uint32_t AddPose(const SE3t& t_wp, const bool is_active = true, const double time = -1,
                   const int external_id = -1) {
    return AddPose(t_wp, std::vector<Scalar>(), Vector3t::Zero(), Vector6t::Zero(), is_active, time,
                   external_id);
  }
  // reference :292-323; cam_params (any indexable container of the pinhole intrinsics fx, fy, u0,
  // v0; may be empty) is what Options::use_per_pose_cam_params projects this pose's
  // measurements with (parallel_algos.h:54-57)
  template <typename CamParams>
  uint32_t AddPose(const SE3t& t_wv, const CamParams& cam_params, const Vector3t& v_w,
                   const Vector6t& b, const bool is_active = true, const double time = -1,
                   const int external_id = -1) {
    Pose pose;
    for (size_t i = 0; i < (size_t)cam_params.size(); ++i) pose.cam_params.push_back((Scalar)cam_params[i]);
    pose.external_id = external_id;
    pose.time = time;
    pose.t_wp = t_wv;
    pose.v_w = v_w;
    pose.b = b;
    pose.is_active = is_active;
    pose.id = (uint32_t)poses_.size();
    pose.opt_id = is_active ? num_active_poses_++ : UINT_MAX;
    poses_.push_back(pose);
    structure_dirty_ = true;
    return pose.id;
  }

  // reference :326-367
  uint32_t AddLandmark(const Vector4t& x_w, const uint32_t ref_pose_id, const uint32_t ref_cam_id,
                       const bool is_active, const int external_id = -1) {
    assert(ref_pose_id < poses_.size());
    Landmark lm;
    lm.external_id = external_id;
    lm.x_w = x_w;
    lm.ref_pose_id = ref_pose_id;
    lm.ref_cam_id = ref_cam_id;
    lm.is_active = is_active;
    lm.id = (uint32_t)landmarks_.size();
    lm.opt_id = is_active ? num_active_landmarks_++ : UINT_MAX;
    landmarks_.push_back(lm);
    structure_dirty_ = true;
    return lm.id;
  }

  // reference :377-407 — covariance (not information) in, cov^-1 kept
  uint32_t AddUnaryConstraint(const uint32_t pose_id, const SE3t& t_wv, Matrix6t covariance,
                              bool use_rotation = true) {
    assert(pose_id < poses_.size());
    if (!use_rotation) covariance(3, 3) = covariance(4, 4) = covariance(5, 5) = 1.0;
    const Matrix6t ci = hostmath::inverse6(covariance);
    const uint32_t id = (uint32_t)un_pose_.size();
    un_pose_.push_back(pose_id);
    double t7[7]; t_wv.to7(t7);
    un_t_.insert(un_t_.end(), t7, t7 + 7);
    un_cov_inv_.insert(un_cov_inv_.end(), ci.m, ci.m + 36);
    un_rot_.push_back(use_rotation ? 1 : 0);
    poses_[pose_id].num_unary_residuals++;
    structure_dirty_ = true;
    return id;
  }

  // reference :410-422
  uint32_t AddBinaryConstraint(const uint32_t pose1_id, const uint32_t pose2_id, const SE3t& t_12,
                               Scalar weight = 1.0, bool use_rotation = true) {
    return AddBinaryConstraint(pose1_id, pose2_id, t_12, Matrix6t::Identity(), weight, use_rotation);
  }
  // reference :425-456
  uint32_t AddBinaryConstraint(const uint32_t pose1_id, const uint32_t pose2_id, const SE3t& t_12,
                               Matrix6t covariance, Scalar weight = 1.0, bool use_rotation = true) {
    assert(pose1_id < poses_.size() && pose2_id < poses_.size());
    const Matrix6t ci = hostmath::inverse6(covariance);
    const Matrix6t cis = hostmath::sqrt_spd6(ci);
    const uint32_t id = (uint32_t)bin_p1_.size();
    bin_p1_.push_back(pose1_id); bin_p2_.push_back(pose2_id);
    double t7[7]; t_12.to7(t7);
    bin_t_.insert(bin_t_.end(), t7, t7 + 7);
    bin_cov_inv_.insert(bin_cov_inv_.end(), ci.m, ci.m + 36);
    bin_cov_inv_sqrt_.insert(bin_cov_inv_sqrt_.end(), cis.m, cis.m + 36);
    bin_w_.push_back(weight);
    bin_rot_.push_back(use_rotation ? 1 : 0);
    poses_[pose1_id].num_binary_residuals++;
    poses_[pose2_id].num_binary_residuals++;
    structure_dirty_ = true;
    return id;
  }

  // reference :459-513 — returns (uint32_t)-1 when the observation comes from the
  // landmark's privileged frame (inverse-depth landmarks only)
  uint32_t AddProjectionResidual(const Vector2t z, const uint32_t meas_pose_id,
                                 const uint32_t landmark_id, const uint32_t cam_id,
                                 const Scalar weight = 1.0) {
    assert(landmark_id < landmarks_.size() && meas_pose_id < poses_.size());
    Landmark& lm = landmarks_[landmark_id];
    const uint32_t ref_id = lm.ref_pose_id;
    if (meas_pose_id == ref_id && cam_id == lm.ref_cam_id) lm.z_ref = z;
    const bool diff_poses = meas_pose_id != ref_id;
    if (!(diff_poses || cam_id != lm.ref_cam_id || LmSize != 1)) return (uint32_t)-1;
    const uint32_t res_id = (uint32_t)pr_pose_.size();
    lm.num_proj_residuals++;
    if (diff_poses || LmSize != 1) {
      poses_[meas_pose_id].num_proj_residuals++;
      if (LmSize == 1) poses_[ref_id].num_proj_residuals++;
    }
    pr_z_.push_back(z[0]); pr_z_.push_back(z[1]);
    pr_pose_.push_back(meas_pose_id); pr_lm_.push_back(landmark_id); pr_cam_.push_back(cam_id);
    pr_w_.push_back(weight);
    if (!poses_[ref_id].is_active && poses_[meas_pose_id].is_active)
      conditioning_proj_residuals_.push_back(res_id);
    structure_dirty_ = true;
    return res_id;
  }

  // reference :516-546
  uint32_t AddImuResidual(const uint32_t pose1_id, const uint32_t pose2_id,
                          const std::vector<ImuMeasurement>& imu_meas, const Scalar weight = 1.0) {
    assert(pose1_id < poses_.size() && pose2_id < poses_.size());
    const uint32_t id = (uint32_t)imu_p1_.size();
    imu_p1_.push_back(pose1_id); imu_p2_.push_back(pose2_id);
    for (const ImuMeasurement& m : imu_meas) {
      for (int i = 0; i < 3; ++i) imu_meas_.push_back(m.w[i]);
      for (int i = 0; i < 3; ++i) imu_meas_.push_back(m.a[i]);
      imu_meas_.push_back(m.time);
    }
    imu_ptr_.push_back((uint32_t)(imu_meas_.size() / 7));
    imu_w_.push_back(weight);
    poses_[pose1_id].num_inertial_residuals++;
    poses_[pose2_id].num_inertial_residuals++;
    if (!poses_[pose1_id].is_active && poses_[pose2_id].is_active)
      conditioning_inertial_residuals_.push_back(id);
    structure_dirty_ = true;
    return id;
  }

  // The one non-inline member of the reference (BundleAdjuster.cpp:278-705).
  void Solve(const uint32_t uMaxIter, const Scalar gn_damping = 1.0,
             const bool error_increase_allowed = false);

  void SetRootPoseId(const uint32_t id) { root_pose_id_ = id; }
  uint32_t GetRootPoseId() { return root_pose_id_; }
  bool IsTranslationEnabled() { return true; }
  uint32_t GetNumPoses() const { return (uint32_t)poses_.size(); }
  uint32_t GetNumImuResiduals() const { return (uint32_t)imu_p1_.size(); }
  uint32_t GetNumProjResiduals() const { return (uint32_t)pr_pose_.size(); }
  // reference :568-571.  The residual vectors and Huber weights are fetched from the device on
  // the first call after a Solve() (one copy of 3 doubles per residual), then served from the host.
  // reference :563-565.  The view is rebuilt on every call (valid until the next one); `residual` is
  // read back from the device after a Solve() (ba_hip_get_imu_residuals), zero before.
  const ImuResidual& GetImuResidual(const uint32_t id) const {
    assert(id < imu_p1_.size());
    if (imu_view_dirty_ && engine_ && !imu_p1_.empty() && uploaded_once_ && !structure_dirty_) {
      imu_view_r_.assign(15 * imu_p1_.size(), 0.0);
      if (ba_hip_get_imu_residuals(engine_, imu_view_r_.data()) == 0) imu_view_dirty_ = false;
    }
    ImuResidual& r = imu_view_;
    r.residual_id = id;
    r.residual_offset = id * ImuResidual::kResSize;
    r.pose1_id = imu_p1_[id]; r.pose2_id = imu_p2_[id];
    r.orig_weight = r.weight = imu_w_[id];
    r.measurements.clear();
    for (uint32_t m = imu_ptr_[id]; m < imu_ptr_[id + 1]; ++m) {
      ImuMeasurement meas;
      for (int i = 0; i < 3; ++i) { meas.w[i] = imu_meas_[7 * (size_t)m + i]; meas.a[i] = imu_meas_[7 * (size_t)m + 3 + i]; }
      meas.time = imu_meas_[7 * (size_t)m + 6];
      r.measurements.push_back(meas);
    }
    const bool have = !imu_view_dirty_ && !structure_dirty_ && imu_view_r_.size() == 15 * imu_p1_.size();
    for (int i = 0; i < 15; ++i) r.residual[i] = have ? (Scalar)imu_view_r_[15 * (size_t)id + i] : (Scalar)0;
    return r;
  }
  const ProjectionResidual& GetProjectionResidual(uint32_t id) const {
    if (proj_view_dirty_ && engine_ && !pr_pose_.empty() && uploaded_once_ && !structure_dirty_) {
      proj_view_r_.assign(2 * pr_pose_.size(), 0.0);
      proj_view_w_.assign(pr_pose_.size(), 0.0);
      if (ba_hip_get_proj_residuals(engine_, proj_view_r_.data()) == 0 &&
          ba_hip_get_proj_weights(engine_, proj_view_w_.data()) == 0)
        proj_view_dirty_ = false;
    }
    ProjectionResidual& r = proj_view_;
    r.residual_id = id;
    r.residual_offset = id * ProjectionResidual::kResSize;
    r.z[0] = pr_z_[2 * (size_t)id]; r.z[1] = pr_z_[2 * (size_t)id + 1];
    r.x_meas_id = pr_pose_[id]; r.landmark_id = pr_lm_[id]; r.cam_id = pr_cam_[id];
    r.x_ref_id = landmarks_[pr_lm_[id]].ref_pose_id;
    r.orig_weight = pr_w_[id];
    const bool have = !proj_view_dirty_ && !structure_dirty_ && proj_view_w_.size() == pr_pose_.size();
    r.weight = have ? proj_view_w_[id] : pr_w_[id];
    r.residual[0] = have ? proj_view_r_[2 * (size_t)id] : 0.0;
    r.residual[1] = have ? proj_view_r_[2 * (size_t)id + 1] : 0.0;
    r.mahalanobis_distance = (r.residual[0] * r.residual[0] + r.residual[1] * r.residual[1]) * r.weight;
    return r;
  }
  uint32_t GetNumLandmarks() const { return (uint32_t)landmarks_.size(); }
  uint32_t GetNumUnaryResiduals() const { return (uint32_t)un_pose_.size(); }
  const ImuCalibration& GetImuCalibration() const { EnsureHostState(); return imu_; }
  void SetImuCalibration(const ImuCalibration& calib) { imu_ = calib; }

  const Pose& GetPose(const uint32_t id) const {  // reference :573-582
    EnsureHostState();
    if (id >= poses_.size()) {
      std::cerr << "Attempted to get pose with id " << id << " from BA.  when poses_.size() is only "
                << poses_.size() << "Aborting..." << std::endl;
      throw 0;
    }
    return poses_[id];
  }
  const Landmark& GetLandmarkObj(const uint32_t id) const { EnsureHostState(); return landmarks_[id]; }
  const Vector4t& GetLandmark(const uint32_t id) const { EnsureHostState(); return landmarks_[id].x_w; }
  bool IsLandmarkReliable(const uint32_t id) const { EnsureHostState(); return landmarks_[id].is_reliable; }
  double LandmarkOutlierRatio(const uint32_t id) const {  // reference BundleAdjuster.cpp:1805-1812
    EnsureHostState();
    const Landmark& l = landmarks_[id];
    return l.num_proj_residuals == 0 ? 0 : (double)l.num_outlier_residuals / l.num_proj_residuals;
  }
  void GetErrors(Scalar& proj_error, Scalar& unary_error, Scalar& binary_error, Scalar& inertial_error) {
    proj_error = proj_error_; unary_error = unary_error_;
    binary_error = binary_error_; inertial_error = inertial_error_;
  }
  const SolutionSummary<Scalar>& GetSolutionSummary() const { return summary_; }
  Options<Scalar>& options() { return options_; }
  // extension (not in the reference, where the parameters can only be given to AddPose): replace
  // the camera intrinsics stored on a pose; used by the flat C API
  template <typename CamParams>
  void SetPoseCamParams(const uint32_t pose_id, const CamParams& cam_params) {
    assert(pose_id < poses_.size());
    poses_[pose_id].cam_params.clear();
    for (size_t i = 0; i < (size_t)cam_params.size(); ++i) poses_[pose_id].cam_params.push_back((Scalar)cam_params[i]);
  }
  const std::shared_ptr<Rig<Scalar>> rig() const { return rig_; }

  // reference :608-631 (the rotation flag masks indices 2,4,5 — kept as is)
  void RegularizePose(uint32_t pose_id, bool translation, bool gravity, bool bias, bool rotation) {
    EnsureHostState();
    Pose& pose = poses_[pose_id];
    pose.is_param_mask_used = true;
    pose.param_mask.assign(kPoseDim, true);
    if (translation) pose.param_mask[0] = pose.param_mask[1] = pose.param_mask[2] = false;
    if (rotation) pose.param_mask[2] = pose.param_mask[4] = pose.param_mask[5] = false;
    if (gravity) pose.param_mask[GetGravityRegularizationDimension(pose_id)] = false;
    if (bias && kBiasInState)
      for (int i = 9; i < 15; ++i) pose.param_mask[i] = false;
  }

  // ---- additions for tests / benchmarks (not in the reference) ------------------------
  ba_hip_engine* engine() { return engine_; }
  const Delta& GetLastStep() const {  // fetched from the device on first use after a Solve()
    if (last_step_stale_ && engine_) {
      const uint32_t n = ba_hip_num_pose_params(engine_), nl = ba_hip_num_lm_params(engine_);
      const uint32_t nk = ba_hip_num_calib_params(engine_);
      last_step_.delta_p.assign(n + nk, 0); last_step_.delta_l.assign(nl, 0);
      ba_hip_get_step(engine_, last_step_.delta_p.data(), last_step_.delta_l.data());
      last_step_.delta_k.assign(last_step_.delta_p.begin() + n, last_step_.delta_p.end());  // the tail (:766-769)
      last_step_.delta_p.resize(n);
      last_step_stale_ = false;
    }
    return last_step_;
  }
  Scalar trust_region_size() const { return trust_region_size_; }
  const ba_hip_timers& GetLastTimers() const { return last_timers_; }
  uint32_t iterations_run() const { return iterations_run_; }
  // multi-GPU: every rank holds all poses and its landmark shard; sums over shards go
  // through this hook (include/ba_hip.h)
  void SetAllReduce(ba_hip_allreduce_fn fn, void* ctx, int rank, int nranks) {
    allreduce_ = fn; allreduce_ctx_ = ctx; rank_ = rank; nranks_ = nranks;
    mask_counts_dirty_ = true;
  }
  // ... and, on top of SetAllReduce, the collectives hook (broadcast / reduce-scatter / send / receive,
  // include/ba_hip.h: ba_hip_set_collectives): the reduced solve is then distributed instead of replicated.
  // For foreign communicators and the thread-emulated tests; with SetCommunicator neither hook is needed.
  void SetCollectives(ba_hip_collective_fn fn, void* ctx) { collectives_ = fn; collectives_ctx_ = ctx; collectives_dirty_ = true; }
  // multi-GPU without hooks: the engine's own RCCL communicator over xGMI (include/ba_hip.h: ba_hip_comm_*).
  // One process per GPU; rank 0 creates the 128-byte id with CreateCommunicatorId and hands it to the other
  // ranks out of band (a file, MPI, a socket); every rank then calls SetCommunicator before Solve().  The
  // next Solve() joins the communicator (collective over all ranks: they must all call Solve()).  Every
  // rank holds all poses and its landmark shard, pose-pose residuals live on rank 0, and
  //   distributed_solve = true   the reduced system is reduce-scattered onto the owners of its tile blocks
  //                              and factorised by all GPUs (ba_amd/csrc/dist_plan.h);
  //   distributed_solve = false  it is all-reduced and factorised by every rank (replicated).
  // The reference (single process) has no counterpart; replaces the loop body of BundleAdjuster.cpp:298-663
  // across the GPUs of a node.
  static bool CreateCommunicatorId(void* id128) { return ba_hip_comm_unique_id(id128) == 0; }
  void SetCommunicator(const void* id128, int rank, int nranks, bool distributed_solve = true) {
    std::memcpy(comm_id_, id128, sizeof(comm_id_));
    comm_set_ = true; comm_dirty_ = true; comm_dist_ = distributed_solve;
    allreduce_ = nullptr; allreduce_ctx_ = nullptr; rank_ = rank; nranks_ = nranks;
    mask_counts_dirty_ = true;
  }
  void ClearCommunicator() {
    if (comm_set_ && engine_) ba_hip_comm_destroy(engine_);
    comm_set_ = false; comm_dirty_ = false; rank_ = 0; nranks_ = 1;
    mask_counts_dirty_ = true;
  }
  bool SolveIsDistributed() const { return engine_ && ba_hip_solve_is_distributed(engine_) != 0; }

 private:
  uint32_t GetGravityRegularizationDimension(uint32_t pose_id) {  // reference :634-652
    const Matrix3t rot = poses_[pose_id].t_wp.rotationMatrix();
    double max_dot = 0;
    uint32_t max_dim = 0;
    for (uint32_t ii = 0; ii < 3; ++ii) {
      double dot = 0;
      for (int r = 0; r < 3; ++r) dot += rot(r, ii) * imu_.g_vec[r];
      dot = std::fabs(dot);
      if (dot > max_dot) { max_dot = dot; max_dim = ii; }
    }
    return max_dim + 3;
  }

  void ReleaseEngine() {
    if (engine_) ba_hip_destroy(engine_);
    engine_ = nullptr;
  }
  bool Check(int rc, const char* what) {
    if (rc >= 0) return true;
    std::cerr << "ba::BundleAdjuster: " << what << " failed (" << rc << "): "
              << (engine_ ? ba_hip_last_error(engine_) : "no engine") << std::endl;
    summary_.result = SolverError;
    return false;
  }
  bool UploadProblem();
  bool SyncEngine();
  // The solution stays on the device when Solve() returns; the host copies (poses_, landmarks_,
  // imu_ biases) are refreshed by the first getter that needs them — a SLAM loop that calls
  // Solve(1) repeatedly does not pay a PCIe round trip of the whole state per call.
  void EnsureHostState() const {
    if (host_state_stale_) const_cast<BundleAdjuster*>(this)->DownloadState();
  }
  void ComputeMasks(std::vector<uint16_t>& masks);
  void WriteReducedCameraMatrix();
  bool SolveInternal(const Scalar gn_damping, const bool error_increase_allowed, const bool use_dogleg);
  // reference BundleAdjuster.cpp:771-784 (inside CalculateGn, only with active poses)
  bool FetchCalibrationMarginals() {
    if (kCalibDim == 0 || !options_.calculate_calibration_marginals || num_active_poses_ == 0) return true;
    summary_.calibration_marginals = MatX((int)kCalibDim, (int)kCalibDim);
    return Check(ba_hip_get_calibration_marginals(engine_, summary_.calibration_marginals.data()),
                 "ba_hip_get_calibration_marginals");
  }
  bool DownloadState();

  // ---- problem graph, flat (ids = insertion order, as the reference returns them) ----
  std::shared_ptr<Rig<Scalar>> rig_;
  std::vector<Pose> poses_;
  std::vector<Landmark> landmarks_;
  std::vector<double> pr_z_, pr_w_;
  mutable std::vector<double> proj_view_r_, proj_view_w_;  // GetProjectionResidual cache
  mutable ProjectionResidual proj_view_;
  mutable bool proj_view_dirty_ = true;
  mutable std::vector<double> imu_view_r_;  // GetImuResidual cache
  mutable ImuResidual imu_view_;
  mutable bool imu_view_dirty_ = true;
  bool imu_cov_reset_ = true;  // calculate_inertial_covariance_once: next upload starts afresh
  bool uploaded_once_ = false;
  std::vector<uint32_t> pr_pose_, pr_lm_, pr_cam_;
  std::vector<uint32_t> un_pose_; std::vector<double> un_t_, un_cov_inv_; std::vector<uint8_t> un_rot_;
  std::vector<uint32_t> bin_p1_, bin_p2_; std::vector<double> bin_t_, bin_cov_inv_, bin_cov_inv_sqrt_, bin_w_;
  std::vector<uint8_t> bin_rot_;
  std::vector<uint32_t> imu_p1_, imu_p2_, imu_ptr_; std::vector<double> imu_meas_, imu_w_;
  std::vector<uint32_t> conditioning_proj_residuals_, conditioning_inertial_residuals_;

  ImuCalibration imu_;
  Options<Scalar> options_;
  SolutionSummary<Scalar> summary_;
  Scalar trust_region_size_ = kTrustRegionAuto;
  Scalar proj_error_ = 0, binary_error_ = 0, unary_error_ = 0, inertial_error_ = 0;
  uint32_t root_pose_id_ = 0, num_active_poses_ = 0, num_active_landmarks_ = 0;
  uint32_t iterations_run_ = 0;
  bool structure_dirty_ = true;
  bool host_state_stale_ = false;          // the device holds a newer state than poses_ / landmarks_
  uint32_t uploaded_poses_ = 0, uploaded_landmarks_ = 0;  // sizes of the engine's copy of the graph
  std::vector<double> un_scale_seen_;      // cumulative Huber scale already folded into un_cov_inv_
  std::vector<uint64_t> mask_counts_;      // per pose: proj, binary, unary, inertial residual counts (global); + #unary
  bool mask_counts_dirty_ = true;
  std::vector<uint16_t> masks_uploaded_;   // what the engine currently holds
  mutable Delta last_step_;
  mutable bool last_step_stale_ = false;
  ba_hip_timers last_timers_ = {};

  ba_hip_engine* engine_ = nullptr;
  int engine_device_ = -1;
  ba_hip_allreduce_fn allreduce_ = nullptr;
  void* allreduce_ctx_ = nullptr;
  int rank_ = 0, nranks_ = 1;
  ba_hip_allreduce_fn engine_allreduce_ = nullptr;  // what the engine currently has installed
  void* engine_allreduce_ctx_ = nullptr;
  bool engine_per_pose_cam_ = false;
  ba_hip_collective_fn collectives_ = nullptr;   // SetCollectives
  void* collectives_ctx_ = nullptr;
  bool collectives_dirty_ = false;
  unsigned char comm_id_[128] = {};     // SetCommunicator: the RCCL unique id
  bool comm_set_ = false, comm_dirty_ = false, comm_dist_ = true;
};

template <typename Scalar>
using SelfCalBundleAdjuster = BundleAdjuster<Scalar, 1, 6, 5>;          // reference :758-759 (camera 0: a FovCamera)
template <typename Scalar>
using VisualBundleAdjuster = BundleAdjuster<Scalar, 1, 6, 0>;           // reference :760-761
template <typename Scalar>
using VisualInertialBundleAdjuster = BundleAdjuster<Scalar, 1, 15, 0>;  // reference :762-763

// ======================================================================================
// Engine creation + everything that is cheap to refresh on every Solve(): options, IMU noise,
// gravity, the all-reduce hook.  The graph itself is uploaded (and ba_hip_finalize run) only when
// an Add* / Init call changed it since the last Solve — "Solve may be called repeatedly"
// (reference BundleAdjuster.h:549-551) then costs no rebuild.
template <typename Scalar, int LmSize, int PoseSize, int CalibSize, bool DoTvs>
bool BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs>::SyncEngine() {
  if (!engine_ || engine_device_ != options_.device) {
    EnsureHostState();  // a live engine on another device holds the current solution
    ReleaseEngine();
    const int rc = ba_hip_create(LmSize, PoseSize, options_.device, nullptr, &engine_);
    if (rc != 0) {
      std::cerr << "ba::BundleAdjuster: no usable HIP device (ba_hip_create rc=" << rc
                << "); this class has no CPU solver" << std::endl;
      engine_ = nullptr;
      summary_.result = SolverError;
      return false;
    }
    if (kCalibDim > 0 && !Check(ba_hip_set_calibration(engine_, CalibSize, DoTvs ? 1 : 0), "ba_hip_set_calibration")) return false;
    engine_device_ = options_.device;
    structure_dirty_ = true;
    if (comm_set_) comm_dirty_ = true;  // a new engine joins the communicator again
    collectives_dirty_ = true;
  }
  ba_hip_options o;
  std::memset(&o, 0, sizeof(o));
  o.projection_outlier_threshold = options_.projection_outlier_threshold;
  o.use_robust_norm_for_proj_residuals = options_.use_robust_norm_for_proj_residuals;
  o.use_robust_norm_for_inertial_residuals = options_.use_robust_norm_for_inertial_residuals;
  o.use_triangular_matrices = options_.use_triangular_matrices;
  o.keep_reduced_system = options_.write_reduced_camera_matrix || options_.keep_reduced_system;  // debug tap (reference :600-627)
  o.gyro_sigma = options_.gyro_sigma; o.accel_sigma = options_.accel_sigma;
  o.gyro_bias_sigma = options_.gyro_bias_sigma; o.accel_bias_sigma = options_.accel_bias_sigma;
  o.pivot_rel_tolerance = options_.factorization_pivot_tolerance;
  if (!Check(ba_hip_set_options(engine_, &o), "ba_hip_set_options")) return false;
  if (comm_set_) {
    // native communicator: joined once per engine (a collective), the solve switch refreshed with it
    if (comm_dirty_) {
      if (!Check(ba_hip_comm_init(engine_, comm_id_, rank_, nranks_), "ba_hip_comm_init")) return false;
      if (!comm_dist_) ba_hip_set_collectives(engine_, nullptr, nullptr);  // replicated solve: all-reduce of S only
      comm_dirty_ = false;
      engine_allreduce_ = nullptr; engine_allreduce_ctx_ = nullptr;
    }
  } else {
    if (structure_dirty_ || allreduce_ != engine_allreduce_ || allreduce_ctx_ != engine_allreduce_ctx_) {
      ba_hip_set_allreduce(engine_, allreduce_, allreduce_ctx_, rank_, nranks_);
      engine_allreduce_ = allreduce_; engine_allreduce_ctx_ = allreduce_ctx_;
      collectives_dirty_ = true;
    }
    if (collectives_dirty_) {
      ba_hip_set_collectives(engine_, collectives_, collectives_ctx_);
      collectives_dirty_ = false;
    }
  }
  if (structure_dirty_) {
    EnsureHostState();  // the graph is re-marshalled from poses_ / landmarks_: they must be current
    if (!UploadProblem()) return false;
  } else if (options_.use_per_pose_cam_params != engine_per_pose_cam_) {
    structure_dirty_ = true;  // (rare) option flip: simplest is a full upload
    EnsureHostState();
    if (!UploadProblem()) return false;
  }
  // reference parallel_algos.h:189-205 (ImuResidualT::covariance_computed lives with the residual:
  // it survives Solve() calls until the problem is rebuilt by Init())
  if (!Check(ba_hip_set_inertial_covariance_once(engine_, options_.calculate_inertial_covariance_once ? 1 : 0,
                                                 imu_cov_reset_ ? 1 : 0),
             "ba_hip_set_inertial_covariance_once")) return false;
  imu_cov_reset_ = false;
  {
    // the noise diagonals come from imu_ (Init() fills them from the option sigmas; a caller may
    // have replaced them with SetImuCalibration) — reference parallel_algos.h:204,288.  Like the
    // gravity vector they take effect at ba_hip_begin_solve (no rebuild).
    double r6[6], rb6[6];
    for (int i = 0; i < 6; ++i) { r6[i] = (double)imu_.r[i]; rb6[i] = (double)imu_.r_b[i]; }
    if (!Check(ba_hip_set_imu_noise(engine_, r6, rb6), "ba_hip_set_imu_noise")) return false;
  }
  const double g[3] = {imu_.g_vec[0], imu_.g_vec[1], imu_.g_vec[2]};
  if (!Check(ba_hip_set_gravity(engine_, g), "ba_hip_set_gravity")) return false;
  return true;
}

// Marshal the graph into the engine (only when the graph changed since the last Solve).
template <typename Scalar, int LmSize, int PoseSize, int CalibSize, bool DoTvs>
bool BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs>::UploadProblem() {
  const uint32_t C = rig_->NumCams(), P = (uint32_t)poses_.size(), L = (uint32_t)landmarks_.size();
  std::vector<double> cam_p(4 * (size_t)C), cam_t(7 * (size_t)C);
  for (uint32_t c = 0; c < C; ++c) {
    const Vector4t pp = rig_->cameras_[c]->GetParams();
    for (int i = 0; i < 4; ++i) cam_p[4 * c + i] = pp[i];
    rig_->cameras_[c]->Pose().to7(&cam_t[7 * (size_t)c]);
  }
  std::vector<double> pt(7 * (size_t)P), pv(3 * (size_t)P), pb(6 * (size_t)P);
  std::vector<uint8_t> pa(P);
  for (uint32_t p = 0; p < P; ++p) {
    poses_[p].t_wp.to7(&pt[7 * (size_t)p]);
    for (int i = 0; i < 3; ++i) pv[3 * (size_t)p + i] = poses_[p].v_w[i];
    for (int i = 0; i < 6; ++i) pb[6 * (size_t)p + i] = poses_[p].b[i];
    pa[p] = poses_[p].is_active ? 1 : 0;
  }
  std::vector<double> lx(4 * (size_t)L);
  std::vector<uint32_t> lrp(L), lrc(L);
  std::vector<uint8_t> la(L);
  for (uint32_t l = 0; l < L; ++l) {
    for (int i = 0; i < 4; ++i) lx[4 * (size_t)l + i] = landmarks_[l].x_w[i];
    lrp[l] = landmarks_[l].ref_pose_id; lrc[l] = landmarks_[l].ref_cam_id;
    la[l] = landmarks_[l].is_active ? 1 : 0;
  }
  if (!Check(ba_hip_set_cameras(engine_, C, cam_p.data(), cam_t.data()), "ba_hip_set_cameras")) return false;
  {
    std::vector<int32_t> cam_m(C);
    std::vector<double> cam_w(C);
    for (uint32_t c = 0; c < C; ++c) { cam_m[c] = rig_->cameras_[c]->Type(); cam_w[c] = rig_->cameras_[c]->Param(4); }
    if (C && !Check(ba_hip_set_camera_models(engine_, C, cam_m.data(), cam_w.data()), "ba_hip_set_camera_models")) return false;
  }
  if (!Check(ba_hip_set_poses(engine_, P, pt.data(), pv.data(), pb.data(), pa.data()), "ba_hip_set_poses")) return false;
  if (options_.use_per_pose_cam_params) {
    // reference parallel_algos.h:54-57: cam->SetParams(pose.cam_params) per residual
    std::vector<double> pc(4 * (size_t)P);
    for (uint32_t p = 0; p < P; ++p) {
      if (poses_[p].cam_params.size() != 4) {
        std::cerr << "ba::BundleAdjuster: use_per_pose_cam_params needs 4 pinhole parameters on every pose "
                     "(AddPose with cam_params); pose " << p << " has " << poses_[p].cam_params.size() << std::endl;
        summary_.result = SolverError;
        return false;
      }
      for (int i = 0; i < 4; ++i) pc[4 * (size_t)p + i] = (double)poses_[p].cam_params[i];
    }
    if (!Check(ba_hip_set_pose_cam_params(engine_, P, pc.data()), "ba_hip_set_pose_cam_params")) return false;
  } else if (!Check(ba_hip_set_pose_cam_params(engine_, 0, nullptr), "ba_hip_set_pose_cam_params")) {
    return false;
  }
  if (!Check(ba_hip_set_landmarks(engine_, L, lx.data(), lrp.data(), lrc.data(), la.data()), "ba_hip_set_landmarks")) return false;
  if (kCamParamsInCalib) {  // dTransfer_dparams is taken at the landmark's reference pixel (parallel_algos.h:115-118)
    std::vector<double> zr(2 * (size_t)L);
    for (uint32_t l = 0; l < L; ++l) { zr[2 * (size_t)l] = landmarks_[l].z_ref[0]; zr[2 * (size_t)l + 1] = landmarks_[l].z_ref[1]; }
    if (!Check(ba_hip_set_landmark_ref_pixels(engine_, L, zr.data()), "ba_hip_set_landmark_ref_pixels")) return false;
  }
  if (!Check(ba_hip_set_projection_residuals(engine_, (uint32_t)pr_pose_.size(), pr_z_.data(), pr_pose_.data(),
                                             pr_lm_.data(), pr_cam_.data(), pr_w_.data()),
             "ba_hip_set_projection_residuals")) return false;
  if (!Check(ba_hip_set_conditioning_residuals(engine_, (uint32_t)conditioning_proj_residuals_.size(),
                                               conditioning_proj_residuals_.data()),
             "ba_hip_set_conditioning_residuals")) return false;
  if (!Check(ba_hip_set_unary_residuals(engine_, (uint32_t)un_pose_.size(), un_pose_.data(), un_t_.data(),
                                        un_cov_inv_.data(), un_rot_.data()), "ba_hip_set_unary_residuals")) return false;
  if (!Check(ba_hip_set_binary_residuals(engine_, (uint32_t)bin_p1_.size(), bin_p1_.data(), bin_p2_.data(),
                                         bin_t_.data(), bin_cov_inv_.data(), bin_cov_inv_sqrt_.data(),
                                         bin_w_.data(), bin_rot_.data()), "ba_hip_set_binary_residuals")) return false;
  if (!Check(ba_hip_set_imu_residuals(engine_, (uint32_t)imu_p1_.size(), imu_p1_.data(), imu_p2_.data(),
                                      imu_ptr_.data(), imu_meas_.data(), imu_w_.data()),
             "ba_hip_set_imu_residuals")) return false;
  if (!Check(ba_hip_finalize(engine_), "ba_hip_finalize")) return false;
  structure_dirty_ = false;
  engine_per_pose_cam_ = options_.use_per_pose_cam_params;
  uploaded_poses_ = P; uploaded_landmarks_ = L;
  un_scale_seen_.assign(un_pose_.size(), 1.0);  // the engine's cumulative Huber scales restart at 1
  mask_counts_dirty_ = true;
  masks_uploaded_.clear();
  return true;
}

// The reference's interchange dump (BundleAdjuster.cpp:600-616, Utils.h:66): the reduced camera
// matrix s_, the reduced right-hand side and the dense images of j_pr_, r_pr_, j_l_ as CSV in
// Eigen's FullPrecision format (", " between coefficients, one row per line) into s.txt, rhs.txt,
// j_pr.txt, r_pr.txt, j_l.txt of the working directory, so that a build of the original can be
// diffed against this one (ba_amd/dumps.py loads and cross-checks them).  j_pr.txt is
// (2 x residuals) x (6 x active poses), j_l.txt (2 x residuals) x (LmSize x active landmarks) — dense:
// a debug tool for small problems, as in the reference.  With DoTvs s.txt / rhs.txt are the bordered
// (n + 6) system and j_kpr.txt ((2 x residuals) x 6) and jt_kpr_j_kpr.txt (6 x 6) follow (:619-626).
template <typename Scalar, int LmSize, int PoseSize, int CalibSize, bool DoTvs>
void BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs>::WriteReducedCameraMatrix() {
  const uint32_t n_pose = ba_hip_num_pose_params(engine_), n = n_pose + ba_hip_num_calib_params(engine_);
  std::cerr << "Writing reduced camera matrix for " << n_pose << " pose parameters and " << kCalibDim
            << " calib  parameters " << std::endl;
  auto put_row = [](FILE* f, const double* v, size_t cnt) {
    for (size_t c = 0; c < cnt; ++c) std::fprintf(f, c ? ", %.17g" : "%.17g", v[c]);
    std::fputc('\n', f);
  };
  if (n > 0) {
    std::vector<double> sm((size_t)n * n), rhs(n);
    if (!Check(ba_hip_get_S(engine_, sm.data()), "ba_hip_get_S")) return;
    if (!Check(ba_hip_get_rhs(engine_, rhs.data(), nullptr, nullptr), "ba_hip_get_rhs")) return;
    if (FILE* f = std::fopen("s.txt", "w")) {
      for (uint32_t r = 0; r < n; ++r) put_row(f, &sm[(size_t)r * n], n);
      std::fclose(f);
    }
    if (FILE* f = std::fopen("rhs.txt", "w")) {
      for (uint32_t r = 0; r < n; ++r) put_row(f, &rhs[r], 1);
      std::fclose(f);
    }
  }
  const size_t O = pr_pose_.size();
  if (O == 0 || LmSize == 0) return;
  constexpr int LL = LmSize > 0 ? LmSize : 1;
  std::vector<double> jm(12 * O), jr(12 * O), jl(2 * LL * O), rr(2 * O);
  if (!Check(ba_hip_get_proj_jacobians(engine_, jm.data(), jr.data(), jl.data(), rr.data()), "ba_hip_get_proj_jacobians")) return;
  const size_t pcols = (size_t)kPrPoseDim * num_active_poses_, lcols = (size_t)LL * num_active_landmarks_;
  FILE* fp = std::fopen("j_pr.txt", "w");
  FILE* fl = std::fopen("j_l.txt", "w");
  FILE* fr = std::fopen("r_pr.txt", "w");
  std::vector<double> prow(std::max<size_t>(pcols, 1)), lrow(std::max<size_t>(lcols, 1));
  for (size_t a = 0; a < O && fp && fl && fr; ++a) {
    const Landmark& lm = landmarks_[pr_lm_[a]];
    const Pose& pm = poses_[pr_pose_[a]];
    const Pose& pr = poses_[lm.ref_pose_id];
    // blocks are inserted for active poses of "listed" residuals only (BundleAdjuster.h:489-497,
    // BundleAdjuster.cpp:1613-1643, 1694-1720) and for active landmarks (:1788-1797)
    const bool listed = LmSize != 1 || pr_pose_[a] != lm.ref_pose_id;
    for (int k = 0; k < 2; ++k) {
      std::fill(prow.begin(), prow.end(), 0.0);
      std::fill(lrow.begin(), lrow.end(), 0.0);
      if (listed && pm.is_active)
        for (int c = 0; c < 6; ++c) prow[(size_t)pm.opt_id * kPrPoseDim + c] += jm[12 * a + 6 * k + c];
      if (LmSize == 1 && listed && pr.is_active)
        for (int c = 0; c < 6; ++c) prow[(size_t)pr.opt_id * kPrPoseDim + c] += jr[12 * a + 6 * k + c];
      if (lm.is_active)
        for (int c = 0; c < LL; ++c) lrow[(size_t)lm.opt_id * LL + c] = jl[2 * LL * a + LL * k + c];
      if (pcols) put_row(fp, prow.data(), pcols);
      if (lcols) put_row(fl, lrow.data(), lcols);
      put_row(fr, &rr[2 * a + k], 1);
    }
  }
  if (fp) std::fclose(fp);
  if (fl) std::fclose(fl);
  if (fr) std::fclose(fr);
  if (kCalibDim > 0) {  // with CalibSize 4 / 5 the last two / the last of the six columns are zero
    std::vector<double> jk(12 * O), jtj(36, 0.0);
    if (!Check(ba_hip_get_calib_jacobians(engine_, jk.data()), "ba_hip_get_calib_jacobians")) return;
    if (FILE* f = std::fopen("j_kpr.txt", "w")) {
      for (size_t r = 0; r < 2 * O; ++r) put_row(f, &jk[6 * r], kCalibDim);
      std::fclose(f);
    }
    for (size_t r = 0; r < 2 * O; ++r)
      for (int x = 0; x < 6; ++x)
        for (int y = 0; y < 6; ++y) jtj[6 * x + y] += jk[6 * r + x] * jk[6 * r + y];
    if (FILE* f = std::fopen("jt_kpr_j_kpr.txt", "w")) {
      for (uint32_t x = 0; x < kCalibDim; ++x) put_row(f, &jtj[6 * x], kCalibDim);
      std::fclose(f);
    }
  }
}

template <typename Scalar, int LmSize, int PoseSize, int CalibSize, bool DoTvs>
void BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs>::ComputeMasks(std::vector<uint16_t>& masks) {
  // Residual counts per pose.  With landmark shards (SetAllReduce / native communicator) every rank
  // holds only its share of the projection residuals and rank 0 the pose-pose ones, but the masks
  // must be the same on all ranks — they follow from the GLOBAL counts: summed once per graph.
  const size_t P = poses_.size();
  if (mask_counts_.size() != 4 * P + 1 || mask_counts_dirty_) {
    mask_counts_.assign(4 * P + 1, 0);
    for (size_t p = 0; p < P; ++p) {
      mask_counts_[4 * p + 0] = poses_[p].num_proj_residuals;
      mask_counts_[4 * p + 1] = poses_[p].num_binary_residuals;
      mask_counts_[4 * p + 2] = poses_[p].num_unary_residuals;
      mask_counts_[4 * p + 3] = poses_[p].num_inertial_residuals;
    }
    mask_counts_[4 * P] = un_pose_.size();
    if (nranks_ > 1 && engine_)
      Check(ba_hip_allreduce_host(engine_, mask_counts_.data(), mask_counts_.size(), 1), "ba_hip_allreduce_host");
    mask_counts_dirty_ = false;
  }
  const uint64_t* cnt = mask_counts_.data();
  // :1240-1259 — note the loop stops at the first inactive pose
  bool are_all_active = true;
  for (size_t p = 0; p < P; ++p) {
    Pose& pose = poses_[p];
    if (!pose.is_active) { are_all_active = false; break; }
    if (cnt[4 * p] == 0 && cnt[4 * p + 1] == 0 && cnt[4 * p + 2] == 0 && cnt[4 * p + 3] == 0) {
      pose.is_param_mask_used = true;
      pose.param_mask.assign(kPoseDim, false);
    }
  }
  if (kVelInState) {  // :1263-1279
    for (size_t p = 0; p < P; ++p) {
      Pose& pose = poses_[p];
      if (cnt[4 * p + 3] == 0 && pose.is_active) {
        pose.is_param_mask_used = true;
        pose.param_mask.assign(kPoseDim, true);
        for (uint32_t i = 6; i < kPoseDim; ++i) pose.param_mask[i] = false;
      }
    }
  }
  if (are_all_active && cnt[4 * P] == 0 && options_.enable_auto_regularization && !poses_.empty()) {
    Pose& root = poses_[root_pose_id_];  // :1285-1330
    root.is_param_mask_used = true;
    root.param_mask.assign(kPoseDim, true);
    root.param_mask[0] = root.param_mask[1] = root.param_mask[2] = false;
    if (kBiasInState && options_.regularize_biases_in_batch)
      for (int i = 9; i < 15; ++i) root.param_mask[i] = false;
    if (!kVelInState) root.param_mask[3] = root.param_mask[4] = root.param_mask[5] = false;
    else root.param_mask[GetGravityRegularizationDimension(root_pose_id_)] = false;
  }
  masks.assign(poses_.size(), 0);
  for (size_t p = 0; p < poses_.size(); ++p) {
    const Pose& pose = poses_[p];
    if (!pose.is_param_mask_used) continue;
    for (uint32_t i = 0; i < kPoseDim && i < pose.param_mask.size(); ++i)
      if (!pose.param_mask[i]) masks[p] |= (uint16_t)(1u << i);
  }
}

template <typename Scalar, int LmSize, int PoseSize, int CalibSize, bool DoTvs>
bool BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs>::DownloadState() {
  host_state_stale_ = false;
  if (!engine_) return false;
  // the engine's copy of the graph may be older than poses_ / landmarks_ (Add* calls since the
  // last Solve append at the end): only the objects it knows are refreshed
  const uint32_t P = std::min<uint32_t>(uploaded_poses_, (uint32_t)poses_.size());
  const uint32_t L = std::min<uint32_t>(uploaded_landmarks_, (uint32_t)landmarks_.size());
  std::vector<double> pt(7 * (size_t)uploaded_poses_), pv(3 * (size_t)uploaded_poses_), pb(6 * (size_t)uploaded_poses_),
      lx(4 * (size_t)uploaded_landmarks_);
  std::vector<uint8_t> rel(uploaded_landmarks_);
  std::vector<uint32_t> outl(uploaded_landmarks_);
  if (!Check(ba_hip_get_poses(engine_, pt.data(), pv.data(), pb.data()), "ba_hip_get_poses")) return false;
  if (!Check(ba_hip_get_landmarks(engine_, lx.data()), "ba_hip_get_landmarks")) return false;
  if (!Check(ba_hip_get_landmark_flags(engine_, rel.data(), outl.data()), "ba_hip_get_landmark_flags")) return false;
  for (uint32_t p = 0; p < P; ++p) {
    poses_[p].t_wp = SE3::from7(&pt[7 * (size_t)p]);
    for (int i = 0; i < 3; ++i) poses_[p].v_w[i] = pv[3 * (size_t)p + i];
    for (int i = 0; i < 6; ++i) poses_[p].b[i] = pb[6 * (size_t)p + i];
  }
  for (uint32_t l = 0; l < L; ++l) {
    for (int i = 0; i < 4; ++i) landmarks_[l].x_w[i] = lx[4 * (size_t)l + i];
    // is_reliable persists across Solve() calls in the reference (the Landmark objects live on,
    // BundleAdjuster.cpp:127-134); so do the engine's flags while the graph is unchanged, and a
    // re-upload starts them afresh — hence the conjunction with the host's copy
    landmarks_[l].is_reliable = landmarks_[l].is_reliable && rel[l] != 0;
    landmarks_[l].num_outlier_residuals = outl[l];
  }
  if (kBiasInState && P > 0 && P == poses_.size()) {  // :666-669
    for (int i = 0; i < 3; ++i) { imu_.b_g[i] = poses_.back().b[i]; imu_.b_a[i] = poses_.back().b[3 + i]; }
  }
  return true;
}

// SolveInternal, reference BundleAdjuster.cpp:838-1161: host control flow only.
template <typename Scalar, int LmSize, int PoseSize, int CalibSize, bool DoTvs>
bool BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs>::SolveInternal(
    const Scalar gn_damping, const bool error_increase_allowed, const bool use_dogleg) {
  ba_hip_errors pre, post;
  ba_hip_step_norms norms;
  auto total = [](const ba_hip_errors& e) {
    return e.proj_error + e.inertial_error + e.binary_error + e.unary_error;
  };
  auto accept = [&](const ba_hip_errors& e) {
    proj_error_ = e.proj_error; unary_error_ = e.unary_error;
    binary_error_ = e.binary_error; inertial_error_ = e.inertial_error;
  };
  if (use_dogleg) {
    bool gn_computed = false;
    ba_hip_dogleg_scalars s;
    if (!Check(ba_hip_dogleg_terms(engine_, 0, &s), "ba_hip_dogleg_terms")) return false;
    // the calibration terms (rhs_k, delta_k) are zero without DoTvs
    const Scalar numerator = s.rhs_p_sq + s.rhs_l_sq + s.rhs_k_sq;  // :858-859
    const Scalar factor = numerator / s.j_rhs_sq;             // :919
    const Scalar sd_sq = factor * factor * numerator;         // |delta_sd|^2 over p, k and l (:1001-1003)
    // :927-929 — the steepest-descent norm leaves delta_k out
    const Scalar delta_sd_norm = std::sqrt(factor * factor * (s.rhs_p_sq + s.rhs_l_sq));
    uint32_t iteration_count = 0;
    while (1) {
      iteration_count++;
      if (iteration_count > options_.dogleg_max_inner_iterations) break;
      Scalar coef_rhs = 0, coef_gn = 0;
      if (delta_sd_norm > trust_region_size_ && trust_region_size_ != kTrustRegionAuto) {
        coef_rhs = factor * trust_region_size_ / delta_sd_norm;  // :947-950
      } else {
        if (!gn_computed) {
          {
            // CalculateGn only with active poses (:959-964); GetLandmarkDelta always (:966-967) —
            // the engine skips the factorisation itself when n == 0 and still back-substitutes
            if (num_active_poses_ > 0) summary_.result = Success;
            const int rc = ba_hip_solve_gn(engine_);
            if (rc < 0) { Check(rc, "ba_hip_solve_gn"); return false; }
            if (rc == BA_HIP_FACTORIZATION_ERROR) { summary_.result = FactorizationError; return false; }
            if (rc == BA_HIP_SOLVER_ERROR) { summary_.result = SolverError; return false; }
            if (!FetchCalibrationMarginals()) return false;
          }
          if (!Check(ba_hip_dogleg_terms(engine_, 1, &s), "ba_hip_dogleg_terms")) return false;
          gn_computed = true;
        }
        const Scalar gn_sq = s.gn_p_sq + s.gn_k_sq + s.gn_l_sq;
        const Scalar delta_gn_norm = std::sqrt(gn_sq);  // :971-973
        const bool delta_gn_good = !std::isnan(delta_gn_norm) && !std::isinf(delta_gn_norm);
        if (delta_gn_good && trust_region_size_ == kTrustRegionAuto) trust_region_size_ = delta_gn_norm;
        if (delta_gn_good && delta_gn_norm <= trust_region_size_) {
          coef_gn = 1.0;  // :985
        } else {
          // :991-1017 with sd = factor * rhs:  diff = gn - sd
          const Scalar rhs_gn = s.rhs_gn_p + s.rhs_gn_k + s.rhs_gn_l;
          const Scalar a = gn_sq - 2 * factor * rhs_gn + sd_sq;
          const Scalar b = 2 * (factor * rhs_gn - sd_sq);
          const Scalar c = sd_sq - trust_region_size_ * trust_region_size_;
          Scalar beta = 0;
          if (b * b > 4 * a * c && a > 1e-10)
            beta = (-(b * b) + std::sqrt(b * b - 4 * a * c)) / (2 * a);  // sic (:1008)
          coef_rhs = factor * (1 - beta);
          coef_gn = beta;
        }
      }
      if (!Check(ba_hip_compose_step(engine_, coef_rhs, coef_gn, &norms), "ba_hip_compose_step")) return false;
      if (!Check(ba_hip_eval_residuals(engine_, &pre), "ba_hip_eval_residuals")) return false;
      summary_.pre_solve_norm = total(pre);
      if (options_.apply_results) {
        summary_.delta_norm = norms.step_l_norm + norms.step_p_norm;  // :26
        if (!Check(ba_hip_apply_step(engine_), "ba_hip_apply_step")) return false;
      }
      if (!Check(ba_hip_eval_residuals(engine_, &post), "ba_hip_eval_residuals")) return false;
      summary_.post_solve_norm = total(post);
      if (summary_.post_solve_norm > summary_.pre_solve_norm) {
        if (options_.apply_results)
          if (!Check(ba_hip_rollback(engine_), "ba_hip_rollback")) return false;
        trust_region_size_ /= 2;
      } else {
        accept(post);
        trust_region_size_ *= 2;
        break;
      }
    }
  } else {
    {
      // :1089-1094 CalculateGn only with active poses; :1105-1106 GetLandmarkDelta always
      if (num_active_poses_ > 0) summary_.result = Success;
      const int rc = ba_hip_solve_gn(engine_);
      if (rc < 0) { Check(rc, "ba_hip_solve_gn"); return false; }
      if (rc == BA_HIP_FACTORIZATION_ERROR) { summary_.result = FactorizationError; return false; }
      if (rc == BA_HIP_SOLVER_ERROR) { summary_.result = SolverError; return false; }
      if (!FetchCalibrationMarginals()) return false;
    }
    if (!Check(ba_hip_compose_step(engine_, 0.0, gn_damping, &norms), "ba_hip_compose_step")) return false;  // :1108-1110
    if (!Check(ba_hip_eval_residuals(engine_, &pre), "ba_hip_eval_residuals")) return false;
    const Scalar prev_error = total(pre);
    if (options_.apply_results) {
      summary_.delta_norm = norms.step_l_norm + norms.step_p_norm;
      if (!Check(ba_hip_apply_step(engine_), "ba_hip_apply_step")) return false;
    }
    if (!Check(ba_hip_eval_residuals(engine_, &post), "ba_hip_eval_residuals")) return false;
    const Scalar post_error = total(post);
    if (post_error > prev_error && !error_increase_allowed) {
      if (options_.apply_results)
        if (!Check(ba_hip_rollback(engine_), "ba_hip_rollback")) return false;
      summary_.result = ErrorIncreased;
      return false;
    }
    accept(post);
  }
  return true;
}

// Solve, reference BundleAdjuster.cpp:278-705.
template <typename Scalar, int LmSize, int PoseSize, int CalibSize, bool DoTvs>
void BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs>::Solve(
    const uint32_t uMaxIter, const Scalar gn_damping, const bool error_increase_allowed) {
  if (pr_pose_.empty() && bin_p1_.empty() && un_pose_.empty() && imu_p1_.empty()) return;
  // AoS -> SoA once per graph: the graph is uploaded and the static structure
  // rebuilt only if an Add* / Init call changed it since the last Solve; otherwise the engine
  // already holds the current state
  if (!SyncEngine()) return;
  iterations_run_ = 0;
  if (!Check(ba_hip_begin_solve(engine_), "ba_hip_begin_solve")) return;  // :288-296
  const bool host_was_stale = host_state_stale_;
  host_state_stale_ = true;  // from here on the device holds the state; getters fetch it lazily (:666-678)
  std::vector<uint16_t> masks;
  for (uint32_t kk = 0; kk < uMaxIter; ++kk) {
    // masks may depend on the current root orientation (gravity axis): refresh per
    // iteration as BuildProblem does
    if (kVelInState && (kk > 0 || host_was_stale) && root_pose_id_ < uploaded_poses_) {
      std::vector<double> pt(7 * (size_t)uploaded_poses_);
      if (!Check(ba_hip_get_poses(engine_, pt.data(), nullptr, nullptr), "ba_hip_get_poses")) return;
      poses_[root_pose_id_].t_wp = SE3::from7(&pt[7 * (size_t)root_pose_id_]);
    }
    ComputeMasks(masks);
    if (masks != masks_uploaded_) {  // usually unchanged between iterations and Solve() calls
      if (!Check(ba_hip_set_pose_masks(engine_, (uint32_t)masks.size(), masks.data()), "ba_hip_set_pose_masks")) return;
      masks_uploaded_ = masks;
    }
    ba_hip_errors built;
    if (!Check(ba_hip_linearize(engine_, &built), "ba_hip_linearize")) return;  // BuildProblem .. Schur
    proj_error_ = built.proj_error; binary_error_ = built.binary_error;
    unary_error_ = built.unary_error; inertial_error_ = built.inertial_error;
    if (options_.write_reduced_camera_matrix) WriteReducedCameraMatrix();  // :600-606
    iterations_run_++;
    const bool ok = SolveInternal(gn_damping, error_increase_allowed, options_.use_dogleg);
    ba_hip_get_timers(engine_, &last_timers_);
    if (!ok) break;  // :639-644
    // :648-661 exit tests
    if ((std::fabs(summary_.post_solve_norm - summary_.pre_solve_norm) / summary_.pre_solve_norm) <
        options_.error_change_threshold) {
      summary_.result = ErrorChangeBelowThreshold;
      break;
    }
    if (summary_.delta_norm < options_.param_change_threshold) {
      summary_.result = ParamChangeBelowThreshold;
      break;
    }
  }
  if (!Check(ba_hip_end_solve(engine_), "ba_hip_end_solve")) return;  // :672-678
  if (DoTvs && rig_->NumCams() > 0) {
    // :72-83 moved the rig's camera 0 on every applied step (host-side copy in the engine: no transfer)
    std::vector<double> tv(7 * (size_t)rig_->NumCams());
    if (Check(ba_hip_get_cameras(engine_, rig_->NumCams(), tv.data()), "ba_hip_get_cameras")) rig_->cameras_[0]->SetPose(SE3::from7(tv.data()));
  }
  if (kCamParamsInCalib && rig_->NumCams() > 0) {  // :46-53
    std::vector<double> cp(4 * (size_t)rig_->NumCams()), cw(rig_->NumCams());
    if (Check(ba_hip_get_camera_params(engine_, rig_->NumCams(), cp.data()), "ba_hip_get_camera_params") &&
        Check(ba_hip_get_camera_fov(engine_, rig_->NumCams(), cw.data()), "ba_hip_get_camera_fov"))
      rig_->cameras_[0]->SetParams(std::vector<double>({cp[0], cp[1], cp[2], cp[3], cw[0]}));
  }
  last_step_stale_ = true;
  uploaded_once_ = true;
  proj_view_dirty_ = true;  // GetProjectionResidual re-reads the device on its next call
  imu_view_dirty_ = true;
  if (!un_pose_.empty()) {
    // the reference scales each unary cov_inv in place every BuildProblem
    // (BundleAdjuster.cpp:1469), so the compounded weights survive across Solve() calls: on the
    // device as a cumulative scale per residual, on the host folded into un_cov_inv_ (used by the
    // next full upload, which restarts the device scales at 1)
    std::vector<double> sc(un_pose_.size(), 1.0);
    if (Check(ba_hip_get_unary_scales(engine_, sc.data()), "ba_hip_get_unary_scales"))
      for (size_t i = 0; i < sc.size(); ++i) {
        const double ratio = sc[i] / un_scale_seen_[i];
        for (int k = 0; k < 36; ++k) un_cov_inv_[36 * i + k] *= ratio;
        un_scale_seen_[i] = sc[i];
      }
  }
  // :680-704
  summary_.num_cond_inertial_residuals = (uint32_t)conditioning_inertial_residuals_.size();
  summary_.num_inertial_residuals = (uint32_t)imu_p1_.size();
  summary_.inertial_error = inertial_error_;
  summary_.num_cond_proj_residuals = (uint32_t)conditioning_proj_residuals_.size();
  summary_.num_proj_residuals = (uint32_t)pr_pose_.size();
  summary_.proj_error_ = proj_error_;
  // the conditioning sums (:680-704) are read back only when there is something to sum
  summary_.cond_inertial_error = 0;
  summary_.cond_proj_error = 0;
  if (!conditioning_inertial_residuals_.empty()) {
    std::vector<double> md(imu_p1_.size(), 0.0);
    if (Check(ba_hip_get_imu_errors(engine_, md.data()), "ba_hip_get_imu_errors"))
      for (uint32_t id : conditioning_inertial_residuals_) summary_.cond_inertial_error += md[id];
  }
  if (!conditioning_proj_residuals_.empty()) {
    // res.mahalanobis_distance / res.weight = |residual|^2 (:700-703), summed on the device
    double sq = 0;
    if (Check(ba_hip_get_conditioning_error(engine_, &sq), "ba_hip_get_conditioning_error")) summary_.cond_proj_error = sq;
  }
}

}  // namespace ba
