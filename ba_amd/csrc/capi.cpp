// Flat C wrapper (include/ba_capi.h) over the C++ host class ba::BundleAdjuster<>.
#include "../../include/ba_capi.h"

#include <cmath>
#include <vector>

#include "../../include/ba/BundleAdjuster.h"

namespace {

struct Iface {
  virtual ~Iface() {}
  virtual void init(const ba_options* o) = 0;
  virtual void set_gravity(const double* g) = 0;
  virtual uint32_t add_camera(const double* p, const double* t) = 0;
  virtual uint32_t add_camera_fov(const double* p5, const double* t) = 0;
  virtual double camera_fov(uint32_t cam) const = 0;
  virtual uint32_t add_pose(const double* t, const double* v, const double* b, int act, double time) = 0;
  virtual int set_pose_cam_params(uint32_t n, const double* params4) = 0;
  virtual void set_cov_once(int on) = 0;
  virtual void cond_errors(double* out2) const = 0;
  virtual void set_imu_noise(const double* r6, const double* rb6) = 0;
  virtual uint32_t add_landmark(const double* x, uint32_t rp, uint32_t rc, int act) = 0;
  virtual uint32_t add_proj(const double* z, uint32_t p, uint32_t l, uint32_t c, double w) = 0;
  virtual uint32_t add_unary(uint32_t p, const double* t, const double* cov, int rot) = 0;
  virtual uint32_t add_binary(uint32_t p1, uint32_t p2, const double* t, const double* cov, double w, int rot) = 0;
  virtual uint32_t add_imu(uint32_t p1, uint32_t p2, const double* m, uint32_t n, double w) = 0;
  virtual void regularize(uint32_t p, int t, int g, int b, int r) = 0;
  virtual void set_root(uint32_t id) = 0;
  virtual void solve(uint32_t it, double damping, int allow) = 0;
  virtual uint32_t num_poses() const = 0;
  virtual uint32_t num_landmarks() const = 0;
  virtual uint32_t num_proj() const = 0;
  virtual void get_poses(double* t, double* v, double* b) const = 0;
  virtual void get_landmarks(double* x) const = 0;
  virtual int reliable(uint32_t id) const = 0;
  virtual double outlier_ratio(uint32_t id) const = 0;
  virtual void proj_residual(uint32_t id, double* out11) const = 0;
  virtual uint32_t imu_residual(uint32_t id, double* out19) const = 0;
  virtual void summary(ba_summary* s) = 0;
  virtual void timers(ba_hip_timers* t) const = 0;
  virtual ba_hip_engine* engine() = 0;
  virtual void set_allreduce(ba_hip_allreduce_fn fn, void* ctx, int rank, int nranks) = 0;
  virtual void set_communicator(const void* id128, int rank, int nranks, int distributed) = 0;
  virtual void set_collectives(ba_hip_collective_fn fn, void* ctx) = 0;
  virtual int solve_is_distributed() = 0;
  virtual void camera_pose(uint32_t cam, double* t7) const = 0;
  virtual void camera_params(uint32_t cam, double* p4) const = 0;
  virtual void last_calib_step(double* d6) const = 0;
  virtual uint32_t marginals(double* cov) const = 0;
};

template <int LM, int PD, bool TVS = false, int CS = 0>
struct Impl : Iface {
  typedef ba::BundleAdjuster<double, LM, PD, CS, TVS> BA;
  BA ba;
  void init(const ba_options* o) override {
    ba::Options<double> opt;
    opt.trust_region_size = o->trust_region_size;
    opt.gyro_sigma = o->gyro_sigma; opt.accel_sigma = o->accel_sigma;
    opt.gyro_bias_sigma = o->gyro_bias_sigma; opt.accel_bias_sigma = o->accel_bias_sigma;
    opt.projection_outlier_threshold = o->projection_outlier_threshold;
    opt.error_change_threshold = o->error_change_threshold;
    opt.param_change_threshold = o->param_change_threshold;
    opt.dogleg_max_inner_iterations = o->dogleg_max_inner_iterations;
    opt.apply_results = o->apply_results != 0; opt.use_dogleg = o->use_dogleg != 0;
    opt.use_triangular_matrices = o->use_triangular_matrices != 0;
    opt.use_sparse_solver = o->use_sparse_solver != 0;
    opt.regularize_biases_in_batch = o->regularize_biases_in_batch != 0;
    opt.enable_auto_regularization = o->enable_auto_regularization != 0;
    opt.use_robust_norm_for_proj_residuals = o->use_robust_norm_for_proj_residuals != 0;
    opt.use_robust_norm_for_inertial_residuals = o->use_robust_norm_for_inertial_residuals != 0;
    opt.write_reduced_camera_matrix = o->write_reduced_camera_matrix >= 2;
    opt.keep_reduced_system = o->write_reduced_camera_matrix != 0;
    opt.device = o->device;
    opt.factorization_pivot_tolerance = o->factorization_pivot_tolerance;
    opt.calculate_calibration_marginals = o->calculate_calibration_marginals != 0;
    ba.Init(opt);
  }
  void set_gravity(const double* g) override { ba.SetGravity(ba::Vector3t({g[0], g[1], g[2]})); }
  uint32_t add_camera(const double* p, const double* t) override {
    return ba.AddCamera(std::make_shared<ba::CameraInterface<double>>(p[0], p[1], p[2], p[3], ba::SE3::from7(t)));
  }
  uint32_t add_camera_fov(const double* p, const double* t) override {
    return ba.AddCamera(std::make_shared<ba::FovCamera<double>>(p[0], p[1], p[2], p[3], p[4], ba::SE3::from7(t)));
  }
  double camera_fov(uint32_t cam) const override { return cam < ba.rig()->NumCams() ? ba.rig()->cameras_[cam]->Param(4) : std::nan(""); }
  void set_imu_noise(const double* r6, const double* rb6) override {
    auto calib = ba.GetImuCalibration();
    for (int i = 0; i < 6; ++i) { calib.r[i] = r6[i]; calib.r_b[i] = rb6[i]; }
    ba.SetImuCalibration(calib);
  }
  void cond_errors(double* o) const override {
    const auto& m = ba.GetSolutionSummary();
    o[0] = m.cond_proj_error; o[1] = m.cond_inertial_error;
  }
  void set_cov_once(int on) override { ba.options().calculate_inertial_covariance_once = on != 0; }
  int set_pose_cam_params(uint32_t n, const double* params4) override {
    if (n == 0 || !params4) { ba.options().use_per_pose_cam_params = false; return 0; }
    if (n != ba.GetNumPoses()) return 1;
    for (uint32_t i = 0; i < n; ++i)
      ba.SetPoseCamParams(i, std::vector<double>(params4 + 4 * (size_t)i, params4 + 4 * (size_t)i + 4));
    ba.options().use_per_pose_cam_params = true;
    return 0;
  }
  uint32_t add_pose(const double* t, const double* v, const double* b, int act, double time) override {
    ba::Vector3t vv; ba::Vector6t bb;
    if (v) for (int i = 0; i < 3; ++i) vv[i] = v[i];
    if (b) for (int i = 0; i < 6; ++i) bb[i] = b[i];
    return ba.AddPose(ba::SE3::from7(t), std::vector<double>(), vv, bb, act != 0, time);
  }
  uint32_t add_landmark(const double* x, uint32_t rp, uint32_t rc, int act) override {
    return ba.AddLandmark(ba::Vector4t({x[0], x[1], x[2], x[3]}), rp, rc, act != 0);
  }
  uint32_t add_proj(const double* z, uint32_t p, uint32_t l, uint32_t c, double w) override {
    return ba.AddProjectionResidual(ba::Vector2t({z[0], z[1]}), p, l, c, w);
  }
  static ba::Matrix6t cov_of(const double* cov) {
    ba::Matrix6t m = ba::Matrix6t::Identity();
    if (cov) for (int i = 0; i < 36; ++i) m.m[i] = cov[i];
    return m;
  }
  uint32_t add_unary(uint32_t p, const double* t, const double* cov, int rot) override {
    return ba.AddUnaryConstraint(p, ba::SE3::from7(t), cov_of(cov), rot != 0);
  }
  uint32_t add_binary(uint32_t p1, uint32_t p2, const double* t, const double* cov, double w, int rot) override {
    return ba.AddBinaryConstraint(p1, p2, ba::SE3::from7(t), cov_of(cov), w, rot != 0);
  }
  uint32_t add_imu(uint32_t p1, uint32_t p2, const double* m, uint32_t n, double w) override {
    std::vector<ba::ImuMeasurementT<double>> meas(n);
    for (uint32_t i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) { meas[i].w[k] = m[7 * i + k]; meas[i].a[k] = m[7 * i + 3 + k]; }
      meas[i].time = m[7 * i + 6];
    }
    return ba.AddImuResidual(p1, p2, meas, w);
  }
  void regularize(uint32_t p, int t, int g, int b, int r) override { ba.RegularizePose(p, t != 0, g != 0, b != 0, r != 0); }
  void set_root(uint32_t id) override { ba.SetRootPoseId(id); }
  void solve(uint32_t it, double damping, int allow) override { ba.Solve(it, damping, allow != 0); }
  uint32_t num_poses() const override { return ba.GetNumPoses(); }
  uint32_t num_landmarks() const override { return ba.GetNumLandmarks(); }
  uint32_t num_proj() const override { return ba.GetNumProjResiduals(); }
  void get_poses(double* t, double* v, double* b) const override {
    for (uint32_t p = 0; p < ba.GetNumPoses(); ++p) {
      const auto& pose = ba.GetPose(p);
      if (t) pose.t_wp.to7(t + 7 * (size_t)p);
      if (v) for (int i = 0; i < 3; ++i) v[3 * (size_t)p + i] = pose.v_w[i];
      if (b) for (int i = 0; i < 6; ++i) b[6 * (size_t)p + i] = pose.b[i];
    }
  }
  void get_landmarks(double* x) const override {
    for (uint32_t l = 0; l < ba.GetNumLandmarks(); ++l) {
      const ba::Vector4t& v = ba.GetLandmark(l);
      for (int i = 0; i < 4; ++i) x[4 * (size_t)l + i] = v[i];
    }
  }
  int reliable(uint32_t id) const override { return ba.IsLandmarkReliable(id) ? 1 : 0; }
  double outlier_ratio(uint32_t id) const override { return ba.LandmarkOutlierRatio(id); }
  uint32_t imu_residual(uint32_t id, double* o) const override {
    if (id >= ba.GetNumImuResiduals()) return 0;
    const auto& r = ba.GetImuResidual(id);
    o[0] = r.pose1_id; o[1] = r.pose2_id; o[2] = r.weight; o[3] = (double)r.measurements.size();
    for (int i = 0; i < 15; ++i) o[4 + i] = r.residual[i];
    return (uint32_t)r.measurements.size();
  }
  void proj_residual(uint32_t id, double* o) const override {
    const typename BA::ProjectionResidual& r = ba.GetProjectionResidual(id);
    o[0] = r.z[0]; o[1] = r.z[1]; o[2] = r.residual[0]; o[3] = r.residual[1];
    o[4] = r.weight; o[5] = r.orig_weight; o[6] = r.mahalanobis_distance;
    o[7] = r.x_meas_id; o[8] = r.x_ref_id; o[9] = r.landmark_id; o[10] = r.cam_id;
  }
  void summary(ba_summary* s) override {
    const auto& m = ba.GetSolutionSummary();
    s->num_proj_residuals = m.num_proj_residuals; s->num_inertial_residuals = m.num_inertial_residuals;
    s->num_cond_proj_residuals = m.num_cond_proj_residuals;
    s->num_cond_inertial_residuals = m.num_cond_inertial_residuals;
    double pe, ue, be, ie;
    ba.GetErrors(pe, ue, be, ie);
    s->proj_error = pe; s->unary_error = ue; s->binary_error = be; s->inertial_error = ie;
    s->delta_norm = m.delta_norm; s->pre_solve_norm = m.pre_solve_norm; s->post_solve_norm = m.post_solve_norm;
    s->result = (int32_t)m.result; s->iterations_run = ba.iterations_run();
    s->trust_region_size = ba.trust_region_size();
  }
  void timers(ba_hip_timers* t) const override { *t = ba.GetLastTimers(); }
  ba_hip_engine* engine() override { return ba.engine(); }
  void set_communicator(const void* id128, int rank, int nranks, int distributed) override {
    if (id128) ba.SetCommunicator(id128, rank, nranks, distributed != 0);
    else ba.ClearCommunicator();
  }
  int solve_is_distributed() override { return ba.SolveIsDistributed() ? 1 : 0; }
  void set_collectives(ba_hip_collective_fn fn, void* ctx) override { ba.SetCollectives(fn, ctx); }
  void set_allreduce(ba_hip_allreduce_fn fn, void* ctx, int rank, int nranks) override {
    ba.SetAllReduce(fn, ctx, rank, nranks);
  }
  // (an out-of-range camera id writes NaN, not whatever lies behind the rig's vector)
  void camera_pose(uint32_t cam, double* t7) const override {
    if (cam >= ba.rig()->NumCams()) { for (int i = 0; i < 7; ++i) t7[i] = std::nan(""); return; }
    ba.rig()->cameras_[cam]->Pose().to7(t7);
  }
  void camera_params(uint32_t cam, double* p4) const override {
    if (cam >= ba.rig()->NumCams()) { for (int i = 0; i < 4; ++i) p4[i] = std::nan(""); return; }
    const ba::Vector4t p = ba.rig()->cameras_[cam]->GetParams();
    for (int i = 0; i < 4; ++i) p4[i] = p[i];
  }
  uint32_t marginals(double* cov) const override {
    const ba::MatX& m = ba.GetSolutionSummary().calibration_marginals;
    for (int i = 0; i < m.rows() * m.cols(); ++i) cov[i] = m.data()[i];
    return (uint32_t)m.rows();
  }
  void last_calib_step(double* d6) const override {
    const auto& d = ba.GetLastStep().delta_k;
    for (size_t i = 0; i < 6; ++i) d6[i] = i < d.size() ? d[i] : 0.0;
  }
};

Iface* make(int lm, int pd, int do_tvs, int calib_size) {
  // CalibSize 4 = the pinhole parameters of a LinearCamera 0 (LmSize 1, without DoTvs)
  if (calib_size == 4 && !do_tvs && lm == 1) {
    if (pd == 6) return new Impl<1, 6, false, 4>();
    if (pd == 9) return new Impl<1, 9, false, 4>();
    if (pd == 15) return new Impl<1, 15, false, 4>();
  }
  // CalibSize 5 = the parameters of a FovCamera 0 (reference BundleAdjuster.cpp:1816, 1822: <1,6,5>, <1,15,5>)
  if (calib_size == 5 && !do_tvs && lm == 1) {
    if (pd == 6) return new Impl<1, 6, false, 5>();
    if (pd == 15) return new Impl<1, 15, false, 5>();
  }
  if (calib_size != 0) return nullptr;
#define CASE(L, P) if (lm == L && pd == P && !do_tvs) return new Impl<L, P>()
  CASE(0, 6); CASE(0, 9); CASE(0, 15); CASE(1, 6); CASE(1, 9); CASE(1, 15);
  CASE(3, 6); CASE(3, 9); CASE(3, 15);
#undef CASE
  // DoTvs instantiations (reference BundleAdjuster.cpp:1816-1822 instantiates <1,6,5,true> and
  // <1,15,5,true>; here with CalibSize 0)
  if (do_tvs && lm == 1 && pd == 6) return new Impl<1, 6, true>();
  if (do_tvs && lm == 1 && pd == 9) return new Impl<1, 9, true>();
  if (do_tvs && lm == 1 && pd == 15) return new Impl<1, 15, true>();
  return nullptr;
}

}  // namespace

struct ba_adjuster { Iface* p; };

extern "C" {

void ba_default_options(ba_options* o) {
  const ba::Options<double> d;
  o->trust_region_size = d.trust_region_size;
  o->gyro_sigma = d.gyro_sigma; o->accel_sigma = d.accel_sigma;
  o->gyro_bias_sigma = d.gyro_bias_sigma; o->accel_bias_sigma = d.accel_bias_sigma;
  o->projection_outlier_threshold = d.projection_outlier_threshold;
  o->error_change_threshold = d.error_change_threshold; o->param_change_threshold = d.param_change_threshold;
  o->dogleg_max_inner_iterations = d.dogleg_max_inner_iterations;
  o->apply_results = d.apply_results; o->use_dogleg = d.use_dogleg;
  o->use_triangular_matrices = d.use_triangular_matrices; o->use_sparse_solver = d.use_sparse_solver;
  o->regularize_biases_in_batch = d.regularize_biases_in_batch;
  o->enable_auto_regularization = d.enable_auto_regularization;
  o->use_robust_norm_for_proj_residuals = d.use_robust_norm_for_proj_residuals;
  o->use_robust_norm_for_inertial_residuals = d.use_robust_norm_for_inertial_residuals;
  o->write_reduced_camera_matrix = d.write_reduced_camera_matrix ? 2 : (d.keep_reduced_system ? 1 : 0);
  o->device = d.device;
  o->factorization_pivot_tolerance = d.factorization_pivot_tolerance;
  o->calculate_calibration_marginals = d.calculate_calibration_marginals;
  o->reserved = 0;
}
ba_adjuster* ba_adjuster_create(int lm_dim, int pose_dim) { return ba_adjuster_create_calib(lm_dim, pose_dim, 0, 0); }
ba_adjuster* ba_adjuster_create_calib(int lm_dim, int pose_dim, int calib_size, int do_tvs) {
  Iface* p = make(lm_dim, pose_dim, do_tvs, calib_size);
  if (!p) return nullptr;
  ba_adjuster* a = new ba_adjuster();
  a->p = p;
  return a;
}
void ba_adjuster_destroy(ba_adjuster* a) { if (a) { delete a->p; delete a; } }
void ba_adjuster_init(ba_adjuster* a, const ba_options* o) { a->p->init(o); }
void ba_adjuster_set_gravity(ba_adjuster* a, const double g[3]) { a->p->set_gravity(g); }
uint32_t ba_adjuster_add_camera(ba_adjuster* a, const double params[4], const double t_vs[7]) { return a->p->add_camera(params, t_vs); }
uint32_t ba_adjuster_add_camera_fov(ba_adjuster* a, const double params[5], const double t_vs[7]) { return a->p->add_camera_fov(params, t_vs); }
double ba_adjuster_get_camera_fov(const ba_adjuster* a, uint32_t cam_id) { return a->p->camera_fov(cam_id); }
uint32_t ba_adjuster_add_pose(ba_adjuster* a, const double t_wp[7], const double v_w[3], const double b[6], int is_active, double time) {
  return a->p->add_pose(t_wp, v_w, b, is_active, time);
}
uint32_t ba_adjuster_add_landmark(ba_adjuster* a, const double x_w[4], uint32_t ref_pose_id, uint32_t ref_cam_id, int is_active) {
  return a->p->add_landmark(x_w, ref_pose_id, ref_cam_id, is_active);
}
uint32_t ba_adjuster_add_projection_residual(ba_adjuster* a, const double z[2], uint32_t meas_pose_id, uint32_t landmark_id, uint32_t cam_id, double weight) {
  return a->p->add_proj(z, meas_pose_id, landmark_id, cam_id, weight);
}
uint32_t ba_adjuster_add_unary_constraint(ba_adjuster* a, uint32_t pose_id, const double t_wv[7], const double cov[36], int use_rotation) {
  return a->p->add_unary(pose_id, t_wv, cov, use_rotation);
}
uint32_t ba_adjuster_add_binary_constraint(ba_adjuster* a, uint32_t p1, uint32_t p2, const double t_12[7], const double cov[36], double weight, int use_rotation) {
  return a->p->add_binary(p1, p2, t_12, cov, weight, use_rotation);
}
uint32_t ba_adjuster_add_imu_residual(ba_adjuster* a, uint32_t p1, uint32_t p2, const double* meas7, uint32_t n, double weight) {
  return a->p->add_imu(p1, p2, meas7, n, weight);
}
void ba_adjuster_regularize_pose(ba_adjuster* a, uint32_t pose_id, int translation, int gravity, int bias, int rotation) {
  a->p->regularize(pose_id, translation, gravity, bias, rotation);
}
void ba_adjuster_set_root_pose_id(ba_adjuster* a, uint32_t id) { a->p->set_root(id); }
int ba_adjuster_set_pose_cam_params(ba_adjuster* a, uint32_t n, const double* params4) { return a->p->set_pose_cam_params(n, params4); }
void ba_adjuster_set_calculate_inertial_covariance_once(ba_adjuster* a, int on) { a->p->set_cov_once(on); }
void ba_adjuster_set_imu_noise(ba_adjuster* a, const double r6[6], const double rb6[6]) { a->p->set_imu_noise(r6, rb6); }
void ba_adjuster_get_cond_errors(const ba_adjuster* a, double out2[2]) { a->p->cond_errors(out2); }
void ba_adjuster_add_poses(ba_adjuster* a, uint32_t n, const double* t_wp, const double* v_w, const double* b, const uint8_t* is_active, const double* time) {
  for (uint32_t i = 0; i < n; ++i)
    a->p->add_pose(t_wp + 7 * (size_t)i, v_w ? v_w + 3 * (size_t)i : nullptr, b ? b + 6 * (size_t)i : nullptr,
                   is_active ? is_active[i] : 1, time ? time[i] : -1);
}
void ba_adjuster_add_landmarks(ba_adjuster* a, uint32_t n, const double* x_w, const uint32_t* ref_pose_id, const uint32_t* ref_cam_id, const uint8_t* is_active) {
  for (uint32_t i = 0; i < n; ++i)
    a->p->add_landmark(x_w + 4 * (size_t)i, ref_pose_id[i], ref_cam_id ? ref_cam_id[i] : 0, is_active ? is_active[i] : 1);
}
void ba_adjuster_add_projection_residuals(ba_adjuster* a, uint32_t n, const double* z, const uint32_t* meas_pose_id, const uint32_t* landmark_id, const uint32_t* cam_id, const double* weight, uint32_t* out_ids) {
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t id = a->p->add_proj(z + 2 * (size_t)i, meas_pose_id[i], landmark_id[i], cam_id ? cam_id[i] : 0, weight ? weight[i] : 1.0);
    if (out_ids) out_ids[i] = id;
  }
}
void ba_adjuster_solve(ba_adjuster* a, uint32_t max_iter, double gn_damping, int error_increase_allowed) { a->p->solve(max_iter, gn_damping, error_increase_allowed); }
uint32_t ba_adjuster_num_poses(const ba_adjuster* a) { return a->p->num_poses(); }
uint32_t ba_adjuster_num_landmarks(const ba_adjuster* a) { return a->p->num_landmarks(); }
uint32_t ba_adjuster_num_proj_residuals(const ba_adjuster* a) { return a->p->num_proj(); }
void ba_adjuster_get_poses(const ba_adjuster* a, double* t_wp, double* v_w, double* b) { a->p->get_poses(t_wp, v_w, b); }
void ba_adjuster_get_landmarks(const ba_adjuster* a, double* x_w) { a->p->get_landmarks(x_w); }
int ba_adjuster_is_landmark_reliable(const ba_adjuster* a, uint32_t id) { return a->p->reliable(id); }
double ba_adjuster_landmark_outlier_ratio(const ba_adjuster* a, uint32_t id) { return a->p->outlier_ratio(id); }
void ba_adjuster_get_projection_residual(const ba_adjuster* a, uint32_t id, double* out11) { a->p->proj_residual(id, out11); }
uint32_t ba_adjuster_get_imu_residual(const ba_adjuster* a, uint32_t id, double* out19) { return a->p->imu_residual(id, out19); }
void ba_adjuster_get_summary(const ba_adjuster* a, ba_summary* s) { a->p->summary(s); }
void ba_adjuster_get_timers(const ba_adjuster* a, ba_hip_timers* t) { a->p->timers(t); }
ba_hip_engine* ba_adjuster_engine(ba_adjuster* a) { return a->p->engine(); }
void ba_adjuster_get_camera_pose(const ba_adjuster* a, uint32_t cam_id, double t_vs[7]) { a->p->camera_pose(cam_id, t_vs); }
void ba_adjuster_get_camera_params(const ba_adjuster* a, uint32_t cam_id, double params[4]) { a->p->camera_params(cam_id, params); }
void ba_adjuster_get_last_calib_step(const ba_adjuster* a, double delta_k[6]) { a->p->last_calib_step(delta_k); }
uint32_t ba_adjuster_get_calibration_marginals(const ba_adjuster* a, double cov[36]) { return a->p->marginals(cov); }
void ba_adjuster_set_allreduce(ba_adjuster* a, ba_hip_allreduce_fn fn, void* ctx, int rank, int nranks) { a->p->set_allreduce(fn, ctx, rank, nranks); }
void ba_adjuster_set_communicator(ba_adjuster* a, const void* id128, int rank, int nranks, int distributed_solve) { a->p->set_communicator(id128, rank, nranks, distributed_solve); }
int ba_adjuster_solve_is_distributed(ba_adjuster* a) { return a->p->solve_is_distributed(); }
void ba_adjuster_set_collectives(ba_adjuster* a, ba_hip_collective_fn fn, void* ctx) { a->p->set_collectives(fn, ctx); }

}  // extern "C"
