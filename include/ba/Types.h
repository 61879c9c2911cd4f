// Value types of the MI355X-native ba::BundleAdjuster host layer.
//
// The reference's public signatures mention Eigen, Sophus and Calibu types
// (/root/reference/include/ba/BundleAdjuster.h:136-155, Types.h:41-321).  None of those
// libraries is a dependency here: the host layer owns small plain value types and
// accepts the third-party ones through converting constructors (anything exposing
// .translation() / .unit_quaternion(), operator[] / operator(), .data()), so existing
// application code that passes Sophus::SE3d / Eigen vectors keeps compiling when those
// headers are present.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <type_traits>
#include <vector>

namespace ba {

static const double Gravity = 9.8007;  // reference Types.h:39
#ifndef IMU_GYRO_SIGMA                 // reference Types.h:33-36
#define IMU_GYRO_SIGMA 5.3088444e-5
#define IMU_GYRO_BIAS_SIGMA 1.4125375e-4
#define IMU_ACCEL_SIGMA 0.001883649
#define IMU_ACCEL_BIAS_SIGMA 1.2589254e-2
#endif

// Fixed-size column vector / row-major matrix of doubles.
template <int R, int C = 1>
struct Mat {
  double m[R * C];
  Mat() { for (int i = 0; i < R * C; ++i) m[i] = 0.0; }
  // from any indexable container of at least R*C scalars (vectors) ...
  template <typename V, typename = decltype(std::declval<const V&>()[0]),
            typename = typename std::enable_if<!std::is_arithmetic<V>::value>::type>
  Mat(const V& v) { for (int i = 0; i < R * C; ++i) m[i] = (double)v[i]; }
  Mat(std::initializer_list<double> l) {
    int i = 0;
    for (double x : l) { if (i < R * C) m[i++] = x; }
    for (; i < R * C; ++i) m[i] = 0.0;
  }
  double& operator[](int i) { return m[i]; }
  double operator[](int i) const { return m[i]; }
  double& operator()(int r, int c = 0) { return m[r * C + c]; }
  double operator()(int r, int c = 0) const { return m[r * C + c]; }
  double* data() { return m; }
  const double* data() const { return m; }
  static Mat Zero() { return Mat(); }
  static Mat Identity() { Mat a; for (int i = 0; i < (R < C ? R : C); ++i) a.m[i * C + i] = 1.0; return a; }
  double norm() const { double s = 0; for (int i = 0; i < R * C; ++i) s += m[i] * m[i]; return std::sqrt(s); }
};
// Run-time sized row-major matrix (SolutionSummary::calibration_marginals is a MatrixXt in the reference)
struct MatX {
  int nr = 0, nc = 0;
  std::vector<double> m;
  MatX() {}
  MatX(int r, int c) : nr(r), nc(c), m((size_t)r * c, 0.0) {}
  int rows() const { return nr; }
  int cols() const { return nc; }
  double& operator()(int r, int c) { return m[(size_t)r * nc + c]; }
  double operator()(int r, int c) const { return m[(size_t)r * nc + c]; }
  double* data() { return m.data(); }
  const double* data() const { return m.data(); }
};
typedef Mat<2> Vector2t;
typedef Mat<3> Vector3t;
typedef Mat<4> Vector4t;
typedef Mat<6> Vector6t;
typedef Mat<7> Vector7t;
typedef Mat<9> Vector9t;
typedef Mat<3, 3> Matrix3t;
typedef Mat<6, 6> Matrix6t;

// Rigid transform: translation + unit quaternion (x,y,z,w), Sophus storage order.
struct SE3 {
  double t[3];
  double q[4];
  SE3() { t[0] = t[1] = t[2] = 0; q[0] = q[1] = q[2] = 0; q[3] = 1; }
  SE3(const double* t3, const double* q4) { for (int i = 0; i < 3; ++i) t[i] = t3[i]; for (int i = 0; i < 4; ++i) q[i] = q4[i]; }
  static SE3 from7(const double* p) { return SE3(p, p + 3); }
  // from Sophus::SE3Group<double> or anything alike
  template <typename S, typename = decltype(std::declval<const S&>().unit_quaternion()),
            typename = decltype(std::declval<const S&>().translation())>
  SE3(const S& s) {
    const auto tr = s.translation();
    const auto qq = s.unit_quaternion();
    for (int i = 0; i < 3; ++i) t[i] = tr[i];
    q[0] = qq.x(); q[1] = qq.y(); q[2] = qq.z(); q[3] = qq.w();
  }
  void to7(double* p) const { for (int i = 0; i < 3; ++i) p[i] = t[i]; for (int i = 0; i < 4; ++i) p[3 + i] = q[i]; }
  // group operations as Sophus::SE3Group offers them to callers (applications/: `pose * update`,
  // `pose *= update`, `inverse()`); the quaternion product is renormalised as Sophus does
  SE3 operator*(const SE3& o) const {
    SE3 r;
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    // t + R o.t  (Eigen's _transformVector form)
    const double ux = y * o.t[2] - z * o.t[1], uy = z * o.t[0] - x * o.t[2], uz = x * o.t[1] - y * o.t[0];
    const double vx = 2 * ux, vy = 2 * uy, vz = 2 * uz;
    r.t[0] = t[0] + o.t[0] + w * vx + (y * vz - z * vy);
    r.t[1] = t[1] + o.t[1] + w * vy + (z * vx - x * vz);
    r.t[2] = t[2] + o.t[2] + w * vz + (x * vy - y * vx);
    r.q[0] = w * o.q[0] + x * o.q[3] + y * o.q[2] - z * o.q[1];
    r.q[1] = w * o.q[1] + y * o.q[3] + z * o.q[0] - x * o.q[2];
    r.q[2] = w * o.q[2] + z * o.q[3] + x * o.q[1] - y * o.q[0];
    r.q[3] = w * o.q[3] - x * o.q[0] - y * o.q[1] - z * o.q[2];
    const double n = std::sqrt(r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3]);
    for (int i = 0; i < 4; ++i) r.q[i] /= n;
    return r;
  }
  SE3& operator*=(const SE3& o) { *this = *this * o; return *this; }
  SE3 inverse() const {
    SE3 r;
    r.q[0] = -q[0]; r.q[1] = -q[1]; r.q[2] = -q[2]; r.q[3] = q[3];
    SE3 rot = r;  // pure rotation R^T
    SE3 tt;
    tt.t[0] = -t[0]; tt.t[1] = -t[1]; tt.t[2] = -t[2];
    r = rot * tt;
    r.q[0] = -q[0]; r.q[1] = -q[1]; r.q[2] = -q[2]; r.q[3] = q[3];
    return r;
  }
  Vector3t translation() const { Vector3t v; for (int i = 0; i < 3; ++i) v[i] = t[i]; return v; }
  Matrix3t rotationMatrix() const {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    Matrix3t R;
    R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - w * z); R(0, 2) = 2 * (x * z + w * y);
    R(1, 0) = 2 * (x * y + w * z); R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - w * x);
    R(2, 0) = 2 * (x * z - w * y); R(2, 1) = 2 * (y * z + w * x); R(2, 2) = 1 - 2 * (x * x + y * y);
    return R;
  }
};

// What the hot path needs of calibu::CameraInterface: the camera parameters and the
// vehicle-to-sensor pose (reference call sites parallel_algos.h:44-62).  Two models: the pinhole
// (calibu::LinearCamera: fx, fy, u0, v0) and the FOV camera (calibu::FovCamera: fx, fy, u0, v0, w — the
// five parameters of SelfCalBundleAdjuster = BundleAdjuster<Scalar, 1, 6, 5>, reference BundleAdjuster.h:758-759).
// Calibu is not in the reference tree; the FOV model is the published one (Devernay & Faugeras 2001,
// r_d = atan(2 r_u tan(w/2)) / w, applied as a factor on the normalised point), as in ba_amd/csrc/dmath.h.
template <typename Scalar = double>
class CameraInterface {
 public:
  enum Model { kLinear = 0, kFov = 1 };
  CameraInterface() { p_[0] = p_[1] = 1; p_[2] = p_[3] = 0; p_[4] = 0; }
  CameraInterface(double fx, double fy, double u0, double v0, const SE3& t_vs = SE3()) : t_vs_(t_vs) {
    p_[0] = fx; p_[1] = fy; p_[2] = u0; p_[3] = v0; p_[4] = 0;
  }
  int Type() const { return model_; }
  uint32_t NumParams() const { return model_ == kFov ? 5u : 4u; }
  // the first four parameters (all of them for the pinhole); Param(i) / ParamsVector() reach every one
  Vector4t GetParams() const { Vector4t v; for (int i = 0; i < 4; ++i) v[i] = p_[i]; return v; }
  double Param(uint32_t i) const { return p_[i]; }
  std::vector<double> ParamsVector() const { return std::vector<double>(p_, p_ + NumParams()); }
  // takes min(NumParams(), what the vector holds) entries — a Vector4t leaves w of a FOV camera alone
  template <typename V> void SetParams(const V& v) { for (int i = 0; i < 4; ++i) p_[i] = v[i]; }
  void SetParams(const std::vector<double>& v) { for (size_t i = 0; i < v.size() && i < NumParams(); ++i) p_[i] = v[i]; }
  const SE3& Pose() const { return t_vs_; }
  void SetPose(const SE3& t) { t_vs_ = t; }
  // the camera model itself, for callers that synthesise or check measurements on the host
  // (semantics of the calibu calls at parallel_algos.h:59-62,73-74)
  Vector2t Project(const Vector3t& P) const {
    if (model_ == kFov) {
      const double px = P[0] / P[2], py = P[1] / P[2], f = Factor(std::sqrt(px * px + py * py));
      return Vector2t({p_[0] * (f * px) + p_[2], p_[1] * (f * py) + p_[3]});
    }
    return Vector2t({p_[0] * P[0] / P[2] + p_[2], p_[1] * P[1] / P[2] + p_[3]});
  }
  Vector3t Unproject(const Vector2t& pix) const {  // the ray with z = 1
    double x = (pix[0] - p_[2]) / p_[0], y = (pix[1] - p_[3]) / p_[1];
    if (model_ == kFov) { const double g = FactorInv(std::sqrt(x * x + y * y)); x *= g; y *= g; }
    return Vector3t({x, y, 1.0});
  }
  // Transfer3d(T_ba, ray, rho) = Project(R ray + rho t)
  Vector2t Transfer3d(const SE3& t_ba, const Vector3t& ray, const Scalar rho) const {
    const Matrix3t R = t_ba.rotationMatrix();
    Vector3t P;
    for (int r = 0; r < 3; ++r) P[r] = R(r, 0) * ray[0] + R(r, 1) * ray[1] + R(r, 2) * ray[2] + rho * t_ba.t[r];
    return Project(P);
  }
  // dTransfer3d_dray(T_ba, ray, rho) = [dProject R, dProject t]  (2 x 4)
  Mat<2, 4> dTransfer3d_dray(const SE3& t_ba, const Vector3t& ray, const Scalar rho) const {
    const Matrix3t R = t_ba.rotationMatrix();
    Vector3t P;
    for (int r = 0; r < 3; ++r) P[r] = R(r, 0) * ray[0] + R(r, 1) * ray[1] + R(r, 2) * ray[2] + rho * t_ba.t[r];
    double d[2][3] = {{p_[0] / P[2], 0.0, -p_[0] * P[0] / (P[2] * P[2])},
                      {0.0, p_[1] / P[2], -p_[1] * P[1] / (P[2] * P[2])}};
    if (model_ == kFov) {
      const double iz = 1.0 / P[2], px = P[0] * iz, py = P[1] * iz, r = std::sqrt(px * px + py * py);
      double df_dr;
      const double f = Factor(r, &df_dr), k = r > 0.0 ? df_dr / r : 0.0;
      const double a00 = f + k * px * px, a01 = k * px * py, a11 = f + k * py * py;
      d[0][0] = p_[0] * a00 * iz; d[0][1] = p_[0] * a01 * iz; d[0][2] = -p_[0] * (a00 * px + a01 * py) * iz;
      d[1][0] = p_[1] * a01 * iz; d[1][1] = p_[1] * a11 * iz; d[1][2] = -p_[1] * (a01 * px + a11 * py) * iz;
    }
    Mat<2, 4> J;
    for (int r = 0; r < 2; ++r) {
      for (int c = 0; c < 3; ++c) J(r, c) = d[r][0] * R(0, c) + d[r][1] * R(1, c) + d[r][2] * R(2, c);
      J(r, 3) = d[r][0] * t_ba.t[0] + d[r][1] * t_ba.t[1] + d[r][2] * t_ba.t[2];
    }
    return J;
  }
 protected:
  // r_d / r_u and r_u / r_d of the FOV model, with their limits for a vanishing radius / vanishing w
  double Factor(double r, double* df_dr = nullptr) const {
    const double w = p_[4];
    if (df_dr) *df_dr = 0.0;
    if (w * w <= 1e-5) return 1.0;
    const double m = 2.0 * std::tan(0.5 * w);
    if (r * r < 1e-5) return m / w;
    const double at = std::atan(r * m);
    if (df_dr) *df_dr = m / ((1.0 + r * r * m * m) * r * w) - at / (r * r * w);
    return at / (r * w);
  }
  double FactorInv(double rd) const {
    const double w = p_[4];
    if (w * w <= 1e-5) return 1.0;
    const double m = 2.0 * std::tan(0.5 * w);
    if (rd * rd < 1e-5) return w / m;
    return std::tan(rd * w) / (rd * m);
  }
  double p_[5];
  int model_ = kLinear;
  SE3 t_vs_;
};
template <typename Scalar = double>
using LinearCamera = CameraInterface<Scalar>;
// calibu::FovCamera stand-in: parameters fx, fy, u0, v0, w
template <typename Scalar = double>
class FovCamera : public CameraInterface<Scalar> {
 public:
  FovCamera(double fx, double fy, double u0, double v0, double w, const SE3& t_vs = SE3())
      : CameraInterface<Scalar>(fx, fy, u0, v0, t_vs) {
    this->p_[4] = w;
    this->model_ = CameraInterface<Scalar>::kFov;
  }
};

// calibu::Rig stand-in
template <typename Scalar = double>
struct Rig {
  std::vector<std::shared_ptr<CameraInterface<Scalar>>> cameras_;
  void AddCamera(std::shared_ptr<CameraInterface<Scalar>> c) { cameras_.push_back(c); }
  uint32_t NumCams() const { return (uint32_t)cameras_.size(); }
};

// reference Types.h:222-244
template <typename Scalar = double>
struct ImuMeasurementT {
  Vector3t w, a;
  double time;
  ImuMeasurementT() : time(0) {}
  template <typename V>
  ImuMeasurementT(const V& w_, const V& a_, double t) : w(w_), a(a_), time(t) {}
  ImuMeasurementT operator*(const Scalar& rhs) const {
    ImuMeasurementT r = *this;
    for (int i = 0; i < 3; ++i) { r.w[i] *= rhs; r.a[i] *= rhs; }
    return r;
  }
  ImuMeasurementT operator+(const ImuMeasurementT& rhs) const {
    ImuMeasurementT r = *this;
    for (int i = 0; i < 3; ++i) { r.w[i] += rhs.w[i]; r.a[i] += rhs.a[i]; }
    return r;
  }
};

// The fields of the reference's PoseT / LandmarkT that callers read back
// (Types.h:41-89); adjacency lists and caches live in the engine instead.
template <typename Scalar = double>
struct PoseT {
  SE3 t_wp;
  Vector3t v_w;
  Vector6t b;
  std::vector<bool> param_mask;
  bool is_param_mask_used = false;
  bool is_active = true;
  int external_id = -1;
  uint32_t id = 0, opt_id = 0;
  double time = -1;
  // camera intrinsics of this pose (reference Types.h:46), pinhole [fx, fy, u0, v0]; used instead
  // of the rig camera's when Options::use_per_pose_cam_params is set
  std::vector<Scalar> cam_params;
  // constraint counts (the reference keeps id lists; only emptiness is ever tested
  // outside the solver, BundleAdjuster.cpp:1252-1265)
  uint32_t num_proj_residuals = 0, num_inertial_residuals = 0, num_binary_residuals = 0,
           num_unary_residuals = 0;
};

template <typename Scalar = double, int LmSize = 1>
struct LandmarkT {
  Vector2t z_ref;
  Vector4t x_w;
  int external_id = -1;
  uint32_t num_outlier_residuals = 0, num_proj_residuals = 0;
  uint32_t id = 0, opt_id = 0, ref_pose_id = 0, ref_cam_id = 0;
  bool is_active = true, is_reliable = true;
};

// reference Types.h:246-253: the common head of every residual type
template <typename Scalar, int kParamSize>
struct ResidualT {
  uint32_t residual_id = 0, residual_offset = 0;
  Scalar mahalanobis_distance = 0;
  Scalar weight = 1, orig_weight = 1;
};
// reference Types.h:255-266 / 268-280: the unary prior and the binary pose-pose constraint as the Add*
// calls store them (measurement, information, its square root); the Jacobian blocks dz_dx* and the
// residual vector live only on the device
template <typename Scalar = double>
struct UnaryResidualT : public ResidualT<Scalar, 6> {
  static const uint32_t kResSize = 6;
  uint32_t pose_id = 0;
  SE3 t_wp;
  Matrix6t cov_inv, cov_inv_sqrt;
  bool use_rotation = true;
};
template <typename Scalar = double>
struct BinaryResidualT : public ResidualT<Scalar, 6> {
  static const uint32_t kResSize = 6;
  uint32_t x1_id = 0, x2_id = 0;
  SE3 t_12;
  Matrix6t cov_inv, cov_inv_sqrt;
  bool use_rotation = true;
};

// reference Types.h:246-253, 282-298: what callers read back of a projection residual (the
// cached Jacobian blocks dz_dx_meas / dz_dx_ref / dz_dlm live only on the device)
template <typename Scalar = double, int LmSize = 1>
struct ProjectionResidualT {
  static const uint32_t kResSize = 2;
  uint32_t residual_id = 0, residual_offset = 0;
  Scalar mahalanobis_distance = 0;  // |residual|^2 * weight at the last evaluation
  Scalar weight = 1, orig_weight = 1;
  Vector2t z;
  uint32_t x_meas_id = 0, x_ref_id = 0, landmark_id = 0, cam_id = 0;
  Vector2t residual;                // z - pi at the state the last Solve() left behind
  bool is_conditioning = false;
};

// reference Types.h:91-110: the 3d gravity vector from the 2d direction (pitch, roll) vector
template <typename Scalar = double>
inline Vector3t GetGravityVector(const Vector2t& dir, const Scalar g = (Scalar)Gravity) {
  const double sp = std::sin(dir[0]), cp = std::cos(dir[0]), sq = std::sin(dir[1]), cq = std::cos(dir[1]);
  return Vector3t({-g * cp * sq, g * sp, -g * cp * cq});
}
// reference Types.h:161-180: its 3x2 Jacobian with respect to the direction
template <typename Scalar = double>
inline Mat<3, 2> dGravity_dDirection(const Vector2t& dir, const Scalar g = (Scalar)Gravity) {
  const double sp = std::sin(dir[0]), cp = std::cos(dir[0]), sq = std::sin(dir[1]), cq = std::cos(dir[1]);
  Mat<3, 2> v({-sp * sq, cp * cq, -cp, 0.0, -cq * sp, -cp * sq});
  for (int i = 0; i < 6; ++i) v[i] *= -g;
  return v;
}

// reference Types.h:160-197
template <typename Scalar = double>
struct ImuPoseT {
  ImuPoseT() : time(0) {}
  ImuPoseT(const PoseT<Scalar>& pose) : t_wp(pose.t_wp), v_w(pose.v_w), time(pose.time) {}
  ImuPoseT(const SE3& twp, const Vector3t& v, const Vector3t& w, const double time_)
      : t_wp(twp), v_w(v), w_w(w), time(time_) {}
  SE3 t_wp;       // pose in world coordinates
  Vector3t v_w;   // velocity in world coordinates
  Vector3t w_w;   // angular rates in world coordinates
  double time;    // seconds
};

extern "C" int ba_hip_integrate_imu(const double t_wp7[7], const double v_w3[3], const double bg3[3],
                                    const double ba3[3], const double g3[3], const double* meas7,
                                    uint32_t nmeas, double* states10);
extern "C" int ba_hip_integrate_imu_jacobians(const double t_wp7[7], const double v_w3[3], const double bg3[3],
                                              const double ba3[3], const double g3[3], const double* meas7,
                                              uint32_t nmeas, const double r6[6], double* states10,
                                              double* dpose_db60, double* dpose_dpose100, double* c_res100);
extern "C" int ba_hip_imu_pose_derivative(const double state10[10], const double g3[3], const double z_start7[7],
                                          const double z_end7[7], const double bg3[3], const double ba3[3], double dt,
                                          double k9[9], double* dk_db54, double* dk_dx90);
extern "C" int ba_hip_imu_integrate_pose(const double state10[10], const double k9[9], double dt, double out10[10],
                                         double* dy_dk90, double* dy_dy16);

// reference Types.h:300-321, the host-visible part (the Jacobian / covariance blocks dz_dx1, dz_dx2,
// cov_inv ... live only on the device)
template <typename Scalar = double, int ResidualSize = 15, int PoseSize = 15>
struct ImuResidualT {
  typedef ImuMeasurementT<Scalar> ImuMeasurement;
  typedef ImuPoseT<Scalar> ImuPose;
  static const uint32_t kResSize = ResidualSize;

  // reference Types.h:643-738: RK4 integration of the samples from `pose`; `poses` receives the start
  // state and the state at every later sample.  The Jacobian outputs of the reference's signature:
  // dpose_db (state [t q v] over the two biases), dpose_dpose (over the start state), the covariance
  // c_res (in/out) with the noise diagonal r — formed, as there, only when a Jacobian is asked for and
  // r is given.  Runs on the host with the code the device kernels use (libba_hip.so).
  static ImuPose IntegrateResidual(const ImuPose& pose, const std::vector<ImuMeasurement>& measurements,
                                   const Vector3t& bg, const Vector3t& ba, const Vector3t& g,
                                   std::vector<ImuPose>& poses, Mat<10, 6>* dpose_db = nullptr,
                                   Mat<10, 10>* dpose_dpose = nullptr, Mat<10, 10>* c_res = nullptr,
                                   const Vector6t* r = nullptr) {
    const size_t n = measurements.size();
    std::vector<double> m(7 * std::max<size_t>(n, 1)), st(10 * std::max<size_t>(n, 1));
    for (size_t i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) { m[7 * i + k] = measurements[i].w[k]; m[7 * i + 3 + k] = measurements[i].a[k]; }
      m[7 * i + 6] = measurements[i].time;
    }
    double t7[7];
    pose.t_wp.to7(t7);
    const double v[3] = {pose.v_w[0], pose.v_w[1], pose.v_w[2]}, b1[3] = {bg[0], bg[1], bg[2]},
                 b2[3] = {ba[0], ba[1], ba[2]}, gg[3] = {g[0], g[1], g[2]};
    if (dpose_db || dpose_dpose) {
      double r6[6];
      if (r) for (int k = 0; k < 6; ++k) r6[k] = (*r)[k];
      ba_hip_integrate_imu_jacobians(t7, v, b1, b2, gg, m.data(), (uint32_t)n, r ? r6 : nullptr, st.data(),
                                     dpose_db ? dpose_db->data() : nullptr,
                                     dpose_dpose ? dpose_dpose->data() : nullptr, c_res ? c_res->data() : nullptr);
    } else {
      ba_hip_integrate_imu(t7, v, b1, b2, gg, m.data(), (uint32_t)n, st.data());
    }
    poses.clear();
    const size_t rows = std::max<size_t>(n, 1);
    for (size_t i = 0; i < rows; ++i) {
      ImuPose p;
      p.t_wp = SE3::from7(&st[10 * i]);
      for (int k = 0; k < 3; ++k) p.v_w[k] = st[10 * i + 7 + k];
      p.w_w = pose.w_w;
      p.time = (i < n) ? measurements[i].time : pose.time;
      poses.push_back(p);
    }
    return poses.back();
  }
  static ImuPose IntegrateResidual(const PoseT<Scalar>& pose, const std::vector<ImuMeasurement>& measurements,
                                   const Vector3t& bg, const Vector3t& ba, const Vector3t& g,
                                   std::vector<ImuPose>& poses, Mat<10, 6>* dpose_db = nullptr,
                                   Mat<10, 10>* dpose_dpose = nullptr, Mat<10, 10>* c_res = nullptr,
                                   const Vector6t* r = nullptr) {
    return IntegrateResidual(ImuPose(pose), measurements, bg, ba, g, poses, dpose_db, dpose_dpose, c_res, r);
  }
  // reference Types.h:419-643: one RK4 step between two samples; dy_db / dy_dy0 are the Jacobians of the
  // step over the biases / the start state (formed when both are given; c_prior with r: the covariance)
  static ImuPose IntegrateImu(const ImuPose& pose, const ImuMeasurement& z_start, const ImuMeasurement& z_end,
                              const Vector3t& bg, const Vector3t& ba, const Vector3t& g,
                              Mat<10, 6>* dy_db = nullptr, Mat<10, 10>* dy_dy0 = nullptr,
                              Mat<10, 10>* c_prior = nullptr, const Vector6t* r = nullptr) {
    std::vector<ImuMeasurement> two = {z_start, z_end};
    std::vector<ImuPose> out;
    if (dy_db && dy_dy0) {
      // the covariance update needs a noise matrix; without one the Jacobians alone (zero noise)
      const Vector6t r0;
      return IntegrateResidual(pose, two, bg, ba, g, out, dy_db, dy_dy0, r ? c_prior : nullptr, r ? r : &r0);
    }
    return IntegrateResidual(pose, two, bg, ba, g, out);
  }
  // reference Types.h:376-416: k = [v; R (w + b_g); R (a + b_a) - g] with the two samples interpolated at
  // z_start.time + dt; dk_db (9 x 6), dk_dx (9 x 10 over [t q v]) optional
  static Mat<9, 1> GetPoseDerivative(const ImuPose& pose, const Vector3t& g_w, const ImuMeasurement& z_start,
                                     const ImuMeasurement& z_end, const Vector3t& bg, const Vector3t& ba,
                                     const Scalar dt, Mat<9, 6>* dk_db = nullptr, Mat<9, 10>* dk_dx = nullptr) {
    double s10[10], z0[7], z1[7], k[9];
    State10(pose, s10);
    for (int i = 0; i < 3; ++i) { z0[i] = z_start.w[i]; z0[3 + i] = z_start.a[i]; z1[i] = z_end.w[i]; z1[3 + i] = z_end.a[i]; }
    z0[6] = z_start.time; z1[6] = z_end.time;
    const double gg[3] = {g_w[0], g_w[1], g_w[2]}, b1[3] = {bg[0], bg[1], bg[2]}, b2[3] = {ba[0], ba[1], ba[2]};
    ba_hip_imu_pose_derivative(s10, gg, z0, z1, b1, b2, (double)dt, k, dk_db ? dk_db->data() : nullptr,
                               dk_dx ? dk_dx->data() : nullptr);
    Mat<9, 1> out;
    for (int i = 0; i < 9; ++i) out[i] = k[i];
    return out;
  }
  // reference Types.h:324-373: the state advanced by k * dt (rotation: q <- exp(k_w dt) q, not renormalised);
  // pdy_dk (10 x 9) and the quaternion block pdy_dy (4 x 4) optional
  static ImuPose IntegratePose(const ImuPose& pose, const Mat<9, 1>& k, const Scalar dt, Mat<10, 9>* pdy_dk = nullptr,
                               Mat<4, 4>* pdy_dy = nullptr) {
    double s10[10], k9[9], o[10];
    State10(pose, s10);
    for (int i = 0; i < 9; ++i) k9[i] = k[i];
    ba_hip_imu_integrate_pose(s10, k9, (double)dt, o, pdy_dk ? pdy_dk->data() : nullptr, pdy_dy ? pdy_dy->data() : nullptr);
    ImuPose y = pose;
    y.t_wp = SE3::from7(o);  // the raw quaternion, as the reference keeps it (memcpy, Types.h:338-339)
    for (int i = 0; i < 3; ++i) y.v_w[i] = o[7 + i];
    return y;
  }
 private:
  static void State10(const ImuPose& pose, double* s10) {
    pose.t_wp.to7(s10);
    for (int i = 0; i < 3; ++i) s10[7 + i] = pose.v_w[i];
  }
 public:

  uint32_t residual_id = 0, residual_offset = 0;
  uint32_t pose1_id = 0, pose2_id = 0;
  Scalar mahalanobis_distance = 0;
  Scalar weight = 1, orig_weight = 1;
  std::vector<ImuMeasurement> measurements;
  // residual at the state the last Solve() left behind: translation, rotation, velocity
  // [, gyro bias, accelerometer bias] (Types.h:654-689); the first kResSize entries are used
  Scalar residual[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

// reference Types.h:112-159 (fields used on the hot path)
template <typename Scalar = double>
struct ImuCalibrationT {
  SE3 t_vs;
  Vector3t b_g, b_a;
  Vector2t g;
  Vector3t g_vec;
  Vector6t r;    // diagonal of the measurement noise
  Vector6t r_b;  // bias random walk
  ImuCalibrationT() {
    g_vec[0] = 0; g_vec[1] = 0; g_vec[2] = -Gravity;  // GetGravityVector((0,0))
    for (int i = 0; i < 3; ++i) {
      r[i] = IMU_GYRO_SIGMA * IMU_GYRO_SIGMA; r[3 + i] = IMU_ACCEL_SIGMA * IMU_ACCEL_SIGMA;
      r_b[i] = IMU_GYRO_BIAS_SIGMA * IMU_GYRO_BIAS_SIGMA;
      r_b[3 + i] = IMU_ACCEL_BIAS_SIGMA * IMU_ACCEL_BIAS_SIGMA;
    }
  }
};

}  // namespace ba
