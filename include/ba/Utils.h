// Lie-group / quaternion helpers of the reference's include/ba/Utils.h with the reference's names,
// on the value types of Types.h (ba::SE3 = translation + quaternion x,y,z,w; ba::Mat<R, C> row-major).
// The arithmetic is the host build of what the gfx950 kernels evaluate (ba_amd/csrc/dmath.h, dpose.h,
// reached through ba_hip_lie of include/ba_hip.h): one source for device and host.
//
//   reference                          here
//   Eigen::Quaternion<Scalar>          ba::Vector4t (x, y, z, w) or anything indexable alike
//   Sophus::SE3Group<Scalar>           ba::SE3 (converts from Sophus types, Types.h)
//
// Not restated: dlog_dx, dlog_dr, dlog_dw, dt1t2_dt1 (unused in the reference's sources), dlog_dse3
// (applications/math_test only), the Eigen::IOFormat constants, StreamMessage.
#pragma once
#include <sys/time.h>

#include "Types.h"

extern "C" int ba_hip_lie(int op, const double* a, const double* b, double* out);

namespace ba {

// reference Utils.h:86-98
template <typename Scalar = double>
inline Scalar powi(const Scalar x, const int y) {
  if (y == 0) return 1.0;
  if (y < 0) return 1.0 / powi(x, -y);
  Scalar r = x;
  for (int i = 1; i < y; ++i) r *= x;
  return r;
}
// reference Utils.h:102-110
inline double Tic() {
  struct timeval tv;
  gettimeofday(&tv, 0);
  return tv.tv_sec + 1e-6 * (tv.tv_usec);
}
inline double Toc(double tic) { return Tic() - tic; }

namespace lie_detail {
inline void se3(const SE3& t, double* p) { t.to7(p); }
template <typename Q> inline void quat(const Q& q, double* p) { for (int i = 0; i < 4; ++i) p[i] = q[i]; }
template <int R, int C> inline Mat<R, C> call(int op, const double* a, const double* b) {
  Mat<R, C> o;
  double buf[64];
  const int n = ba_hip_lie(op, a, b, buf);
  for (int i = 0; i < n && i < R * C; ++i) o.data()[i] = buf[i];
  return o;
}
}  // namespace lie_detail

// reference Utils.h:72-82: (R x[0:3] + t x[3], x[3])
inline Vector4t MultHomogeneous(const SE3& lhs, const Vector4t& rhs) {
  double a[7]; lie_detail::se3(lhs, a);
  return lie_detail::call<4, 1>(17, a, rhs.data());
}
// reference Utils.h:137-185: d log(q) / dq, 3 x 4
template <typename Q> inline Mat<3, 4> dlog_dq(const Q& q) {
  double a[4]; lie_detail::quat(q, a);
  return lie_detail::call<3, 4>(1, a, nullptr);
}
// reference Utils.h:252-266: d exp(w) / dw as a quaternion, 4 x 3
inline Mat<4, 3> dq_exp_dw(const Vector3t& w) { return lie_detail::call<4, 3>(2, w.data(), nullptr); }
// reference Utils.h:270-273
inline Mat<4, 4> dqinv_dq() {
  Mat<4, 4> m;
  m(0, 0) = m(1, 1) = m(2, 2) = -1; m(3, 3) = 1;
  return m;
}
// reference Utils.h:277-291: d(q1 q2)/dq2 (takes q1) and d(q1 q2)/dq1 (takes q2), 4 x 4
template <typename Q> inline Mat<4, 4> dq1q2_dq2(const Q& q1) {
  double a[4]; lie_detail::quat(q1, a);
  return lie_detail::call<4, 4>(4, a, nullptr);
}
template <typename Q> inline Mat<4, 4> dq1q2_dq1(const Q& q2) {
  double a[4]; lie_detail::quat(q2, a);
  return lie_detail::call<4, 4>(3, a, nullptr);
}
// reference Utils.h:295-333: d(R(q) x)/dq, 3 x 4 (the polynomial form, valid off the unit sphere) and d(R(q) x)/dx
template <typename Q> inline Mat<3, 4> dqx_dq(const Q& q, const Vector3t& x) {
  double a[4]; lie_detail::quat(q, a);
  return lie_detail::call<3, 4>(5, a, x.data());
}
template <typename Q> inline Matrix3t dqx_dx(const Q& q) {
  double a[4]; lie_detail::quat(q, a);
  return lie_detail::call<3, 3>(6, a, nullptr);
}
// reference Utils.h:354-369: (t_a - t_b, log(R_a R_b^-1)) and (R_a exp(x[3:6]), t_a + x[0:3])
inline Vector6t log_decoupled(const SE3& a, const SE3& b) {
  double pa[7], pb[7]; lie_detail::se3(a, pa); lie_detail::se3(b, pb);
  return lie_detail::call<6, 1>(7, pa, pb);
}
inline SE3 exp_decoupled(const SE3& a, const Vector6t& x) {
  double pa[7], o[64]; lie_detail::se3(a, pa);
  ba_hip_lie(8, pa, x.data(), o);
  return SE3::from7(o);
}
// reference Utils.h:374-447
inline Matrix6t dlog_decoupled_dx(const SE3& a, const SE3& b) {
  double pa[7], pb[7]; lie_detail::se3(a, pa); lie_detail::se3(b, pb);
  return lie_detail::call<6, 6>(9, pa, pb);
}
inline Mat<6, 7> dLog_decoupled_dt1(const SE3& t1, const SE3& t2) {
  double pa[7], pb[7]; lie_detail::se3(t1, pa); lie_detail::se3(t2, pb);
  return lie_detail::call<6, 7>(10, pa, pb);
}
inline Mat<6, 7> dlog_decoupled_dt2(const SE3& t1, const SE3& t2) {
  double pa[7], pb[7]; lie_detail::se3(t1, pa); lie_detail::se3(t2, pb);
  return lie_detail::call<6, 7>(11, pa, pb);
}
// reference Utils.h:451-536: d exp_decoupled(t, x)/dx and d exp_decoupled(t, x)^-1/dx at x = 0, 7 x 6
inline Mat<7, 6> dexp_decoupled_dx(const SE3& t) {
  double pa[7]; lie_detail::se3(t, pa);
  return lie_detail::call<7, 6>(12, pa, nullptr);
}
inline Mat<7, 6> dinv_exp_decoupled_dx(const SE3& t) {
  double pa[7]; lie_detail::se3(t, pa);
  return lie_detail::call<7, 6>(13, pa, nullptr);
}
// reference Utils.h:540-694: d(T x)/d(t, q) 4 x 7; d(T1 T2)/dT1 and d(T1 T2)/dT2, 7 x 7
inline Mat<4, 7> dt_x_dt(const SE3& t, const Vector4t& x) {
  double pa[7]; lie_detail::se3(t, pa);
  return lie_detail::call<4, 7>(14, pa, x.data());
}
inline Mat<7, 7> dt1_t2_dt1(const SE3& t1, const SE3& t2) {
  double pa[7], pb[7]; lie_detail::se3(t1, pa); lie_detail::se3(t2, pb);
  return lie_detail::call<7, 7>(15, pa, pb);
}
inline Mat<7, 7> dt1_t2_dt2(const SE3& t1) {
  double pa[7]; lie_detail::se3(t1, pa);
  return lie_detail::call<7, 7>(16, pa, nullptr);
}

}  // namespace ba
