// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h).
// Restatement of the Lie-group / quaternion Jacobian helpers of
// /root/reference/include/ba/Utils.h and of the Calibu pinhole camera calls the hot
// path makes.  Each function cites the reference lines it follows.  The formulas
// are written from their mathematical definition (partial derivatives of the
// polynomial rotation matrix, left/right quaternion multiplication matrices, ...)
// and reproduce the reference's choices, including its truncated series.
#pragma once
#include "omath.h"

namespace orc {

// Utils.h:85-100  powi
inline double powi(double x, int y) {
  if (y == 0) return 1.0;
  if (y < 0) return 1.0 / powi(x, -y);
  double r = x;
  for (int i = 1; i < y; ++i) r *= x;
  return r;
}

// Utils.h:72-82  MultHomogeneous: (R x[0:3] + t x[3], x[3])
inline Vec4 MultHomogeneous(const SE3& lhs, const Vec4& rhs) {
  Vec3 h; h[0] = rhs[0]; h[1] = rhs[1]; h[2] = rhs[2];
  const Vec3 o = lhs.so3() * h + lhs.translation() * rhs[3];
  Vec4 out; out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = rhs[3];
  return out;
}

// Utils.h:137-185  dlog_dq: d(so3 log)/d(quaternion coeffs x,y,z,w), 3x4.
// log(q) = f(n,w) v with f = 2 atan(n/w)/n, n = |v|.  Nominal branch: the exact
// derivative f I + v v^T (df/dn)/n | v df/dw.  Small-angle branch (n < 1e-9): the
// derivative of the series f = 2/w - 2 n^2/w^3 the reference uses.
inline Mat<3, 4> dlog_dq(const Quat& q) {
  const double v[3] = {q.x, q.y, q.z};
  const double w = q.w;
  const double n2 = powi(q.x, 2) + powi(q.y, 2) + powi(q.z, 2);
  const double n = std::sqrt(n2);
  Mat<3, 4> J;
  if (n < 1e-9) {
    const double inv_w3 = 1.0 / powi(w, 3);
    const double two_n2 = 2 * n2;
    const double dcol = (3 * two_n2) / powi(w, 4) - 2 / powi(w, 2);
    const double diag = 2 / w - two_n2 * inv_w3;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) J(i, j) = -4 * v[i] * v[j] * inv_w3;
      J(i, i) += diag;
      J(i, 3) = v[i] * dcol;
    }
  } else {
    const double c = 1.0 / (n2 / powi(w, 2) + 1.0);  // 1/(1+n^2/w^2)
    const double at = std::atan(std::sqrt(n2) / w);
    const double inv_n3 = 1.0 / std::pow(n2, 3.0 / 2.0);
    const double inv_n2 = 1.0 / n2;
    const double inv_w = 1.0 / w;
    const double f = (2 * at) / std::sqrt(n2);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j)
        J(i, j) = 2 * v[i] * v[j] * c * inv_n2 * inv_w - 2 * v[i] * v[j] * at * inv_n3;
      J(i, i) = f - 2 * powi(v[i], 2) * at * inv_n3 + 2 * powi(v[i], 2) * c * inv_n2 * inv_w;
      J(i, 3) = -(2 * v[i] * c) / powi(w, 2);
    }
  }
  return J;
}

// Utils.h:252-266  dq_exp_dw: d(quaternion of exp(w))/dw, 4x3, the reference's
// truncated series (note its (t/20 - 1)/24 factor, kept as is).  At w = 0 it is the
// constant [I/2; 0].
inline Mat<4, 3> dq_exp_dw(const Vec3& w) {
  const double t = w.norm();
  const double a = t / 20 - 1;
  const double b = powi(t, 2) / 48 - 0.5;
  const double t2 = powi(t, 2);
  Mat<4, 3> J;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) J(i, j) = (a * w[i] * w[j]) / 24;
    J(i, i) = (a * powi(w[i], 2)) / 24 - t2 / 48 + 0.5;
    J(3, i) = (b * w[i]) / 2;
  }
  return J;
}

// Utils.h:270-273  dqinv_dq = diag(-1,-1,-1,1)
inline Mat4 dqinv_dq() {
  Mat4 m;
  m(0, 0) = -1; m(1, 1) = -1; m(2, 2) = -1; m(3, 3) = 1;
  return m;
}

// Utils.h:277-282  dq1q2_dq2(q1): left-multiplication matrix L(q1), q1 (x) q2 = L q2.
inline Mat4 dq1q2_dq2(const Quat& a) {
  Mat4 m;
  const double r[4][4] = {{a.w, -a.z, a.y, a.x},
                          {a.z, a.w, -a.x, a.y},
                          {-a.y, a.x, a.w, a.z},
                          {-a.x, -a.y, -a.z, a.w}};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = r[i][j];
  return m;
}

// Utils.h:286-291  dq1q2_dq1(q2): right-multiplication matrix R(q2), q1 (x) q2 = R q1.
inline Mat4 dq1q2_dq1(const Quat& b) {
  Mat4 m;
  const double r[4][4] = {{b.w, b.z, -b.y, b.x},
                          {-b.z, b.w, b.x, b.y},
                          {b.y, -b.x, b.w, b.z},
                          {-b.x, -b.y, -b.z, b.w}};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = r[i][j];
  return m;
}

// Utils.h:295-312  dqx_dq(q, v): d(R(q) v)/dq, 3x4 — the plain partial derivatives
// of the polynomial form R(q) = [1-2(y^2+z^2), 2(xy-wz), ...] (not the homogeneous
// |q|^2 form), columns ordered x,y,z,w.
inline Mat<3, 4> dqx_dq(const Quat& q, const Vec3& p) {
  const double x = p[0], y = p[1], z = p[2];
  Mat<3, 4> J;
  // d/dqx
  J(0, 0) = 2 * q.y * y + 2 * q.z * z;
  J(1, 0) = 2 * q.y * x - 4 * q.x * y - 2 * q.w * z;
  J(2, 0) = 2 * q.z * x + 2 * q.w * y - 4 * q.x * z;
  // d/dqy
  J(0, 1) = 2 * q.x * y - 4 * q.y * x + 2 * q.w * z;
  J(1, 1) = 2 * q.x * x + 2 * q.z * z;
  J(2, 1) = 2 * q.z * y - 2 * q.w * x - 4 * q.y * z;
  // d/dqz
  J(0, 2) = 2 * q.x * z - 2 * q.w * y - 4 * q.z * x;
  J(1, 2) = 2 * q.y * z + 2 * q.w * x - 4 * q.z * y;
  J(2, 2) = 2 * q.y * y + 2 * q.x * x;
  // d/dqw
  J(0, 3) = 2 * q.y * z - 2 * q.z * y;
  J(1, 3) = 2 * q.z * x - 2 * q.x * z;
  J(2, 3) = 2 * q.x * y - 2 * q.y * x;
  return J;
}

// Utils.h:354-360  log_decoupled(a,b) = (t_a - t_b, log(R_a R_b^-1))
inline Vec6 log_decoupled(const SE3& a, const SE3& b) {
  Vec6 r;
  const Vec3 dt = a.translation() - b.translation();
  const Vec3 w = (a.so3() * b.so3().inverse()).log();
  for (int i = 0; i < 3; ++i) { r[i] = dt[i]; r[3 + i] = w[i]; }
  return r;
}

// Utils.h:364-369  exp_decoupled(a,x) = (R_a exp(x[3:6]), t_a + x[0:3])
inline SE3 exp_decoupled(const SE3& a, const Vec6& x) {
  Vec3 dt, w;
  for (int i = 0; i < 3; ++i) { dt[i] = x[i]; w[i] = x[3 + i]; }
  return SE3(a.so3() * SO3::exp(w), a.translation() + dt);
}

// Utils.h:374-384  dlog_decoupled_dx(a,b): I6 with the rotation block
// dlog_dq(q_a q_b^-1) L(q_a) R(q_b^-1) dq_exp_dw(0)
inline Mat<6, 6> dlog_decoupled_dx(const SE3& a, const SE3& b) {
  Mat<6, 6> J = Mat<6, 6>::Identity();
  const Mat3 rot = dlog_dq((a * b.inverse()).unit_quaternion()) *
                   dq1q2_dq2(a.unit_quaternion()) *
                   dq1q2_dq1(b.inverse().unit_quaternion()) *
                   dq_exp_dw(Vec3::Zero());
  J.setBlock<3, 3>(3, 3, rot);
  return J;
}

// Utils.h:388-397  dLog_decoupled_dt1(t1,t2), 6x7
inline Mat<6, 7> dLog_decoupled_dt1(const SE3& t1, const SE3& t2) {
  Mat<6, 7> J;
  J.setBlock<3, 3>(0, 0, Mat3::Identity());
  const Mat<3, 4> b = dlog_dq((t1 * t2.inverse()).unit_quaternion()) *
                      dq1q2_dq1(t2.inverse().unit_quaternion());
  J.setBlock<3, 4>(3, 3, b);
  return J;
}

// Utils.h:401-447  dlog_decoupled_dt2(t1,t2), 6x7
inline Mat<6, 7> dlog_decoupled_dt2(const SE3& t1, const SE3& t2) {
  Mat<6, 7> J;
  const Quat qlog = (t1.so3() * t2.so3().inverse()).unit_quaternion();
  J.setBlock<3, 3>(0, 0, -Mat3::Identity());
  const Mat<3, 4> b = dlog_dq(qlog) * dq1q2_dq2(t1.unit_quaternion()) * dqinv_dq();
  J.setBlock<3, 4>(3, 3, b);
  return J;
}

// Utils.h:451-489  dexp_decoupled_dx(t), 7x6
inline Mat<7, 6> dexp_decoupled_dx(const SE3& t) {
  Mat<7, 6> J;
  J.setBlock<3, 3>(0, 0, Mat3::Identity());
  const Mat<4, 3> b = dq1q2_dq2(t.unit_quaternion()) * dq_exp_dw(Vec3::Zero());
  J.setBlock<4, 3>(3, 3, b);
  return J;
}

// Utils.h:493-536  dinv_exp_decoupled_dx(t), 7x6:
// d (exp_decoupled(t,x))^-1 / dx at x = 0.
inline Mat<7, 6> dinv_exp_decoupled_dx(const SE3& t) {
  const Mat<4, 3> dq_exp = dq_exp_dw(Vec3::Zero());
  const Quat qt_inv = t.so3().inverse().unit_quaternion();
  const Mat4 rq = dq1q2_dq1(qt_inv);
  Mat<7, 6> J;
  J.setBlock<3, 3>(0, 0, -t.so3().inverse().matrix());
  const Mat3 tr = dqx_dq(qt_inv, t.translation()) * rq * dq_exp;
  J.setBlock<3, 3>(0, 3, tr);
  const Mat<4, 3> br = rq * (-dq_exp);
  J.setBlock<4, 3>(3, 3, br);
  return J;
}

// Utils.h:540-583  dt_x_dt(t,x), 4x7: d(T x)/d(t,q) for homogeneous x.
inline Mat<4, 7> dt_x_dt(const SE3& t, const Vec4& x) {
  Mat<4, 7> J;
  J.setBlock<3, 3>(0, 0, Mat3::Identity() * x[3]);
  Vec3 h; h[0] = x[0]; h[1] = x[1]; h[2] = x[2];
  J.setBlock<3, 4>(0, 3, dqx_dq(t.unit_quaternion(), h));
  return J;
}

// Utils.h:587-639  dt1_t2_dt1(t1,t2), 7x7
inline Mat<7, 7> dt1_t2_dt1(const SE3& t1, const SE3& t2) {
  Mat<7, 7> J;
  J.setBlock<3, 3>(0, 0, Mat3::Identity());
  J.setBlock<3, 4>(0, 3, dqx_dq(t1.unit_quaternion(), t2.translation()));
  J.setBlock<4, 4>(3, 3, dq1q2_dq1(t2.unit_quaternion()));
  return J;
}

// Utils.h:643-694  dt1_t2_dt2(t1), 7x7
inline Mat<7, 7> dt1_t2_dt2(const SE3& t1) {
  Mat<7, 7> J;
  J.setBlock<3, 3>(0, 0, t1.rotationMatrix());
  J.setBlock<4, 4>(3, 3, dq1q2_dq2(t1.unit_quaternion()));
  return J;
}

// ---------------------------------------------------------------------------
// Calibu camera models as the hot path uses them: "LinearCamera" (pinhole, params fx,fy,u0,v0)
// and "FovCamera" (params fx,fy,u0,v0,w — the five parameters of the reference's CalibSize = 5
// instantiations, BundleAdjuster.cpp:1816-1826, BundleAdjuster.h:758-759).
// Calibu 0.1 is not in the reference tree; the pinhole semantics are reconstructed from the call
// sites parallel_algos.h:59-62,73-78, the FOV model is the published one
// (Devernay & Faugeras 2001, "Straight lines have to be straight": r_d = atan(2 r_u tan(w/2)) / w)
// in the multiplicative form calibu applies it — pix = K (factor(r_u) p), p = P.xy / P.z — with the
// limit values taken for a vanishing radius or a vanishing w.  PARITY UNPINNED for both; every
// derivative below is pinned by finite differences of the function it differentiates.
struct Pinhole {
  double fx = 1, fy = 1, u0 = 0, v0 = 0;
  double w = 0;   // FOV distortion parameter (model 1)
  int model = 0;  // 0 LinearCamera, 1 FovCamera
  static constexpr double kSmall = 1e-5;  // squared radius / squared w below which the limits are used
  int NumParams() const { return model == 1 ? 5 : 4; }
  // distortion factor r_d / r_u and its derivatives over the radius and over w
  void Factor(double r, double* f, double* df_dr, double* df_dw) const {
    *f = 1.0; *df_dr = 0.0; *df_dw = 0.0;
    if (model != 1 || w * w <= kSmall) return;
    const double th = std::tan(0.5 * w), m = 2.0 * th, dm = 1.0 + th * th;
    if (r * r < kSmall) {
      *f = m / w;
      *df_dw = dm / w - m / (w * w);
      return;
    }
    const double at = std::atan(r * m), den = 1.0 + r * r * m * m;
    *f = at / (r * w);
    *df_dr = m / (den * r * w) - at / (r * r * w);
    *df_dw = dm / (den * w) - at / (r * w * w);
  }
  // inverse factor r_u / r_d over the distorted radius
  void FactorInv(double rd, double* g, double* dg_dr, double* dg_dw) const {
    *g = 1.0; *dg_dr = 0.0; *dg_dw = 0.0;
    if (model != 1 || w * w <= kSmall) return;
    const double th = std::tan(0.5 * w), m = 2.0 * th, dm = 1.0 + th * th;
    if (rd * rd < kSmall) {
      *g = w / m;
      *dg_dw = 1.0 / m - w * dm / (m * m);
      return;
    }
    const double tn = std::tan(rd * w), sec2 = 1.0 + tn * tn;
    *g = tn / (rd * m);
    *dg_dr = w * sec2 / (rd * m) - tn / (rd * rd * m);
    *dg_dw = sec2 / m - tn * dm / (rd * m * m);
  }
  Vec2 Project(const Vec3& P) const {
    const double px = P[0] / P[2], py = P[1] / P[2];
    double f, dr, dw;
    Factor(std::sqrt(px * px + py * py), &f, &dr, &dw);
    Vec2 p;
    if (model == 1) { p[0] = fx * (f * px) + u0; p[1] = fy * (f * py) + v0; }
    else { p[0] = fx * P[0] / P[2] + u0; p[1] = fy * P[1] / P[2] + v0; }
    return p;
  }
  Mat<2, 3> dProject_dP(const Vec3& P) const {
    Mat<2, 3> d;
    if (model != 1) {
      d(0, 0) = fx / P[2]; d(0, 2) = -fx * P[0] / (P[2] * P[2]);
      d(1, 1) = fy / P[2]; d(1, 2) = -fy * P[1] / (P[2] * P[2]);
      return d;
    }
    const double iz = 1.0 / P[2], px = P[0] * iz, py = P[1] * iz, r = std::sqrt(px * px + py * py);
    double f, dr, dw;
    Factor(r, &f, &dr, &dw);
    // d (f p) / d p = f I + (df/dr) p p^T / r
    const double k = r > 0.0 ? dr / r : 0.0;
    const double a00 = f + k * px * px, a01 = k * px * py, a11 = f + k * py * py;
    // d p / d P = [[iz, 0, -px iz], [0, iz, -py iz]]
    d(0, 0) = fx * a00 * iz; d(0, 1) = fx * a01 * iz; d(0, 2) = -fx * (a00 * px + a01 * py) * iz;
    d(1, 0) = fy * a01 * iz; d(1, 1) = fy * a11 * iz; d(1, 2) = -fy * (a01 * px + a11 * py) * iz;
    return d;
  }
  // Transfer3d(T_ba, ray, rho) = Project(R ray + rho t)
  Vec2 Transfer3d(const SE3& t_ba, const Vec3& ray, double rho) const {
    return Project(t_ba.so3() * ray + t_ba.translation() * rho);
  }
  // dTransfer3d_dray(T_ba, ray, rho) = [dProject R, dProject t]  (2x4)
  Mat<2, 4> dTransfer3d_dray(const SE3& t_ba, const Vec3& ray, double rho) const {
    const Vec3 P = t_ba.so3() * ray + t_ba.translation() * rho;
    const Mat<2, 3> dp = dProject_dP(P);
    const Mat<2, 3> a = dp * t_ba.so3().matrix();
    const Vec2 b = dp * t_ba.translation();
    Mat<2, 4> J;
    J.setBlock<2, 3>(0, 0, a);
    J(0, 3) = b[0]; J(1, 3) = b[1];
    return J;
  }
  // Unproject(pix): the ray with z = 1
  Vec3 Unproject(const Vec2& pix) const {
    Vec3 r;
    r[0] = (pix[0] - u0) / fx; r[1] = (pix[1] - v0) / fy; r[2] = 1.0;
    if (model == 1) {
      double g, dr, dw;
      FactorInv(std::sqrt(r[0] * r[0] + r[1] * r[1]), &g, &dr, &dw);
      r[0] *= g; r[1] *= g;
    }
    return r;
  }
  // d Unproject(pix).xy / d params (2 x 5; column 4 is zero for the pinhole)
  Mat<2, 5> dUnproject_dparams(const Vec2& pix) const {
    const double dx = (pix[0] - u0) / fx, dy = (pix[1] - v0) / fy;
    double g = 1.0, dg_dr = 0.0, dg_dw = 0.0;
    const double rd = std::sqrt(dx * dx + dy * dy);
    if (model == 1) FactorInv(rd, &g, &dg_dr, &dg_dw);
    // d ray.xy / d d = g I + (dg/dr) d d^T / rd
    const double k = rd > 0.0 ? dg_dr / rd : 0.0;
    const double a00 = g + k * dx * dx, a01 = k * dx * dy, a11 = g + k * dy * dy;
    Mat<2, 5> J;
    J(0, 0) = a00 * (-dx / fx); J(1, 0) = a01 * (-dx / fx);
    J(0, 1) = a01 * (-dy / fy); J(1, 1) = a11 * (-dy / fy);
    J(0, 2) = a00 * (-1.0 / fx); J(1, 2) = a01 * (-1.0 / fx);
    J(0, 3) = a01 * (-1.0 / fy); J(1, 3) = a11 * (-1.0 / fy);
    J(0, 4) = dx * dg_dw; J(1, 4) = dy * dg_dw;
    return J;
  }
  // d Project(P) / d params at a fixed point (2 x 5)
  Mat<2, 5> dProject_dparams(const Vec3& P) const {
    const double px = P[0] / P[2], py = P[1] / P[2];
    double f, dr, dw;
    Factor(std::sqrt(px * px + py * py), &f, &dr, &dw);
    Mat<2, 5> J;
    J(0, 0) = f * px; J(1, 1) = f * py;
    J(0, 2) = 1.0; J(1, 3) = 1.0;
    J(0, 4) = fx * px * dw; J(1, 4) = fy * py * dw;
    return J;
  }
  // dTransfer_dparams(T_ba, pix, rho) = d/dparams Project(R Unproject(pix) + rho t): both the
  // un-projection and the projection depend on the parameters (2 x NumParams, held as 2 x 5).  Call
  // site parallel_algos.h:115-118; Calibu is absent from the reference tree — reconstructed as the
  // derivative of the function the call site names, pinned by finite differences of that function.
  Mat<2, 5> dTransfer_dparams(const SE3& t_ba, const Vec2& pix, double rho) const {
    const Vec3 ray = Unproject(pix);
    const Vec3 P = t_ba.so3() * ray + t_ba.translation() * rho;
    const Mat<2, 3> dp = dProject_dP(P);
    const Mat3 R = t_ba.so3().matrix();
    Mat<2, 5> J;
    if (model != 1) {  // the pinhole expressions of the earlier rounds, operation for operation
      const double dx_dfx = -(pix[0] - u0) / (fx * fx), dy_dfy = -(pix[1] - v0) / (fy * fy);
      for (int r = 0; r < 2; ++r) {
        double c0 = 0, c1 = 0;  // dp * R column 0 / column 1
        for (int k = 0; k < 3; ++k) { c0 += dp(r, k) * R(k, 0); c1 += dp(r, k) * R(k, 1); }
        J(r, 0) = c0 * dx_dfx; J(r, 1) = c1 * dy_dfy;
        J(r, 2) = c0 * (-1.0 / fx); J(r, 3) = c1 * (-1.0 / fy);
      }
      J(0, 0) += P[0] / P[2]; J(1, 1) += P[1] / P[2];
      J(0, 2) += 1.0; J(1, 3) += 1.0;
      return J;
    }
    const Mat<2, 5> du = dUnproject_dparams(pix), dq = dProject_dparams(P);
    for (int r = 0; r < 2; ++r) {
      double c0 = 0, c1 = 0;
      for (int k = 0; k < 3; ++k) { c0 += dp(r, k) * R(k, 0); c1 += dp(r, k) * R(k, 1); }
      for (int c = 0; c < 5; ++c) J(r, c) = dq(r, c) + (c0 * du(0, c) + c1 * du(1, c));
    }
    return J;
  }
};

}  // namespace orc
