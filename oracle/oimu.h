// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h).
// Restatement of the RK4 IMU integrator with bias/state Jacobians and Euler-form
// covariance propagation of /root/reference/include/ba/Types.h:324-738.
#pragma once
#include "outils.h"

namespace orc {

// Types.h:222-244
struct ImuMeasurement {
  Vec3 w, a;
  double time;
};

// Types.h:182-220
struct ImuPose {
  SE3 t_wp;
  Vec3 v_w, w_w;
  double time;
};

// Types.h:324-373  IntegratePose.  The rotated quaternion is stored WITHOUT
// renormalisation (the reference memcpy's the raw product, Types.h:336-339).
inline ImuPose IntegratePose(const ImuPose& pose, const Mat<9, 1>& k, double dt,
                             Mat<10, 9>* pdy_dk = nullptr, Mat4* pdy_dy = nullptr) {
  Vec3 kv, kw, ka;
  for (int i = 0; i < 3; ++i) { kv[i] = k[i]; kw[i] = k[3 + i]; ka[i] = k[6 + i]; }
  const SO3 r_v2_v1 = SO3::exp(kw * dt);
  ImuPose y = pose;
  y.t_wp.translation() = y.t_wp.translation() + kv * dt;
  const Quat q = qmul(r_v2_v1.unit_quaternion(), pose.t_wp.so3().unit_quaternion());
  y.t_wp.so3() = SO3::raw(q);
  y.v_w = y.v_w + ka * dt;
  if (pdy_dk) {
    *pdy_dk = Mat<10, 9>::Zero();
    pdy_dk->setBlock<3, 3>(0, 0, Mat3::Identity() * dt);
    const Mat<4, 3> dq = dq1q2_dq1(pose.t_wp.so3().unit_quaternion()) *
                         dq_exp_dw(kw * dt) * dt;
    pdy_dk->setBlock<4, 3>(3, 3, dq);
    pdy_dk->setBlock<3, 3>(7, 6, Mat3::Identity() * dt);
  }
  if (pdy_dy) *pdy_dy = dq1q2_dq2(r_v2_v1.unit_quaternion());
  return y;
}

// Types.h:376-416  GetPoseDerivative
inline Mat<9, 1> GetPoseDerivative(const ImuPose& pose, const Vec3& g_w,
                                   const ImuMeasurement& z_start,
                                   const ImuMeasurement& z_end, const Vec3& bg,
                                   const Vec3& ba, double dt,
                                   Mat<9, 6>* dk_db = nullptr,
                                   Mat<9, 10>* dk_dx = nullptr) {
  const double alpha =
      (z_end.time - (z_start.time + dt)) / (z_end.time - z_start.time);
  const Vec3 zg = z_start.w * alpha + z_end.w * (1.0 - alpha);
  const Vec3 za = z_start.a * alpha + z_end.a * (1.0 - alpha);
  Mat<9, 1> deriv;
  const Vec3 wv = pose.t_wp.so3().Adj() * (zg + bg);
  const Vec3 av = pose.t_wp.so3() * (za + ba) - g_w;
  for (int i = 0; i < 3; ++i) {
    deriv[i] = pose.v_w[i];
    deriv[3 + i] = wv[i];
    deriv[6 + i] = av[i];
  }
  if (dk_db) {
    *dk_db = Mat<9, 6>::Zero();
    dk_db->setBlock<3, 3>(3, 0, pose.t_wp.so3().Adj());
    dk_db->setBlock<3, 3>(6, 3, pose.t_wp.so3().matrix());
  }
  if (dk_dx) {
    *dk_dx = Mat<9, 10>::Zero();
    dk_dx->setBlock<3, 3>(0, 7, Mat3::Identity());
    const Quat& q = pose.t_wp.so3().unit_quaternion();
    dk_dx->setBlock<3, 4>(3, 3, dqx_dq(q, zg) + dqx_dq(q, bg));
    dk_dx->setBlock<3, 4>(6, 3, dqx_dq(q, za) + dqx_dq(q, ba));
  }
  return deriv;
}

inline void add_identity_blocks(Mat<10, 10>& m, const Mat4& dy_dy) {
  // Types.h:488-490 (and the three repeats): +I on t and v, + dq/dq on the
  // quaternion block.
  m.addBlock<3, 3>(0, 0, Mat3::Identity());
  m.addBlock<3, 3>(7, 7, Mat3::Identity());
  m.addBlock<4, 4>(3, 3, dy_dy);
}

// Types.h:419-643  IntegrateImu (RK4).  Jacobian branch when dy_db, dy_dpose and r
// are all given; covariance in Euler form (euler_covariance defaults to true and
// is never overridden on the hot path, Types.h:429,601-606).
inline ImuPose IntegrateImu(const ImuPose& pose, const ImuMeasurement& z_start,
                            const ImuMeasurement& z_end, const Vec3& bg,
                            const Vec3& ba, const Vec3& g,
                            Mat<10, 6>* dy_db_ptr = nullptr,
                            Mat<10, 10>* dy_dpose_ptr = nullptr,
                            Mat<10, 10>* c_prior = nullptr,
                            const Vec6* r = nullptr) {
  const double dt = z_end.time - z_start.time;
  if (dt == 0) return pose;
  ImuPose res = pose;
  Mat<9, 1> k;
  if (dy_db_ptr && dy_dpose_ptr && r) {
    Mat<10, 6>& dy_db = *dy_db_ptr;
    Mat<10, 10>& dy_dy0 = *dy_dpose_ptr;
    Mat<9, 6> dk_db;
    Mat<9, 10> dk_dy;
    Mat<10, 9> dy_dk;
    Mat4 dy_dy;
    dy_db = Mat<10, 6>::Zero();
    dy_dy0 = Mat<10, 10>::Identity();

    const Mat<9, 1> k1 =
        GetPoseDerivative(pose, g, z_start, z_end, bg, ba, 0, &dk_db, &dk_dy);
    const Mat<9, 6> dk1_db = dk_db;
    const Mat<9, 10> dk1_dy = dk_dy;
    const ImuPose y1 = IntegratePose(pose, k1, dt * 0.5, &dy_dk, &dy_dy);
    dy_db = dy_dk * dk1_db;
    dy_dy0 = dy_dk * dk1_dy;
    add_identity_blocks(dy_dy0, dy_dy);

    const Mat<9, 1> k2 =
        GetPoseDerivative(y1, g, z_start, z_end, bg, ba, dt / 2, &dk_db, &dk_dy);
    const Mat<9, 6> dk2_db = dk_db + dk_dy * dy_db;
    const Mat<9, 10> dk2_dy = dk_dy * dy_dy0;
    const ImuPose y2 = IntegratePose(pose, k2, dt * 0.5, &dy_dk, &dy_dy);
    dy_db = dy_dk * dk2_db;
    dy_dy0 = dy_dk * dk2_dy;
    add_identity_blocks(dy_dy0, dy_dy);

    const Mat<9, 1> k3 =
        GetPoseDerivative(y2, g, z_start, z_end, bg, ba, dt / 2, &dk_db, &dk_dy);
    const Mat<9, 6> dk3_db = dk_db + dk_dy * dy_db;
    const Mat<9, 10> dk3_dy = dk_dy * dy_dy0;
    const ImuPose y3 = IntegratePose(pose, k3, dt, &dy_dk, &dy_dy);
    dy_db = dy_dk * dk3_db;
    dy_dy0 = dy_dk * dk3_dy;
    add_identity_blocks(dy_dy0, dy_dy);

    const Mat<9, 1> k4 =
        GetPoseDerivative(y3, g, z_start, z_end, bg, ba, dt, &dk_db, &dk_dy);
    const Mat<9, 6> dk4_db = dk_db + dk_dy * dy_db;
    const Mat<9, 10> dk4_dy = dk_dy * dy_dy0;

    k = k1 + 2.0 * k2 + 2.0 * k3 + k4;
    const Mat<9, 6> dk_total_db = dk1_db + 2.0 * dk2_db + 2.0 * dk3_db + dk4_db;
    const Mat<9, 10> dk_total_dy = dk1_dy + 2.0 * dk2_dy + 2.0 * dk3_dy + dk4_dy;

    res = IntegratePose(pose, k, dt / 6.0, &dy_dk, &dy_dy);
    dy_db = dy_dk * dk_total_db;
    dy_dy0 = dy_dk * dk_total_dy;
    add_identity_blocks(dy_dy0, dy_dy);

    if (c_prior) {
      // Types.h:601-606  Euler covariance: C <- F C F^T + G R G^T
      Mat<6, 6> R;
      for (int i = 0; i < 6; ++i) R(i, i) = (*r)[i];
      const Mat<10, 10> c_prop = dy_dy0 * (*c_prior) * dy_dy0.T();
      *c_prior = c_prop + dy_db * R * dy_db.T();
    }
  } else {
    const Mat<9, 1> k1 = GetPoseDerivative(pose, g, z_start, z_end, bg, ba, 0);
    const ImuPose y1 = IntegratePose(pose, k1, dt * 0.5);
    const Mat<9, 1> k2 = GetPoseDerivative(y1, g, z_start, z_end, bg, ba, dt / 2);
    const ImuPose y2 = IntegratePose(pose, k2, dt * 0.5);
    const Mat<9, 1> k3 = GetPoseDerivative(y2, g, z_start, z_end, bg, ba, dt / 2);
    const ImuPose y3 = IntegratePose(pose, k3, dt);
    const Mat<9, 1> k4 = GetPoseDerivative(y3, g, z_start, z_end, bg, ba, dt);
    k = k1 + 2.0 * k2 + 2.0 * k3 + k4;
    res = IntegratePose(pose, k, dt / 6.0);
  }
  for (int i = 0; i < 3; ++i) res.w_w[i] = k[3 + i];
  res.time = z_end.time;
  return res;
}

// Types.h:662-738  IntegrateResidual: chain IntegrateImu over the sample list,
// pushing the bias Jacobian forward: dpose_db <- dy_db + dy_dy dpose_db.
inline ImuPose IntegrateResidual(ImuPose pose,
                                 const std::vector<ImuMeasurement>& measurements,
                                 const Vec3& bg, const Vec3& ba, const Vec3& g,
                                 Mat<10, 6>* dpose_db = nullptr,
                                 Mat<10, 10>* dpose_dpose = nullptr,
                                 Mat<10, 10>* c_res = nullptr,
                                 const Vec6* r = nullptr) {
  const ImuMeasurement* prev = nullptr;
  if (dpose_db) *dpose_db = Mat<10, 6>::Zero();
  if (dpose_dpose) *dpose_dpose = Mat<10, 10>::Identity();
  Mat<10, 6> dy_db;
  Mat<10, 10> dy_dy;
  for (const ImuMeasurement& meas : measurements) {
    if (prev) {
      if ((dpose_db || dpose_dpose) && r) {
        pose = IntegrateImu(pose, *prev, meas, bg, ba, g, &dy_db, &dy_dy, c_res, r);
        if (dpose_db) *dpose_db = dy_db + dy_dy * (*dpose_db);
        if (dpose_dpose) *dpose_dpose = dy_dy * (*dpose_dpose);
      } else {
        pose = IntegrateImu(pose, *prev, meas, bg, ba, g);
      }
    }
    prev = &meas;
  }
  return pose;
}

}  // namespace orc
