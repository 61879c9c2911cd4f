// ImuResidualT::IntegrateResidual / IntegrateImu of include/ba/Types.h (reference Types.h:419-738,
// state only) through the BundleAdjuster typedefs: constant-acceleration motion against the
// closed form, the `poses` trajectory, and IntegrateImu == a two-sample IntegrateResidual.
#include <cmath>
#include <cstdio>
#include <ba/BundleAdjuster.h>

typedef ba::BundleAdjuster<double, 1, 15, 0> BA;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

int main() {
  const ba::Vector3t g({0.0, 0.0, -9.8007}), zero({0.0, 0.0, 0.0});
  // no rotation; the reference's model is v' = R (a + b_a) - g (GetPoseDerivative, Types.h:376-400): a constant
  // world acceleration of (1, 0, 0) needs the reading (1, 0, 0) + g
  std::vector<BA::ImuMeasurement> meas;
  for (int i = 0; i <= 20; ++i)
    meas.push_back(BA::ImuMeasurement(zero, ba::Vector3t({1.0, 0.0, -9.8007}), 0.05 * i));
  BA::ImuPose start(ba::SE3(), ba::Vector3t({0.5, 0.0, 0.0}), zero, 0.0);
  std::vector<BA::ImuPose> poses;
  const BA::ImuPose end = BA::ImuResidual::IntegrateResidual(start, meas, zero, zero, g, poses);
  const double T = 1.0;
  CHECK(poses.size() == meas.size());
  CHECK(std::fabs(end.t_wp.t[0] - (0.5 * T + 0.5 * T * T)) < 1e-12);
  CHECK(std::fabs(end.t_wp.t[1]) < 1e-12 && std::fabs(end.t_wp.t[2]) < 1e-12);
  CHECK(std::fabs(end.v_w[0] - 1.5) < 1e-12 && std::fabs(end.v_w[2]) < 1e-12);
  CHECK(std::fabs(poses[10].t_wp.t[0] - (0.25 + 0.125)) < 1e-12 && poses[10].time == 0.5);
  CHECK(poses[0].t_wp.t[0] == 0.0 && poses[0].v_w[0] == 0.5);
  // one step at a time reproduces the trajectory
  BA::ImuPose p = start;
  for (size_t i = 1; i < meas.size(); ++i) p = BA::ImuResidual::IntegrateImu(p, meas[i - 1], meas[i], zero, zero, g);
  CHECK(std::fabs(p.t_wp.t[0] - end.t_wp.t[0]) < 1e-13 && std::fabs(p.v_w[0] - end.v_w[0]) < 1e-13);
  // the bias is ADDED to the reading: -1 along x cancels the acceleration
  const BA::ImuPose biased = BA::ImuResidual::IntegrateResidual(start, meas, zero, ba::Vector3t({-1.0, 0.0, 0.0}), g, poses);
  CHECK(std::fabs(biased.v_w[0] - 0.5) < 1e-12);
  // the Jacobian outputs of the reference's signatures (Types.h:419-738): bias Jacobian against a finite
  // difference of the integration itself, IntegrateImu's step Jacobians chained by hand == IntegrateResidual's
  {
    ba::Mat<10, 6> dpose_db;
    ba::Mat<10, 10> dpose_dpose, c_res;
    ba::Vector6t r;
    for (int i = 0; i < 6; ++i) r[i] = i < 3 ? 1e-4 : 1e-2;
    std::vector<BA::ImuPose> tmp;
    const BA::ImuPose e0 = BA::ImuResidual::IntegrateResidual(start, meas, zero, zero, g, tmp, &dpose_db, &dpose_dpose, &c_res, &r);
    CHECK(e0.t_wp.t[0] == end.t_wp.t[0] && e0.v_w[0] == end.v_w[0]);
    const double h = 1e-6;
    const BA::ImuPose ep = BA::ImuResidual::IntegrateResidual(start, meas, zero, ba::Vector3t({h, 0.0, 0.0}), g, tmp);
    const BA::ImuPose em = BA::ImuResidual::IntegrateResidual(start, meas, zero, ba::Vector3t({-h, 0.0, 0.0}), g, tmp);
    CHECK(std::fabs((ep.t_wp.t[0] - em.t_wp.t[0]) / (2 * h) - dpose_db(0, 3)) < 1e-6);   // d t_x / d b_a,x = T^2 / 2
    CHECK(std::fabs(dpose_db(0, 3) - 0.5 * T * T) < 1e-9 && std::fabs(dpose_db(7, 3) - T) < 1e-9);
    CHECK(std::fabs(dpose_dpose(0, 7) - T) < 1e-12 && dpose_dpose(0, 0) == 1.0);          // d t / d v0 = T
    CHECK(c_res(0, 0) > 0 && c_res(7, 7) > 0 && std::fabs(c_res(0, 7) - c_res(7, 0)) < 1e-15);
    ba::Mat<10, 6> db, dy_db;
    ba::Mat<10, 10> dy_dy;
    BA::ImuPose q = start;
    for (size_t i = 1; i < meas.size(); ++i) {
      q = BA::ImuResidual::IntegrateImu(q, meas[i - 1], meas[i], zero, zero, g, &dy_db, &dy_dy);
      ba::Mat<10, 6> nxt;
      for (int rr = 0; rr < 10; ++rr)
        for (int c = 0; c < 6; ++c) {
          double sum = 0;
          for (int k2 = 0; k2 < 10; ++k2) sum += dy_dy(rr, k2) * db(k2, c);
          nxt(rr, c) = dy_db(rr, c) + sum;
        }
      db = nxt;
    }
    double worst = 0;
    for (int rr = 0; rr < 10; ++rr)
      for (int c = 0; c < 6; ++c) worst = std::fmax(worst, std::fabs(db(rr, c) - dpose_db(rr, c)));
    CHECK(worst < 1e-12);
    // GetPoseDerivative / IntegratePose: one explicit Euler step of the model
    const ba::Mat<9, 1> k = BA::ImuResidual::GetPoseDerivative(start, g, meas[0], meas[1], zero, zero, 0.0);
    CHECK(k[0] == 0.5 && std::fabs(k[6] - 1.0) < 1e-12 && std::fabs(k[8]) < 1e-12 && k[3] == 0.0);
    const BA::ImuPose y = BA::ImuResidual::IntegratePose(start, k, 0.1);
    CHECK(std::fabs(y.t_wp.t[0] - 0.05) < 1e-15 && std::fabs(y.v_w[0] - 0.6) < 1e-15 && y.t_wp.q[3] == 1.0);
  }
  // from a PoseT
  BA::Pose pose;
  pose.v_w = ba::Vector3t({0.5, 0.0, 0.0});
  pose.time = 0.0;
  const BA::ImuPose e2 = BA::ImuResidual::IntegrateResidual(pose, meas, zero, zero, g, poses);
  CHECK(e2.t_wp.t[0] == end.t_wp.t[0]);
  // GetGravityVector / dGravity_dDirection (reference Types.h:91-110,161-180)
  {
    const ba::Vector3t g0 = ba::GetGravityVector(ba::Vector2t({0.0, 0.0}));
    CHECK(g0[0] == 0.0 && g0[1] == 0.0 && std::fabs(g0[2] + ba::Gravity) < 1e-15);
    const ba::Vector2t d({0.3, -0.2});
    const auto J = ba::dGravity_dDirection(d);
    for (int k = 0; k < 2; ++k) {
      ba::Vector2t dp = d, dm = d;
      dp[k] += 1e-6; dm[k] -= 1e-6;
      const ba::Vector3t gp = ba::GetGravityVector(dp), gm = ba::GetGravityVector(dm);
      for (int r = 0; r < 3; ++r) CHECK(std::fabs((gp[r] - gm[r]) / 2e-6 - J(r, k)) < 1e-7);
    }
    CHECK(std::fabs(ba::GetGravityVector(d).norm() - ba::Gravity) < 1e-12);
  }
  // the pinhole stand-in for calibu::LinearCamera: Project / Unproject / Transfer3d / dTransfer3d_dray
  {
    const double q[4] = {0.1, -0.2, 0.05, 0.0};
    double qq[4]; const double nn = std::sqrt(0.01 + 0.04 + 0.0025 + 0.9);
    for (int i = 0; i < 3; ++i) qq[i] = q[i] / nn; qq[3] = std::sqrt(0.9) / nn;
    const double tt[3] = {0.3, -0.1, 0.2};
    const ba::SE3 T(tt, qq);
    const ba::CameraInterface<double> cam(198.969, 198.1284, 329.9368, 240.1017);
    const ba::Vector2t pix({400.0, 200.0});
    const ba::Vector3t ray = cam.Unproject(pix);
    const ba::Vector2t back = cam.Project(ray);
    CHECK(std::fabs(back[0] - pix[0]) < 1e-12 && std::fabs(back[1] - pix[1]) < 1e-12 && ray[2] == 1.0);
    const double rho = 0.25;
    const auto J = cam.dTransfer3d_dray(T, ray, rho);
    for (int c = 0; c < 4; ++c) {
      ba::Vector3t rp = ray, rm = ray; double hp = rho, hm = rho;
      if (c < 3) { rp[c] += 1e-6; rm[c] -= 1e-6; } else { hp += 1e-6; hm -= 1e-6; }
      const ba::Vector2t a = cam.Transfer3d(T, rp, hp), b = cam.Transfer3d(T, rm, hm);
      for (int r = 0; r < 2; ++r) CHECK(std::fabs((a[r] - b[r]) / 2e-6 - J(r, c)) < 1e-5);
    }
  }
  printf(fails ? "imu integrate: %d failures\n" : "imu integrate: ok\n", fails);
  return fails ? 1 : 0;
}
