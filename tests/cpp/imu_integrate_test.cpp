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
