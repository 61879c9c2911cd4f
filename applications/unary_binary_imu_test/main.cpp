// Pose-graph application against include/ba/BundleAdjuster.h with the shape of the reference's
// applications/unary_binary_imu_test (main.cpp:31 `ba::BundleAdjuster<double,0,9,0> slam`,
// :60-230): wheel odometry -> binary constraints between consecutive nodes, position fixes
// -> unary constraints (use_rotation = false), an IMU stream held in a
// ba::InterpolationBufferT and cut per node interval with GetRange -> AddImuResidual, then
// Solve (Gauss-Newton, no dogleg) and GetPose.  The reference reads an elided `log.dat`
// ("ODO t rr rl", "UTM t e n alt", "IMU t wx wy wz ax ay az"); this program synthesises the
// same three streams from a known planar trajectory so that the result can be checked:
// exit code 0 iff the mean node position error drops below a third of its initial value.
#include <ba/BundleAdjuster.h>
#include <ba/InterpolationBuffer.h>
#include <ba/Types.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef ba::ImuMeasurementT<double> ImuMeasurement;

static ba::SE3 planar(double x, double y, double yaw) {
  const double t[3] = {x, y, 0.0}, q[4] = {0.0, 0.0, std::sin(0.5 * yaw), std::cos(0.5 * yaw)};
  return ba::SE3(t, q);
}
// T_1^-1 T_2 for planar poses
static ba::SE3 between(double x1, double y1, double a1, double x2, double y2, double a2) {
  const double dx = x2 - x1, dy = y2 - y1, c = std::cos(a1), s = std::sin(a1);
  return planar(c * dx + s * dy, -s * dx + c * dy, a2 - a1);
}

int main() {
  ba::BundleAdjuster<double, 0, 9, 0> slam;  // pose graph: no landmarks, pose + velocity
  ba::Options<double> options;
  options.use_dogleg = false;
  options.error_change_threshold = 1e-9;
  options.param_change_threshold = 1e-12;
  const int kNodes = 80;
  const double kNodeDt = 0.5, kImuDt = 0.01, kDuration = kNodes * kNodeDt;
  slam.Init(options, kNodes, 0, 0);
  const double g[3] = {0.0, 0.0, -9.8007};  // Types.h:39
  slam.SetGravity(ba::Vector3t({g[0], g[1], g[2]}));

  // ground truth: a figure "8"-like planar drive, analytic position / heading
  const double R1 = 12.0, w1 = 2 * M_PI / 35.0;
  auto pos = [&](double t, double* p) {
    p[0] = R1 * std::sin(w1 * t);
    p[1] = 0.5 * R1 * std::sin(2 * w1 * t);
  };
  auto vel = [&](double t, double* v) {
    v[0] = R1 * w1 * std::cos(w1 * t);
    v[1] = R1 * w1 * std::cos(2 * w1 * t);
  };
  auto acc = [&](double t, double* a) {
    a[0] = -R1 * w1 * w1 * std::sin(w1 * t);
    a[1] = -2 * R1 * w1 * w1 * std::sin(2 * w1 * t);
  };
  auto yaw = [&](double t) { double v[2]; vel(t, v); return std::atan2(v[1], v[0]); };
  auto yaw_rate = [&](double t) {
    double v[2], a[2]; vel(t, v); acc(t, a);
    return (v[0] * a[1] - v[1] * a[0]) / (v[0] * v[0] + v[1] * v[1]);
  };

  std::mt19937 rng(11);
  std::normal_distribution<double> n01(0.0, 1.0);

  // ---- IMU stream into the interpolation buffer (body x forward, z up) --------------------
  ba::InterpolationBufferT<ImuMeasurement, double> imu_buffer;
  for (int k = 0; k * kImuDt <= kDuration + 1e-9; ++k) {
    const double t = k * kImuDt, th = yaw(t);
    double a[2]; acc(t, a);
    // specific force: R^T (a_world + |g| e_z) under v' = R a_m - g_vec with g_vec = (0,0,+9.8)
    // in the engine's convention (see ba_amd/scene.py:add_inertial); planar: rotate by -yaw
    const double fw[3] = {a[0] + g[0], a[1] + g[1], 0.0 + g[2]};
    const double c = std::cos(th), s = std::sin(th);
    const ba::Vector3t am({c * fw[0] + s * fw[1] + 1e-3 * n01(rng), -s * fw[0] + c * fw[1] + 1e-3 * n01(rng),
                           fw[2] + 1e-3 * n01(rng)});
    const ba::Vector3t wm({5e-5 * n01(rng), 5e-5 * n01(rng), yaw_rate(t) + 5e-5 * n01(rng)});
    imu_buffer.AddElement(ImuMeasurement(wm, am, t));
  }

  // ---- nodes: dead-reckoned initial poses from noisy odometry ------------------------------
  std::vector<double> gx(kNodes), gy(kNodes), ga(kNodes);
  for (int i = 0; i < kNodes; ++i) { double p[2]; pos(i * kNodeDt, p); gx[i] = p[0]; gy[i] = p[1]; ga[i] = yaw(i * kNodeDt); }
  std::vector<ba::SE3> odo(kNodes - 1);
  for (int i = 0; i + 1 < kNodes; ++i) {
    ba::SE3 d = between(gx[i], gy[i], ga[i], gx[i + 1], gy[i + 1], ga[i + 1]);
    d.t[0] += 0.03 * n01(rng); d.t[1] += 0.03 * n01(rng);
    const double da = 0.01 * n01(rng);
    const double q2[4] = {0, 0, std::sin(0.5 * da), std::cos(0.5 * da)};
    const double z = d.q[2], w = d.q[3];
    d.q[2] = z * q2[3] + w * q2[2]; d.q[3] = w * q2[3] - z * q2[2];
    odo[i] = d;
  }
  double ex = gx[0], ey = gy[0], ea = ga[0];
  std::vector<double> ix(kNodes), iy(kNodes);
  for (int i = 0; i < kNodes; ++i) {
    ix[i] = ex; iy[i] = ey;
    double v[2]; vel(i * kNodeDt, v);
    const ba::Vector3t v_w({v[0] + 0.05 * n01(rng), v[1] + 0.05 * n01(rng), 0.0});
    slam.AddPose(planar(ex, ey, ea), std::vector<double>(), v_w, ba::Vector6t::Zero(), true, i * kNodeDt);
    if (i + 1 < kNodes) {
      const double c = std::cos(ea), s = std::sin(ea);
      ex += c * odo[i].t[0] - s * odo[i].t[1];
      ey += s * odo[i].t[0] + c * odo[i].t[1];
      ea += 2 * std::atan2(odo[i].q[2], odo[i].q[3]);
    }
  }
  // ---- constraints ---------------------------------------------------------------------------
  ba::Matrix6t cov_odo = ba::Matrix6t::Identity();
  for (int k = 0; k < 3; ++k) { cov_odo(k, k) = 0.03 * 0.03; cov_odo(3 + k, 3 + k) = 0.01 * 0.01; }
  ba::Matrix6t cov_fix = ba::Matrix6t::Identity();
  for (int k = 0; k < 3; ++k) { cov_fix(k, k) = 0.1 * 0.1; cov_fix(3 + k, 3 + k) = 1.0; }
  for (int i = 0; i + 1 < kNodes; ++i) slam.AddBinaryConstraint(i, i + 1, odo[i], cov_odo);
  for (int i = 0; i < kNodes; i += 5)
    slam.AddUnaryConstraint(i, planar(gx[i] + 0.1 * n01(rng), gy[i] + 0.1 * n01(rng), ga[i]), cov_fix,
                            /*use_rotation=*/false);
  for (int i = 0; i + 1 < kNodes; ++i) {
    const std::vector<ImuMeasurement> meas = imu_buffer.GetRange(i * kNodeDt, (i + 1) * kNodeDt);
    if (meas.size() < 2) { std::fprintf(stderr, "empty IMU range\n"); return 2; }
    slam.AddImuResidual(i, i + 1, meas);
  }
  auto mean_err = [&](bool initial) {
    double s = 0;
    for (int i = 0; i < kNodes; ++i) {
      const double x = initial ? ix[i] : slam.GetPose(i).t_wp.t[0];
      const double y = initial ? iy[i] : slam.GetPose(i).t_wp.t[1];
      s += std::hypot(x - gx[i], y - gy[i]);
    }
    return s / kNodes;
  };
  const double e0 = mean_err(true);
  slam.Solve(15);
  const double e1 = mean_err(false);
  const ba::SolutionSummary<double>& sum = slam.GetSolutionSummary();
  std::printf("nodes %d  imu samples %zu  mean position error %.3f m -> %.3f m  (result %d, inertial error %.4g, |delta| %.3g)\n",
              kNodes, imu_buffer.elements.size(), e0, e1, (int)sum.result, sum.inertial_error, sum.delta_norm);
  if (!sum.IsResultGood()) { std::fprintf(stderr, "solver failed (result %d)\n", (int)sum.result); return 3; }
  return e1 < e0 / 3.0 ? 0 : 1;
}
