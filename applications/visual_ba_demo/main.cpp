// Minimal C++ application against include/ba/BundleAdjuster.h — the way the reference's
// applications use the class (cf. /root/reference/applications/unary_binary_imu_test/
// main.cpp:31,60-230: Init / AddPose / Add*Constraint / AddImuResidual / Solve / GetPose).
// Builds a small ring of cameras looking at random points, perturbs the state and runs
// Solve() on the MI355X engine.  Exit code 0 iff the reprojection error went down.
#include <ba/BundleAdjuster.h>

#include <cmath>
#include <cstdio>
#include <random>

int main() {
  typedef ba::BundleAdjuster<double, 1, 6, 0> BA;  // VisualBundleAdjuster<double>
  BA adjuster;
  ba::Options<double> options;  // reference defaults: dogleg, robust norm, auto regularisation
  options.error_change_threshold = 1e-5;
  const int kPoses = 24, kLandmarks = 300;
  adjuster.Init(options, kPoses, kLandmarks * 6, kLandmarks);
  const double fx = 198.969, fy = 198.1284, u0 = 329.9368, v0 = 240.1017;
  adjuster.AddCamera(std::make_shared<ba::CameraInterface<double>>(fx, fy, u0, v0));

  std::mt19937 rng(7);
  std::normal_distribution<double> n01(0.0, 1.0);
  std::uniform_real_distribution<double> u01(0.0, 1.0);
  // cameras on a circle of radius 6 looking at the origin
  std::vector<ba::SE3> gt(kPoses);
  for (int i = 0; i < kPoses; ++i) {
    const double a = 2 * M_PI * i / kPoses;
    const double c[3] = {6 * std::cos(a), 6 * std::sin(a), 0.3 * std::sin(3 * a)};
    // z axis -> origin, y axis ~ world -z
    double z[3] = {-c[0], -c[1], -c[2]};
    const double zn = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    for (double& v : z) v /= zn;
    double x[3] = {z[1], -z[0], 0.0};  // z cross (0,0,1)... any unit vector orthogonal to z
    const double xn = std::sqrt(x[0] * x[0] + x[1] * x[1]);
    for (double& v : x) v /= xn;
    const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    const double R[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
    // rotation matrix -> quaternion (trace branch is fine for this geometry after a sign fix)
    double q[4];
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      const double s = std::sqrt(tr + 1.0) * 2;
      q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
      const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
      q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
      const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
      q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
    } else {
      const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
      q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
    }
    gt[i] = ba::SE3(c, q);
  }
  auto project = [&](const ba::SE3& T, const double* X, double* uv) {
    const ba::Matrix3t R = T.rotationMatrix();
    double d[3] = {X[0] - T.t[0], X[1] - T.t[1], X[2] - T.t[2]}, p[3];
    for (int r = 0; r < 3; ++r) p[r] = R(0, r) * d[0] + R(1, r) * d[1] + R(2, r) * d[2];  // R^T d
    uv[0] = fx * p[0] / p[2] + u0; uv[1] = fy * p[1] / p[2] + v0;
    return p[2] > 0.5 && uv[0] > 0 && uv[0] < 640 && uv[1] > 0 && uv[1] < 480;
  };
  for (int i = 0; i < kPoses; ++i) {
    ba::SE3 init = gt[i];
    if (i >= 2) for (int k = 0; k < 3; ++k) init.t[k] += 0.03 * n01(rng);
    adjuster.AddPose(init, /*is_active=*/i >= 2);  // two fixed poses pin the gauge (and the scale)
  }
  int n_res = 0;
  for (int l = 0; l < kLandmarks; ++l) {
    const double X[3] = {2.5 * (u01(rng) - 0.5), 2.5 * (u01(rng) - 0.5), 1.5 * (u01(rng) - 0.5)};
    int ref = -1;
    double uv[2];
    for (int i = 0; i < kPoses && ref < 0; ++i)
      if (project(gt[(i + l) % kPoses], X, uv)) ref = (i + l) % kPoses;
    if (ref < 0) continue;
    const double Xp[4] = {X[0] * (1 + 0.02 * n01(rng)), X[1] * (1 + 0.02 * n01(rng)), X[2], 1.0};
    const uint32_t lm = adjuster.AddLandmark(ba::Vector4t({Xp[0], Xp[1], Xp[2], Xp[3]}), ref, 0, true);
    for (int i = 0; i < kPoses; ++i) {
      if (!project(gt[i], X, uv)) continue;
      const ba::Vector2t z({uv[0] + 0.5 * n01(rng), uv[1] + 0.5 * n01(rng)});
      if (adjuster.AddProjectionResidual(z, i, lm, 0) != (uint32_t)-1) ++n_res;
    }
  }
  std::printf("poses %u landmarks %u residuals %d\n", adjuster.GetNumPoses(), adjuster.GetNumLandmarks(), n_res);
  adjuster.Solve(1);
  double e0, eu, eb, ei;
  adjuster.GetErrors(e0, eu, eb, ei);
  if (!adjuster.GetSolutionSummary().IsResultGood()) { std::printf("Solve failed\n"); return 2; }
  adjuster.Solve(8);
  double e1;
  adjuster.GetErrors(e1, eu, eb, ei);
  double worst = 0;
  for (int i = 0; i < kPoses; ++i) {
    const auto& p = adjuster.GetPose(i);
    for (int k = 0; k < 3; ++k) worst = std::fmax(worst, std::fabs(p.t_wp.t[k] - gt[i].t[k]));
  }
  std::printf("proj error after 1 iteration %.4f, after 9 %.4f, result %d, worst position error %.4f m\n", e0, e1,
              (int)adjuster.GetSolutionSummary().result, worst);
  return (e1 <= e0 && worst < 0.15) ? 0 : 1;
}
