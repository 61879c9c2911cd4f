// Minimal C++ application against include/ba/BundleAdjuster.h — the way the reference's
// applications use the class (cf. /root/reference/applications/unary_binary_imu_test/
// main.cpp:31,60-230: Init / AddPose / Add*Constraint / AddImuResidual / Solve / GetPose).
// Builds a small ring of cameras looking at random points, perturbs the state and runs
// Solve() on the MI355X engine.  Exit code 0 iff the reprojection error went down.
//
//   visual_ba_demo                          ba::BundleAdjuster<double, 1, 6, 0>        (VisualBundleAdjuster)
//   visual_ba_demo --calibrate-intrinsics   ba::BundleAdjuster<double, 1, 6, 4, false> starting from pinhole
//                                           parameters that are 2-3 % off; a third of the poses held fixed
//   visual_ba_demo --calibrate-extrinsics   ba::BundleAdjuster<double, 1, 6, 0, true>  starting from a camera
//                                           mount T_vs that is a few centimetres / half a degree off
//   visual_ba_demo --calibrate-fov          ba::SelfCalBundleAdjuster<double> = <double, 1, 6, 5> (reference
//                                           BundleAdjuster.h:758-759) on a ba::FovCamera whose five parameters
//                                           (fx, fy, u0, v0, w) start 2-4 % off
//   visual_ba_demo --ranks N --rank R --comm-id-file F [--device D]
//                                           one process per GPU: every process holds all poses and the landmarks
//                                           l with l mod N == R; the class joins the engine-owned RCCL communicator
//                                           (SetCommunicator; rank 0 writes the 128-byte id to F, the others wait for
//                                           it) and the reduced solve is distributed over the GPUs — no torch, no hooks
#include <ba/BundleAdjuster.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>

struct Shard { int rank = 0, ranks = 1, device = 0; std::string id_file; };

// rank 0 creates the communicator id and publishes it through a file; the other ranks wait for it
static bool exchange_id(const Shard& sh, unsigned char* id) {
  if (sh.rank == 0) {
    if (!ba::BundleAdjuster<double, 1, 6, 0>::CreateCommunicatorId(id)) return false;
    const std::string tmp = sh.id_file + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, 128, f) != 128) return false;
    std::fclose(f);
    return std::rename(tmp.c_str(), sh.id_file.c_str()) == 0;
  }
  for (int tries = 0; tries < 600; ++tries) {
    if (FILE* f = std::fopen(sh.id_file.c_str(), "rb")) {
      const size_t n = std::fread(id, 1, 128, f);
      std::fclose(f);
      if (n == 128) return true;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  return false;
}

template <class BA>
int run(int calibrate, const Shard& shard = Shard()) {  // 0 none, 1 intrinsics, 2 extrinsics, 3 the five parameters of a FOV camera
  BA adjuster;
  ba::Options<double> options;  // reference defaults: dogleg, robust norm, auto regularisation
  options.error_change_threshold = 1e-5;
  options.device = shard.device;
  if (!shard.id_file.empty()) {
    unsigned char id[128];
    if (!exchange_id(shard, id)) { std::printf("communicator id exchange failed\n"); return 3; }
    adjuster.SetCommunicator(id, shard.rank, shard.ranks);
  }
  const int kPoses = 24, kLandmarks = 300;
  adjuster.Init(options, kPoses, kLandmarks * 6, kLandmarks);
  const double fx = 198.969, fy = 198.1284, u0 = 329.9368, v0 = 240.1017, fov_w = 0.93;
  ba::SE3 mount0;  // the true mount is the identity
  if (calibrate == 2) {
    const double t[3] = {0.03, -0.02, 0.02}, q[4] = {0.004, -0.005, 0.003, 1.0};
    mount0 = ba::SE3(t, q);
  }
  const ba::FovCamera<double> true_fov(fx, fy, u0, v0, fov_w);
  if (calibrate == 3)
    adjuster.AddCamera(std::make_shared<ba::FovCamera<double>>(fx * 1.03, fy * 0.97, u0 * 1.02, v0 * 0.98, fov_w * 1.04, mount0));
  else if (calibrate == 1)
    adjuster.AddCamera(std::make_shared<ba::CameraInterface<double>>(fx * 1.03, fy * 0.97, u0 * 1.02, v0 * 0.98, mount0));
  else
    adjuster.AddCamera(std::make_shared<ba::CameraInterface<double>>(fx, fy, u0, v0, mount0));

  std::mt19937 rng(7);
  std::normal_distribution<double> n01(0.0, 1.0);
  std::uniform_real_distribution<double> u01(0.0, 1.0);
  // cameras on a circle of radius 6 looking at the origin
  std::vector<ba::SE3> gt(kPoses);
  for (int i = 0; i < kPoses; ++i) {
    const double a = 2 * M_PI * i / kPoses;
    const double c[3] = {6 * std::cos(a), 6 * std::sin(a), 0.3 * std::sin(3 * a)};
    // z axis -> a target near the origin that wanders with the pose (cameras that all look at ONE
    // point make a shift of the mount along the optical axis a pure change of scale), y axis ~ world -z
    const double tgt[3] = {0.8 * std::sin(2 * a), 0.8 * std::cos(3 * a), 0.2 * std::sin(a)};
    double z[3] = {tgt[0] - c[0], tgt[1] - c[1], tgt[2] - c[2]};
    const double zn = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    for (double& v : z) v /= zn;
    double x[3] = {z[1], -z[0], 0.0};  // z cross (0,0,1)... any unit vector orthogonal to z
    const double xn = std::sqrt(x[0] * x[0] + x[1] * x[1]);
    for (double& v : x) v /= xn;
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    {  // roll about the optical axis
      const double r = 0.5 * std::sin(5 * a), cr = std::cos(r), sr = std::sin(r);
      for (int k = 0; k < 3; ++k) {
        const double xk = cr * x[k] + sr * y[k], yk = -sr * x[k] + cr * y[k];
        x[k] = xk; y[k] = yk;
      }
    }
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
    if (calibrate == 3) {  // the scene is seen through the FOV camera
      const ba::Vector2t d = true_fov.Project(ba::Vector3t({p[0], p[1], p[2]}));
      uv[0] = d[0]; uv[1] = d[1];
    }
    return p[2] > 0.5 && uv[0] > 0 && uv[0] < 640 && uv[1] > 0 && uv[1] < 480;
  };
  for (int i = 0; i < kPoses; ++i) {
    ba::SE3 init = gt[i];
    for (int k = 0; k < 3; ++k) init.t[k] += 0.03 * n01(rng);
    // two fixed poses pin the gauge (and the scale); self-calibration needs more of them
    const bool fixed = calibrate ? (i % 3 == 0) : (i < 2);
    adjuster.AddPose(fixed ? gt[i] : init, /*is_active=*/!fixed);
  }
  int n_res = 0;
  for (int l = 0; l < kLandmarks; ++l) {
    const double X[3] = {2.5 * (u01(rng) - 0.5), 2.5 * (u01(rng) - 0.5), 1.5 * (u01(rng) - 0.5)};
    int ref = -1;
    double uv[2];
    for (int i = 0; i < kPoses && ref < 0; ++i)
      if (project(gt[(i + l) % kPoses], X, uv)) ref = (i + l) % kPoses;
    if (ref < 0) continue;
    double Xp[4] = {X[0] * (1 + 0.02 * n01(rng)), X[1] * (1 + 0.02 * n01(rng)), X[2], 1.0};
    if (calibrate == 2) {
      // a front end hands over world points built with ITS mount guess: keep the point's coordinates in
      // the reference camera and map them back through T_wp T_vs(guess)
      auto apply = [](const ba::SE3& T, const double* p, double* o) {
        const ba::Matrix3t R = T.rotationMatrix();
        for (int r = 0; r < 3; ++r) o[r] = R(r, 0) * p[0] + R(r, 1) * p[1] + R(r, 2) * p[2] + T.t[r];
      };
      double xs[3], xw[3];
      apply(gt[ref].inverse(), Xp, xs);
      apply(gt[ref] * mount0, xs, xw);
      for (int k = 0; k < 3; ++k) Xp[k] = xw[k];
    }
    // landmark shards: every rank draws the same scene, a rank keeps the landmarks l with l mod ranks == rank
    const bool mine = (l % shard.ranks) == shard.rank;
    const uint32_t lm = mine ? adjuster.AddLandmark(ba::Vector4t({Xp[0], Xp[1], Xp[2], Xp[3]}), ref, 0, true) : 0;
    for (int i = 0; i < kPoses; ++i) {
      if (!project(gt[i], X, uv)) continue;
      const ba::Vector2t z({uv[0] + 0.5 * n01(rng), uv[1] + 0.5 * n01(rng)});
      if (mine && adjuster.AddProjectionResidual(z, i, lm, 0) != (uint32_t)-1) ++n_res;
    }
  }
  std::printf("poses %u landmarks %u residuals %d\n", adjuster.GetNumPoses(), adjuster.GetNumLandmarks(), n_res);
  adjuster.Solve(1);
  if (!shard.id_file.empty())
    std::printf("rank %d of %d on device %d: reduced solve %s\n", shard.rank, shard.ranks, shard.device,
                adjuster.SolveIsDistributed() ? "distributed over the communicator" : "replicated");
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
  bool ok = e1 <= e0 && worst < 0.15;
  if (calibrate == 1) {
    const ba::Vector4t p = adjuster.rig()->cameras_[0]->GetParams();
    std::printf("camera parameters %.3f %.3f %.3f %.3f (true %.3f %.3f %.3f %.3f)\n", p[0], p[1], p[2], p[3], fx, fy, u0, v0);
    ok = ok && std::fabs(p[0] - fx) < 0.01 * fx && std::fabs(p[1] - fy) < 0.01 * fy && std::fabs(p[2] - u0) < 0.01 * u0 &&
         std::fabs(p[3] - v0) < 0.01 * v0;
  }
  if (calibrate == 3) {
    const std::vector<double> p = adjuster.rig()->cameras_[0]->ParamsVector();
    std::printf("camera parameters %.3f %.3f %.3f %.3f %.4f (true %.3f %.3f %.3f %.3f %.4f)\n", p[0], p[1], p[2], p[3], p[4],
                fx, fy, u0, v0, fov_w);
    ok = ok && p.size() == 5 && std::fabs(p[0] - fx) < 0.01 * fx && std::fabs(p[1] - fy) < 0.01 * fy &&
         std::fabs(p[2] - u0) < 0.01 * u0 && std::fabs(p[3] - v0) < 0.01 * v0 && std::fabs(p[4] - fov_w) < 0.01 * fov_w;
  }
  if (calibrate == 2) {
    const ba::SE3 m = adjuster.rig()->cameras_[0]->Pose();
    std::printf("camera mount t = %.4f %.4f %.4f  q = %.5f %.5f %.5f %.5f (true: identity)\n", m.t[0], m.t[1], m.t[2], m.q[0],
                m.q[1], m.q[2], m.q[3]);
    const double before = std::sqrt(0.03 * 0.03 + 0.02 * 0.02 + 0.02 * 0.02);
    ok = ok && std::sqrt(m.t[0] * m.t[0] + m.t[1] * m.t[1] + m.t[2] * m.t[2]) < before;
  }
  return ok ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::strcmp(argv[1], "--calibrate-intrinsics") == 0) return run<ba::BundleAdjuster<double, 1, 6, 4, false>>(1);
  if (argc > 1 && std::strcmp(argv[1], "--calibrate-extrinsics") == 0) return run<ba::BundleAdjuster<double, 1, 6, 0, true>>(2);
  if (argc > 1 && std::strcmp(argv[1], "--calibrate-fov") == 0) return run<ba::SelfCalBundleAdjuster<double>>(3);
  Shard shard;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (std::strcmp(argv[i], "--ranks") == 0) shard.ranks = std::atoi(argv[i + 1]);
    else if (std::strcmp(argv[i], "--rank") == 0) shard.rank = std::atoi(argv[i + 1]);
    else if (std::strcmp(argv[i], "--device") == 0) shard.device = std::atoi(argv[i + 1]);
    else if (std::strcmp(argv[i], "--comm-id-file") == 0) shard.id_file = argv[i + 1];
  }
  if (shard.ranks < 1 || shard.rank < 0 || shard.rank >= shard.ranks) { std::printf("bad --rank / --ranks\n"); return 3; }
  return run<ba::BundleAdjuster<double, 1, 6, 0>>(0, shard);  // VisualBundleAdjuster<double>
}
