// GPS + IMU pose-graph application: the program of the same name in the reference
// (/root/reference/applications/unary_binary_imu_test/main.cpp) built against this repo's
// include/ba/BundleAdjuster.h — same class instantiation (`ba::BundleAdjuster<double,0,9,0>`, :31),
// same log format and parser (:243-280), same graph construction (:148-230):
//
//   ODO t rr rl            wheel speeds -> differential-drive dead reckoning (:89-141) and the
//                          speed used by the gyro dead reckoning
//   IMU t w(3) a(3)        -> ba::InterpolationBufferT (:61-68) and the incremental gyro pose (:71-86)
//   UTM t e n alt          -> a new node: AddPose at (previous node) * (gyro increment), a unary
//                          constraint at the position fix with covariance diag(1000, 1000, 30000,
//                          DBL_MAX x 3), an IMU residual over imu_buffer.GetRange(previous fix, this fix)
//   then Solve(25, 0.2) with trust_region_size = 100000 (:233-238, 284-287) and GetPose.
//
// The reference's recording (`log.dat`) is elided from its tree; `make_log.py` next to this file
// writes a synthetic one in the same format (committed as log.dat).
//
//   unary_binary_imu_test <log.dat> [--dump-graph <file>]
//
// prints one line per node (id, time, position) after the solve; --dump-graph writes the graph the
// program handed to the adjuster (poses, unary constraints, IMU residuals) so that the test can
// feed the identical graph to the oracle.  Exit code 0 iff the solver reports a good result.
#include <ba/BundleAdjuster.h>
#include <ba/InterpolationBuffer.h>
#include <ba/Types.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef ba::ImuMeasurementT<double> ImuMeasurement;

namespace {

ba::BundleAdjuster<double, 0, 9, 0> slam;  // reference :31
std::vector<unsigned int> nodes;
ba::InterpolationBufferT<ImuMeasurement, double> imu_buffer;

double offset_e = 0, offset_n = 0, offset_u = 0;  // first fix = origin (:151-157)

// differential-drive accumulators (:41-45); reset at every fix
double inc_x = 0, inc_y = 0, inc_yaw = 0, inc_timestamp = 0;
double speed = 0;                  // mean wheel speed of the last ODO record
ba::SE3 gyro_update;               // product of the gyro steps since the last fix (:51)
ba::SE3 differential_update;       // differential-drive pose since the last fix (:47)

FILE* graph_dump = nullptr;

ba::SE3 rot_z(double a, double x, double y, double z) {
  const double t[3] = {x, y, z}, q[4] = {0.0, 0.0, std::sin(0.5 * a), std::cos(0.5 * a)};
  return ba::SE3(t, q);
}
// rotation Rz(az) Ry(ay) Rx(ax) (the reference composes Eigen::AngleAxisd aaZ * aaY * aaX, :79-82)
ba::SE3 rot_zyx(double ax, double ay, double az, double tx, double ty, double tz) {
  auto axis = [](int k, double a) {
    ba::SE3 r;
    r.q[k] = std::sin(0.5 * a);
    r.q[3] = std::cos(0.5 * a);
    return r;
  };
  ba::SE3 r = axis(2, az) * axis(1, ay) * axis(0, ax);
  r.t[0] = tx; r.t[1] = ty; r.t[2] = tz;
  return r;
}

// reference :61-68
void add_imu(double timestamp, const double* rates, const double* accels) {
  const ba::Vector3t w({rates[0], rates[1], rates[2]}), a({accels[0], accels[1], accels[2]});
  imu_buffer.AddElement(ImuMeasurement(w, a, timestamp));
}

// reference :71-86 — one gyro step: rotate by the rates, advance `speed * dt` along body +y
void add_gyro_and_speed(double timestamp, double wx, double wy, double wz, double v) {
  static double last_timestamp = 0;
  if (last_timestamp != 0) {
    const double dt = timestamp - last_timestamp;
    gyro_update = gyro_update * rot_zyx(wx * dt, wy * dt, wz * dt, 0.0, v * dt, 0.0);
  }
  last_timestamp = timestamp;
}

// reference :89-141 — differential-drive odometry (track width 1.5 m)
void update_incremental_pose(double timestamp, double rr, double rl) {
  static bool first = true;
  if (first) { inc_timestamp = timestamp; first = false; return; }
  speed = 0.5 * (rr + rl);
  const double dt = timestamp - inc_timestamp, track = 1.5, tiny = 0.0001;
  if (std::fabs(rr) > tiny || std::fabs(rl) > tiny) {
    if (std::fabs(rr - rl) < tiny) {
      inc_x += std::cos(inc_yaw) * rr * dt;
      inc_y += std::sin(inc_yaw) * rr * dt;
    } else {
      const double w = (rr - rl) / track, radius = track * 0.5 * (rr + rl) / (rr - rl);
      const double cx = inc_x - radius * std::sin(inc_yaw), cy = inc_y + radius * std::cos(inc_yaw);
      const double wdt = w * dt, c = std::cos(wdt), s = std::sin(wdt);
      const double nx = c * (inc_x - cx) - s * (inc_y - cy) + cx;
      const double ny = s * (inc_x - cx) + c * (inc_y - cy) + cy;
      inc_x = nx; inc_y = ny; inc_yaw += wdt;
    }
  }
  differential_update = rot_z(inc_yaw, inc_x, inc_y, 0.0);
  inc_timestamp = timestamp;
}

// reference :148-230
bool f_gps(double timestamp, double utm_e, double utm_n, double altitude) {
  static double last_gps_timestamp = 0;
  static bool first = true;
  if (first) { offset_e = utm_e; offset_n = utm_n; offset_u = altitude; first = false; }
  const double pt[3] = {utm_e - offset_e, utm_n - offset_n, altitude - offset_u}, qi[4] = {0, 0, 0, 1};
  const ba::SE3 utm_prior(pt, qi);
  ba::SE3 estimate;  // first node: the coordinate origin (:167-171)
  if (!nodes.empty()) estimate = slam.GetPose(nodes.back()).t_wp * gyro_update;  // (:173-176)
  nodes.push_back(slam.AddPose(estimate, true, timestamp));
  ba::Matrix6t cov = ba::Matrix6t::Zero();
  const double diag[6] = {1000, 1000, 30000, DBL_MAX, DBL_MAX, DBL_MAX};  // (:183-186)
  for (int i = 0; i < 6; ++i) cov(i, i) = diag[i];
  slam.AddUnaryConstraint(nodes.back(), utm_prior, cov);
  if (graph_dump) {
    std::fprintf(graph_dump, "POSE %u %.17g", nodes.back(), timestamp);
    for (int i = 0; i < 3; ++i) std::fprintf(graph_dump, " %.17g", estimate.t[i]);
    for (int i = 0; i < 4; ++i) std::fprintf(graph_dump, " %.17g", estimate.q[i]);
    std::fprintf(graph_dump, "\nUNARY %u %.17g %.17g %.17g\n", nodes.back(), pt[0], pt[1], pt[2]);
  }
  if (nodes.size() >= 2) {
    const std::vector<ImuMeasurement> meas = imu_buffer.GetRange(last_gps_timestamp, timestamp);
    if (meas.empty()) {
      std::fprintf(stderr, "Could not find imu measurements between : %f and %f\n", last_gps_timestamp, timestamp);
      return false;
    }
    slam.AddImuResidual(nodes.back() - 1, nodes.back(), meas);
    if (graph_dump) {
      std::fprintf(graph_dump, "IMU %u %u %zu\n", nodes.back() - 1, nodes.back(), meas.size());
      for (const ImuMeasurement& m : meas)
        std::fprintf(graph_dump, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", m.w[0], m.w[1], m.w[2], m.a[0],
                     m.a[1], m.a[2], m.time);
    }
  }
  inc_x = inc_y = inc_yaw = 0;       // reset the incremental accumulators (:218-224)
  differential_update = ba::SE3();
  gyro_update = ba::SE3();
  last_gps_timestamp = timestamp;
  return true;
}

// reference :233-238
void setup() {
  std::fprintf(stderr, "Init BA\n");
  ba::Options<double> options;
  options.trust_region_size = 100000;
  slam.Init(options);
  slam.SetGravity(ba::Vector3t({0.0, 0.0, 9.8}));
}

// reference :243-280
bool parse_file(const char* filename) {
  FILE* input = std::fopen(filename, "r");
  if (!input) { std::fprintf(stderr, "cannot open %s\n", filename); return false; }
  char name[16];
  bool ok = true;
  while (ok && std::fscanf(input, "%15s", name) != EOF) {
    if (std::strncmp(name, "ODO", 3) == 0) {
      double t, rr, rl;
      if (std::fscanf(input, "%lf %lf %lf", &t, &rr, &rl) == 3) update_incremental_pose(t, rr, rl);
    } else if (std::strncmp(name, "UTM", 3) == 0) {
      double t, e, n, alt;
      if (std::fscanf(input, "%lf %lf %lf %lf", &t, &e, &n, &alt) == 4) {
        if (nodes.size() >= 10000) break;
        ok = f_gps(t, e, n, alt);
      }
    } else if (std::strncmp(name, "IMU", 3) == 0) {
      double t, w[3], a[3];
      if (std::fscanf(input, "%lf %lf %lf %lf %lf %lf %lf", &t, w, w + 1, w + 2, a, a + 1, a + 2) == 7) {
        add_gyro_and_speed(t, w[0], w[1], w[2], speed);
        add_imu(t, w, a);
      }
    } else {
      std::fprintf(stderr, "Unknown symbol <%s>\n", name);
    }
  }
  std::fclose(input);
  return ok;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <log.dat> [--dump-graph <file>]\n", argv[0]); return 2; }
  if (argc >= 4 && std::strcmp(argv[2], "--dump-graph") == 0) graph_dump = std::fopen(argv[3], "w");
  setup();
  if (!parse_file(argv[1])) return 2;
  if (graph_dump) std::fclose(graph_dump);
  std::fprintf(stderr, "BA::Solve w [%zu] poses\n", nodes.size());
  slam.Solve(25, 0.2);  // reference :284-287
  std::fprintf(stderr, "finish BA::Solve\n");
  const ba::SolutionSummary<double>& s = slam.GetSolutionSummary();
  for (unsigned int id : nodes) {
    const auto& p = slam.GetPose(id);
    std::printf("NODE %u %.6f %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", id, p.time, p.t_wp.t[0],
                p.t_wp.t[1], p.t_wp.t[2], p.t_wp.q[0], p.t_wp.q[1], p.t_wp.q[2], p.t_wp.q[3], p.v_w[0], p.v_w[1], p.v_w[2]);
  }
  double e_proj, e_unary, e_binary, e_inertial;
  slam.GetErrors(e_proj, e_unary, e_binary, e_inertial);
  std::printf("SUMMARY result %d unary_error %.17g inertial_error %.17g delta_norm %.17g\n", (int)s.result, e_unary,
              e_inertial, (double)s.delta_norm);
  return s.IsResultGood() ? 0 : 1;
}
