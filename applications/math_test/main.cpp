// The analytic-vs-numeric Jacobian report of the reference's applications/math_test (main.cpp:26-148):
// d log(q)/dq, and the quaternion chain of a relative-rotation error
//     e(w) = log( (q_wa exp(w)) q_ab q_wb^-1 )
// differentiated at w = 0 in three stages — d(q_wa exp(w))/dw, d(chain quaternion)/dw, de/dw — each against
// central differences, printed as error norms.  Written on include/ba/Utils.h (the reference's helper
// names over the code the gfx950 kernels use); host only, no GPU.  Unlike the reference program the exit
// code says whether every norm is small.  Its second half (block-sparse matrix products against dense
// Eigen, main.cpp:157-314) exercises containers this path does not have.
#include <ba/Utils.h>

#include <cmath>
#include <cstdio>
#include <random>

namespace {
typedef ba::Vector4t Quat;  // x, y, z, w

Quat qmul(const Quat& a, const Quat& b) {  // Hamilton product, no renormalisation
  return Quat({a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
               a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]});
}
Quat qinv(const Quat& a) { return Quat({-a[0], -a[1], -a[2], a[3]}); }
Quat qexp(const ba::Vector3t& w) {  // through exp_decoupled of the identity
  const ba::SE3 t = ba::exp_decoupled(ba::SE3(), ba::Vector6t({0.0, 0.0, 0.0, w[0], w[1], w[2]}));
  return Quat({t.q[0], t.q[1], t.q[2], t.q[3]});
}
ba::Vector3t qlog(const Quat& q) {  // through log_decoupled against the identity
  const double zero[3] = {0, 0, 0};
  const ba::Vector6t l = ba::log_decoupled(ba::SE3(zero, q.data()), ba::SE3());
  return ba::Vector3t({l[3], l[4], l[5]});
}
template <int R, int K, int C>
ba::Mat<R, C> mul(const ba::Mat<R, K>& a, const ba::Mat<K, C>& b) {
  ba::Mat<R, C> o;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < K; ++k) o(r, c) += a(r, k) * b(k, c);
  return o;
}
template <int R, int C>
double report(const char* name, const ba::Mat<R, C>& analytic, const ba::Mat<R, C>& numeric) {
  double n2 = 0;
  for (int i = 0; i < R * C; ++i) n2 += (analytic.data()[i] - numeric.data()[i]) * (analytic.data()[i] - numeric.data()[i]);
  std::printf("%-14s analytic - numeric, norm: %.3e\n", name, std::sqrt(n2));
  return std::sqrt(n2);
}
}  // namespace

int main() {
  std::mt19937 rng(3);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  auto rand3 = [&]() { return ba::Vector3t({u(rng), u(rng), u(rng)}); };
  const double h = 1e-6;
  double worst = 0;

  {  // d log(q) / dq with the four quaternion entries perturbed freely (reference main.cpp:30-53, Utils.h:188-219)
    const Quat q({0.000718076, 0.0139853, -4.9437e-05, 0.999902});  // the reference's sample: a small rotation
    ba::Mat<3, 4> fd;
    for (int j = 0; j < 4; ++j) {
      Quat qp = q, qm = q;
      qp[j] += h; qm[j] -= h;
      // log of the raw (not renormalised) quaternion: 2 atan2(|v|, w) v / |v|
      auto rawlog = [](const Quat& x) {
        const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const double s = n > 0 ? 2.0 * std::atan2(n, x[3]) / n : 2.0 / x[3];
        return ba::Vector3t({s * x[0], s * x[1], s * x[2]});
      };
      const ba::Vector3t lp = rawlog(qp), lm = rawlog(qm);
      for (int r = 0; r < 3; ++r) fd(r, j) = (lp[r] - lm[r]) / (2 * h);
    }
    worst = std::fmax(worst, report("dlog_dq", ba::dlog_dq(q), fd));
  }

  const Quat q_wa = qexp(rand3()), q_ab = qexp(rand3()), q_wb = qexp(rand3());
  const Quat tail = qmul(q_ab, qinv(q_wb));
  const ba::Mat<4, 3> dexp0 = ba::dq_exp_dw(ba::Vector3t({0.0, 0.0, 0.0}));
  {  // d (q_wa exp(w)) / dw = dq1q2_dq2(q_wa) dq_exp_dw(0)   (reference :92-110)
    ba::Mat<4, 3> fd;
    for (int j = 0; j < 3; ++j) {
      ba::Vector3t wp, wm;
      wp[j] = h; wm[j] = -h;
      const Quat a = qmul(q_wa, qexp(wp)), b = qmul(q_wa, qexp(wm));
      for (int r = 0; r < 4; ++r) fd(r, j) = (a[r] - b[r]) / (2 * h);
    }
    worst = std::fmax(worst, report("dExp_dq", mul(ba::dq1q2_dq2(q_wa), dexp0), fd));
  }
  {  // d ((q_wa exp(w)) q_ab q_wb^-1) / dw = dq1q2_dq1(tail) dq1q2_dq2(q_wa) dq_exp_dw(0)   (:113-129)
    ba::Mat<4, 3> fd;
    for (int j = 0; j < 3; ++j) {
      ba::Vector3t wp, wm;
      wp[j] = h; wm[j] = -h;
      const Quat a = qmul(qmul(q_wa, qexp(wp)), tail), b = qmul(qmul(q_wa, qexp(wm)), tail);
      for (int r = 0; r < 4; ++r) fd(r, j) = (a[r] - b[r]) / (2 * h);
    }
    worst = std::fmax(worst, report("dTerror", mul(ba::dq1q2_dq1(tail), mul(ba::dq1q2_dq2(q_wa), dexp0)), fd));
  }
  {  // d log(chain) / dw = dlog_dq(chain) . the product above   (:131-148)
    ba::Mat<3, 3> fd;
    for (int j = 0; j < 3; ++j) {
      ba::Vector3t wp, wm;
      wp[j] = h; wm[j] = -h;
      const ba::Vector3t a = qlog(qmul(qmul(q_wa, qexp(wp)), tail)), b = qlog(qmul(qmul(q_wa, qexp(wm)), tail));
      for (int r = 0; r < 3; ++r) fd(r, j) = (a[r] - b[r]) / (2 * h);
    }
    const ba::Mat<3, 3> an = mul(ba::dlog_dq(qmul(q_wa, tail)), mul(ba::dq1q2_dq1(tail), mul(ba::dq1q2_dq2(q_wa), dexp0)));
    worst = std::fmax(worst, report("dlog_Terror", an, fd));
  }
  const double t0 = ba::Tic();
  for (int i = 0; i < 1000; ++i) (void)ba::dlog_decoupled_dx(ba::SE3(), ba::SE3());
  std::printf("1000 x dlog_decoupled_dx took %.6f s\n", ba::Toc(t0));
  std::printf(worst < 1e-6 ? "math_test: ok\n" : "math_test: FAIL (%.3e)\n", worst);
  return worst < 1e-6 ? 0 : 1;
}
