// ORACLE — TEST INFRASTRUCTURE ONLY.
// Tiny fixed-size dense linear algebra + SO3/SE3 value types for the CPU
// restatement of arpg/ba's Gauss-Newton path.  Nothing under oracle/ is linked,
// imported or executed by the product (ba_amd/, include/); only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// The reference leans on Eigen (fixed-size matrices), Sophus (pre-1.0
// SE3Group/SO3Group) and Calibu; none of them is present in /root/reference or in
// this image, so their *published* semantics are restated here:
//   * quaternion storage order (x,y,z,w), Hamilton product (Eigen::Quaternion);
//   * SO3 exp/log with Sophus' small-angle branches (epsilon 1e-10);
//   * SO3/SE3 products renormalise the quaternion (Sophus SO3GroupBase::operator*=),
//     which is why the reference memcpy's raw coefficients in its integrator
//     (include/ba/Types.h:336-339).
// PARITY UNPINNED at this boundary: the reference holds no golden vectors for it.
#pragma once
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cassert>
#include <vector>
#include <algorithm>

namespace orc {

template <int R, int C>
struct Mat {
  double a[R * C];
  Mat() { for (int i = 0; i < R * C; ++i) a[i] = 0.0; }
  double& operator()(int r, int c) { return a[r * C + c]; }
  double operator()(int r, int c) const { return a[r * C + c]; }
  double& operator[](int i) { return a[i]; }
  double operator[](int i) const { return a[i]; }
  static Mat Zero() { return Mat(); }
  static Mat Identity() {
    Mat m;
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
    return m;
  }
  Mat<C, R> T() const {
    Mat<C, R> t;
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) t(c, r) = (*this)(r, c);
    return t;
  }
  double squaredNorm() const {
    double s = 0;
    for (int i = 0; i < R * C; ++i) s += a[i] * a[i];
    return s;
  }
  double norm() const { return std::sqrt(squaredNorm()); }
  template <int BR, int BC>
  Mat<BR, BC> block(int r0, int c0) const {
    Mat<BR, BC> b;
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) b(r, c) = (*this)(r0 + r, c0 + c);
    return b;
  }
  template <int BR, int BC>
  void setBlock(int r0, int c0, const Mat<BR, BC>& b) {
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) (*this)(r0 + r, c0 + c) = b(r, c);
  }
  template <int BR, int BC>
  void addBlock(int r0, int c0, const Mat<BR, BC>& b) {
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) (*this)(r0 + r, c0 + c) += b(r, c);
  }
  void setColZero(int c) {
    for (int r = 0; r < R; ++r) (*this)(r, c) = 0.0;
  }
};

template <int R, int K, int C>
inline Mat<R, C> operator*(const Mat<R, K>& x, const Mat<K, C>& y) {
  Mat<R, C> z;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += x(r, k) * y(k, c);
      z(r, c) = s;
    }
  return z;
}
template <int R, int C>
inline Mat<R, C> operator+(const Mat<R, C>& x, const Mat<R, C>& y) {
  Mat<R, C> z;
  for (int i = 0; i < R * C; ++i) z.a[i] = x.a[i] + y.a[i];
  return z;
}
template <int R, int C>
inline Mat<R, C> operator-(const Mat<R, C>& x, const Mat<R, C>& y) {
  Mat<R, C> z;
  for (int i = 0; i < R * C; ++i) z.a[i] = x.a[i] - y.a[i];
  return z;
}
template <int R, int C>
inline Mat<R, C> operator-(const Mat<R, C>& x) {
  Mat<R, C> z;
  for (int i = 0; i < R * C; ++i) z.a[i] = -x.a[i];
  return z;
}
template <int R, int C>
inline Mat<R, C> operator*(const Mat<R, C>& x, double s) {
  Mat<R, C> z;
  for (int i = 0; i < R * C; ++i) z.a[i] = x.a[i] * s;
  return z;
}
template <int R, int C>
inline Mat<R, C> operator*(double s, const Mat<R, C>& x) { return x * s; }
template <int R, int C>
inline Mat<R, C>& operator+=(Mat<R, C>& x, const Mat<R, C>& y) {
  for (int i = 0; i < R * C; ++i) x.a[i] += y.a[i];
  return x;
}
template <int R, int C>
inline Mat<R, C>& operator-=(Mat<R, C>& x, const Mat<R, C>& y) {
  for (int i = 0; i < R * C; ++i) x.a[i] -= y.a[i];
  return x;
}

typedef Mat<2, 1> Vec2;
typedef Mat<3, 1> Vec3;
typedef Mat<4, 1> Vec4;
typedef Mat<6, 1> Vec6;
typedef Mat<3, 3> Mat3;
typedef Mat<4, 4> Mat4;

inline Vec3 cross(const Vec3& a, const Vec3& b) {
  Vec3 c;
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
  return c;
}
inline Mat3 hat(const Vec3& v) {
  Mat3 m;
  m(0, 1) = -v[2]; m(0, 2) = v[1];
  m(1, 0) = v[2];  m(1, 2) = -v[0];
  m(2, 0) = -v[1]; m(2, 1) = v[0];
  return m;
}

// General inverse by Gauss-Jordan with partial pivoting (Eigen's fixed-size
// inverse() uses closed forms up to 4x4 and PartialPivLU above; results agree to
// rounding).
template <int N>
inline Mat<N, N> inverse(const Mat<N, N>& m) {
  double w[N][2 * N];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      w[r][c] = m(r, c);
      w[r][N + c] = (r == c) ? 1.0 : 0.0;
    }
  for (int k = 0; k < N; ++k) {
    int p = k;
    for (int r = k + 1; r < N; ++r)
      if (std::fabs(w[r][k]) > std::fabs(w[p][k])) p = r;
    if (p != k)
      for (int c = 0; c < 2 * N; ++c) std::swap(w[p][c], w[k][c]);
    const double d = 1.0 / w[k][k];
    for (int c = 0; c < 2 * N; ++c) w[k][c] *= d;
    for (int r = 0; r < N; ++r) {
      if (r == k) continue;
      const double f = w[r][k];
      if (f == 0.0) continue;
      for (int c = 0; c < 2 * N; ++c) w[r][c] -= f * w[k][c];
    }
  }
  Mat<N, N> out;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) out(r, c) = w[r][N + c];
  return out;
}

// 3x3: Eigen's fixed-size inverse is the cofactor form (Eigen/src/LU/InverseImpl.h,
// compute_inverse<.., 3>: cofactors, determinant from the first column, one reciprocal) — what the
// reference's `jtj.inverse()` of a landmark's 3x3 V runs (BundleAdjuster.cpp:438-440).  For a nearly
// singular V (a landmark seen from three poses with a short baseline) the elimination order matters
// at the 1e-8 level, so the restatement follows the same form.
template <>
inline Mat<3, 3> inverse<3>(const Mat<3, 3>& m) {
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
  };
  const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  const double invdet = 1.0 / (c00 * m(0, 0) + c10 * m(1, 0) + c20 * m(2, 0));
  Mat<3, 3> out;
  out(0, 0) = c00 * invdet; out(0, 1) = c10 * invdet; out(0, 2) = c20 * invdet;
  out(1, 0) = cof(0, 1) * invdet; out(1, 1) = cof(1, 1) * invdet; out(1, 2) = cof(2, 1) * invdet;
  out(2, 0) = cof(0, 2) * invdet; out(2, 1) = cof(1, 2) * invdet; out(2, 2) = cof(2, 2) * invdet;
  return out;
}

// Principal square root of a symmetric positive (semi)definite matrix by cyclic
// Jacobi eigen-decomposition.  The reference calls Eigen's unsupported
// MatrixFunctions sqrt() (Schur method) on cov_inv matrices
// (include/ba/BundleAdjuster.h:399,445; src/BundleAdjuster.cpp:1470,1527); for the
// symmetric PD inputs it receives, the principal root is V diag(sqrt(l)) V^T.
template <int N>
inline Mat<N, N> sqrt_spd(const Mat<N, N>& m_in) {
  Mat<N, N> A;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) A(r, c) = 0.5 * (m_in(r, c) + m_in(c, r));
  Mat<N, N> V = Mat<N, N>::Identity();
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) {
        if (r != c) off += A(r, c) * A(r, c);
        else diag += A(r, c) * A(r, c);
      }
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = A(p, q);
        if (apq == 0.0) continue;
        const double theta = (A(q, q) - A(p, p)) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) /
                         (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) {
          const double akp = A(k, p), akq = A(k, q);
          A(k, p) = c * akp - s * akq;
          A(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = A(p, k), aqk = A(q, k);
          A(p, k) = c * apk - s * aqk;
          A(q, k) = s * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = V(k, p), vkq = V(k, q);
          V(k, p) = c * vkp - s * vkq;
          V(k, q) = s * vkp + c * vkq;
        }
      }
  }
  Mat<N, N> out;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      double s = 0;
      for (int k = 0; k < N; ++k) {
        const double l = A(k, k) > 0 ? std::sqrt(A(k, k)) : 0.0;
        s += V(r, k) * l * V(c, k);
      }
      out(r, c) = s;
    }
  return out;
}

// ---------------------------------------------------------------------------
// Quaternion (x,y,z,w), Eigen::Quaternion semantics.
struct Quat {
  double x, y, z, w;
  Quat() : x(0), y(0), z(0), w(1) {}
  Quat(double x_, double y_, double z_, double w_) : x(x_), y(y_), z(z_), w(w_) {}
  Vec4 coeffs() const { Vec4 v; v[0] = x; v[1] = y; v[2] = z; v[3] = w; return v; }
  Vec3 vec() const { Vec3 v; v[0] = x; v[1] = y; v[2] = z; return v; }
  Quat conjugate() const { return Quat(-x, -y, -z, w); }
  void normalize() {
    const double n = std::sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
  }
  // Eigen::Quaternion::toRotationMatrix()
  Mat3 matrix() const {
    Mat3 R;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
    return R;
  }
};
// Hamilton product a (x) b.
inline Quat qmul(const Quat& a, const Quat& b) {
  return Quat(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
              a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}

static const double kSophusEps = 1e-10;  // Sophus::SophusConstants<double>::epsilon()

// Sophus::SO3Group (pre-1.0).
struct SO3 {
  Quat q;
  SO3() {}
  explicit SO3(const Quat& q_in) : q(q_in) { q.normalize(); }  // ctor normalises
  static SO3 raw(const Quat& q_in) { SO3 s; s.q = q_in; return s; }  // memcpy path
  const Quat& unit_quaternion() const { return q; }
  SO3 inverse() const { return raw(q.conjugate()); }
  Mat3 matrix() const { return q.matrix(); }
  Mat3 Adj() const { return q.matrix(); }
  SO3 operator*(const SO3& o) const {
    SO3 r = raw(qmul(q, o.q));
    r.q.normalize();  // SO3GroupBase::operator*= renormalises
    return r;
  }
  Vec3 operator*(const Vec3& p) const {
    // Eigen QuaternionBase::_transformVector
    const Vec3 qv = q.vec();
    Vec3 uv = cross(qv, p);
    uv = uv + uv;
    return p + uv * q.w + cross(qv, uv);
  }
  // Sophus SO3Group::exp / expAndTheta.
  static SO3 exp(const Vec3& omega) {
    const double theta_sq = omega.squaredNorm();
    const double theta = std::sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    double imag_factor, real_factor;
    if (theta < kSophusEps) {
      const double theta_po4 = theta_sq * theta_sq;
      imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
      real_factor = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
      const double sin_half_theta = std::sin(half_theta);
      imag_factor = sin_half_theta / theta;
      real_factor = std::cos(half_theta);
    }
    return SO3(Quat(imag_factor * omega[0], imag_factor * omega[1],
                    imag_factor * omega[2], real_factor));
  }
  // Sophus SO3Group::log / logAndTheta.
  Vec3 log() const {
    const double squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
    const double n = std::sqrt(squared_n);
    const double w = q.w;
    double two_atan_nbyw_by_n;
    if (n < kSophusEps) {
      const double squared_w = w * w;
      two_atan_nbyw_by_n = 2.0 / w - 2.0 * squared_n / (w * squared_w);
    } else {
      if (std::fabs(w) < kSophusEps) {
        two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
      } else {
        two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
      }
    }
    return q.vec() * two_atan_nbyw_by_n;
  }
  // Sophus SO3Group::generator(i) = hat(e_i).
  static Mat3 generator(int i) {
    Vec3 e;
    e[i] = 1.0;
    return hat(e);
  }
};

// Sophus::SE3Group (pre-1.0).
struct SE3 {
  SO3 r;
  Vec3 t;
  SE3() {}
  SE3(const SO3& r_in, const Vec3& t_in) : r(r_in), t(t_in) {}
  const SO3& so3() const { return r; }
  SO3& so3() { return r; }
  const Vec3& translation() const { return t; }
  Vec3& translation() { return t; }
  const Quat& unit_quaternion() const { return r.q; }
  Mat3 rotationMatrix() const { return r.matrix(); }
  SE3 inverse() const {
    const SO3 ri = r.inverse();
    return SE3(ri, ri * (t * -1.0));
  }
  SE3 operator*(const SE3& o) const {
    SE3 out = *this;
    out.t = out.t + r * o.t;
    out.r = r * o.r;
    return out;
  }
  Mat4 matrix() const {
    Mat4 m = Mat4::Identity();
    m.setBlock<3, 3>(0, 0, r.matrix());
    m(0, 3) = t[0]; m(1, 3) = t[1]; m(2, 3) = t[2];
    return m;
  }
};

}  // namespace orc
