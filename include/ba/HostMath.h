// Small host-side linear algebra for the 6x6 covariance handling the reference does at
// Add* time (covariance.inverse() and cov_inv.sqrt(),
// /root/reference/include/ba/BundleAdjuster.h:398-399,444-445).
#pragma once
#include <cmath>
#include <utility>

#include "Types.h"

namespace ba {
namespace hostmath {

// inverse by LU-free Gauss-Jordan elimination with row pivoting
inline Matrix6t inverse6(const Matrix6t& a_in) {
  const int N = 6;
  double a[N][N], inv[N][N];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) { a[r][c] = a_in(r, c); inv[r][c] = r == c ? 1.0 : 0.0; }
  for (int col = 0; col < N; ++col) {
    int piv = col;
    for (int r = col + 1; r < N; ++r)
      if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
    if (piv != col)
      for (int c = 0; c < N; ++c) { std::swap(a[piv][c], a[col][c]); std::swap(inv[piv][c], inv[col][c]); }
    const double s = 1.0 / a[col][col];
    for (int c = 0; c < N; ++c) { a[col][c] *= s; inv[col][c] *= s; }
    for (int r = 0; r < N; ++r) {
      if (r == col) continue;
      const double f = a[r][col];
      if (f == 0.0) continue;
      for (int c = 0; c < N; ++c) { a[r][c] -= f * a[col][c]; inv[r][c] -= f * inv[col][c]; }
    }
  }
  Matrix6t out;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) out(r, c) = inv[r][c];
  return out;
}

// principal square root of a symmetric positive semi-definite 6x6 (cyclic Jacobi
// eigen-decomposition, V diag(sqrt(lambda)) V^T)
inline Matrix6t sqrt_spd6(const Matrix6t& m) {
  const int N = 6;
  double A[N][N], V[N][N];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) { A[r][c] = 0.5 * (m(r, c) + m(c, r)); V[r][c] = r == c ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0, dia = 0;
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) (r == c ? dia : off) += A[r][c] * A[r][c];
    if (off <= 1e-34 * dia || off < 1e-300) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (A[p][q] == 0.0) continue;
        const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < N; ++k) {
          const double x = A[k][p], y = A[k][q];
          A[k][p] = cs * x - sn * y; A[k][q] = sn * x + cs * y;
        }
        for (int k = 0; k < N; ++k) {
          const double x = A[p][k], y = A[q][k];
          A[p][k] = cs * x - sn * y; A[q][k] = sn * x + cs * y;
        }
        for (int k = 0; k < N; ++k) {
          const double x = V[k][p], y = V[k][q];
          V[k][p] = cs * x - sn * y; V[k][q] = sn * x + cs * y;
        }
      }
  }
  Matrix6t out;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      double s = 0;
      for (int k = 0; k < N; ++k) s += V[r][k] * (A[k][k] > 0 ? std::sqrt(A[k][k]) : 0.0) * V[c][k];
      out(r, c) = s;
    }
  return out;
}

}  // namespace hostmath
}  // namespace ba
