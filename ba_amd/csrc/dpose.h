// Device/host math of the pose-pose residuals (unary prior, binary pose-pose, IMU
// pre-integration) of the gfx950 engine.
//
// What is computed follows the reference:
//   unary    /root/reference/src/BundleAdjuster.cpp:1434-1482   (a17 in SURVEY.md §8a)
//   binary   /root/reference/src/BundleAdjuster.cpp:1395-1428   (a16)
//   inertial /root/reference/include/ba/parallel_algos.h:178-358 with the RK4 integrator
//            of /root/reference/include/ba/Types.h:324-738      (a18)
// using the quaternion/SE3 derivative blocks of /root/reference/include/ba/Utils.h
// (cited per function).  Unlike the reference, no square roots of covariance matrices
// are taken: J^T S^-1 J and J^T S^-1 r are formed directly.
// BA_HD lets tests compile this for the host and compare with the oracle.
#pragma once
#include "dmath.h"

namespace bad {

// small dense row-major matrix
template <int R, int C>
struct DM {
  double m[R * C];
  BA_HD double& operator()(int r, int c) { return m[r * C + c]; }
  BA_HD double operator()(int r, int c) const { return m[r * C + c]; }
  BA_HD void zero() { for (int i = 0; i < R * C; ++i) m[i] = 0.0; }
  BA_HD void identity() { zero(); for (int i = 0; i < (R < C ? R : C); ++i) m[i * C + i] = 1.0; }
};
template <int R, int K, int C>
BA_HD DM<R, C> mm(const DM<R, K>& a, const DM<K, C>& b) {
  DM<R, C> o;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += a(r, k) * b(k, c);
      o(r, c) = s;
    }
  return o;
}
template <int R, int K, int C>
BA_HD DM<R, C> mmT(const DM<R, K>& a, const DM<C, K>& b) {  // a * b^T
  DM<R, C> o;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += a(r, k) * b(c, k);
      o(r, c) = s;
    }
  return o;
}
template <int K, int R, int C>
BA_HD DM<R, C> mTm(const DM<K, R>& a, const DM<K, C>& b) {  // a^T * b
  DM<R, C> o;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += a(k, r) * b(k, c);
      o(r, c) = s;
    }
  return o;
}
// Wave-cooperative products (device only).  When the 64 lanes of ONE wavefront run the same scalar
// code on the same residual — every lane holds identical copies of the operands — a product is dealt
// element-wise to the lanes (each element summed over k in the order of mm / mmT / mTm: bitwise the
// scalar result), exchanged through `lds` (>= 3 * 225 doubles, workgroup = that one wavefront) and read
// back by every lane.  ctx == nullptr (always on the host): the scalar product.
struct WaveCtx { double* lds; int lane; };
#if defined(__HIP_DEVICE_COMPILE__)
// operands first into LDS (each lane copies its share of the — identical — private copies: a few
// dynamically indexed private-memory reads instead of 2 K per output element), then the elements
// from LDS.  NA / NB: element counts of the operands; IA / IB: their index expressions in (r, k, c).
#define BAD_WAVE_PRODUCT(NA, NB, IA, IB)                              \
  if (w) {                                                            \
    double* la = w->lds;                                              \
    double* lb = la + (NA);                                           \
    double* lo = lb + (NB);                                           \
    for (int e = w->lane; e < (NA); e += 64) la[e] = a.m[e];          \
    for (int e = w->lane; e < (NB); e += 64) lb[e] = b.m[e];          \
    __syncthreads();                                                  \
    for (int e = w->lane; e < R * C; e += 64) {                       \
      const int r = e / C, c = e - r * C;                             \
      double s = 0.0;                                                 \
      for (int k = 0; k < K; ++k) s += la[IA] * lb[IB];               \
      lo[e] = s;                                                      \
    }                                                                 \
    __syncthreads();                                                  \
    DM<R, C> o;                                                       \
    for (int e = 0; e < R * C; ++e) o.m[e] = lo[e];                   \
    __syncthreads();                                                  \
    return o;                                                         \
  }
#else
#define BAD_WAVE_PRODUCT(NA, NB, IA, IB) (void)w;
#endif
template <int R, int K, int C>
BA_HD DM<R, C> mm(const DM<R, K>& a, const DM<K, C>& b, const WaveCtx* w) {
  BAD_WAVE_PRODUCT(R * K, K * C, r * K + k, k * C + c)
  return mm(a, b);
}
template <int R, int K, int C>
BA_HD DM<R, C> mmT(const DM<R, K>& a, const DM<C, K>& b, const WaveCtx* w) {
  BAD_WAVE_PRODUCT(R * K, C * K, r * K + k, c * K + k)
  return mmT(a, b);
}
template <int K, int R, int C>
BA_HD DM<R, C> mTm(const DM<K, R>& a, const DM<K, C>& b, const WaveCtx* w) {
  BAD_WAVE_PRODUCT(K * R, K * C, k * R + r, k * C + c)
  return mTm(a, b);
}
#undef BAD_WAVE_PRODUCT
template <int R, int C>
BA_HD DM<R, C> madd(const DM<R, C>& a, const DM<R, C>& b, double sb = 1.0) {
  DM<R, C> o;
  for (int i = 0; i < R * C; ++i) o.m[i] = a.m[i] + sb * b.m[i];
  return o;
}
template <int R, int C, int BR, int BC>
BA_HD void set_block(DM<R, C>& dst, int r0, int c0, const DM<BR, BC>& b, double s = 1.0) {
  for (int r = 0; r < BR; ++r)
    for (int c = 0; c < BC; ++c) dst(r0 + r, c0 + c) = s * b(r, c);
}
template <int R, int C, int BR, int BC>
BA_HD void add_block(DM<R, C>& dst, int r0, int c0, const DM<BR, BC>& b, double s = 1.0) {
  for (int r = 0; r < BR; ++r)
    for (int c = 0; c < BC; ++c) dst(r0 + r, c0 + c) += s * b(r, c);
}
template <int BR, int BC, int R, int C>
BA_HD DM<BR, BC> get_block(const DM<R, C>& a, int r0, int c0) {
  DM<BR, BC> o;
  for (int r = 0; r < BR; ++r)
    for (int c = 0; c < BC; ++c) o(r, c) = a(r0 + r, c0 + c);
  return o;
}
BA_HD DM<3, 3> dm3(const M3& R) {
  DM<3, 3> o;
  for (int i = 0; i < 9; ++i) o.m[i] = R.m[i];
  return o;
}
// in-place inverse of the leading n x n block (Gauss-Jordan, partial pivoting)
template <int N>
BA_HD void invert_leading(DM<N, N>& a, int n) {
  DM<N, N> inv;
  inv.identity();
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = fabs(a(col, col));
    for (int r = col + 1; r < n; ++r)
      if (fabs(a(r, col)) > best) { best = fabs(a(r, col)); piv = r; }
    if (piv != col)
      for (int c = 0; c < n; ++c) {
        double t = a(piv, c); a(piv, c) = a(col, c); a(col, c) = t;
        t = inv(piv, c); inv(piv, c) = inv(col, c); inv(col, c) = t;
      }
    const double s = 1.0 / a(col, col);
    for (int c = 0; c < n; ++c) { a(col, c) *= s; inv(col, c) *= s; }
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const double f = a(r, col);
      if (f == 0.0) continue;
      for (int c = 0; c < n; ++c) { a(r, c) -= f * a(col, c); inv(r, c) -= f * inv(col, c); }
    }
  }
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) a(r, c) = inv(r, c);
}

// The same elimination with the 2 n^2 entries of [a | inv] dealt to the lanes of one wavefront and the
// matrices in LDS (w->lds: 4 * N * N doubles, two copies of [a | inv]): column by column every lane finds
// the pivot (it reads the column: same decision everywhere) and writes its entries of the NEXT copy from
// the current one — row swap, scaling of the pivot row and elimination of the other rows in one pass,
// each entry by the expression of invert_leading (bitwise the same inverse), one barrier per column.
template <int N>
BA_HD void invert_leading(DM<N, N>& a, int n, const WaveCtx* w) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (w) {
    double* B0 = w->lds;              // [A | I] of even columns
    double* B1 = w->lds + 2 * N * N;  // ... of odd columns
    const int lane = w->lane;
    for (int e = lane; e < N * N; e += 64) { B0[e] = a.m[e]; B0[N * N + e] = ((e / N) == (e % N)) ? 1.0 : 0.0; }
    __syncthreads();
    for (int col = 0; col < n; ++col) {
      const double* S = (col & 1) ? B1 : B0;
      double* D = (col & 1) ? B0 : B1;
      int piv = col;
      double best = fabs(S[col * N + col]);
      for (int r = col + 1; r < n; ++r) {
        const double v = fabs(S[r * N + col]);
        if (v > best) { best = v; piv = r; }
      }
      const double s = 1.0 / S[piv * N + col];  // a(col, col) after the swap
      for (int e = lane; e < 2 * n * n; e += 64) {
        const int m = e / (n * n), rc = e % (n * n), r = rc / n, c = rc % n;
        const double* M = S + m * N * N;
        const double pc = M[piv * N + c] * s;  // the scaled pivot row
        double v;
        if (r == col) v = pc;
        else {
          const int rs = r == piv ? col : r;   // the row that sits at r after the swap
          const double f = S[rs * N + col];
          v = M[rs * N + c];
          if (f != 0.0) v -= f * pc;
        }
        D[m * N * N + r * N + c] = v;
      }
      __syncthreads();
    }
    const double* R = ((n & 1) ? B1 : B0) + N * N;
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) a(r, c) = R[r * N + c];
    __syncthreads();
    return;
  }
#else
  (void)w;
#endif
  invert_leading(a, n);
}

// ---- rigid transform with quaternion (x,y,z,w) --------------------------------------
struct Tq { double t[3]; double q[4]; };
BA_HD Tq tq_from7(const double* p) {
  Tq o;
  for (int i = 0; i < 3; ++i) o.t[i] = p[i];
  for (int i = 0; i < 4; ++i) o.q[i] = p[3 + i];
  return o;
}
BA_HD V3 qrot(const double* q, V3 v) {  // Eigen _transformVector (valid for non-unit q as used)
  const V3 qv = v3(q[0], q[1], q[2]);
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + uv * q[3] + cross(qv, uv);
}
BA_HD Tq tq_mul(const Tq& a, const Tq& b) {  // Sophus SE3 product (renormalised rotation)
  Tq o;
  const V3 t = qrot(a.q, v3(b.t[0], b.t[1], b.t[2]));
  o.t[0] = a.t[0] + t.x; o.t[1] = a.t[1] + t.y; o.t[2] = a.t[2] + t.z;
  quat_mul(a.q, b.q, o.q);
  quat_normalize(o.q);
  return o;
}
BA_HD Tq tq_inv(const Tq& a) {
  Tq o;
  o.q[0] = -a.q[0]; o.q[1] = -a.q[1]; o.q[2] = -a.q[2]; o.q[3] = a.q[3];
  const V3 t = qrot(o.q, v3(-a.t[0], -a.t[1], -a.t[2]));
  o.t[0] = t.x; o.t[1] = t.y; o.t[2] = t.z;
  return o;
}
// Utils.h:354-360
BA_HD void log_decoupled(const Tq& a, const Tq& b, double* r6) {
  for (int i = 0; i < 3; ++i) r6[i] = a.t[i] - b.t[i];
  double qb[4] = {-b.q[0], -b.q[1], -b.q[2], b.q[3]}, q[4];
  quat_mul(a.q, qb, q);
  quat_normalize(q);
  const V3 w = so3_log(q);
  r6[3] = w.x; r6[4] = w.y; r6[5] = w.z;
}
// Utils.h:277-282 / 286-291: left / right quaternion multiplication matrices
BA_HD DM<4, 4> qL(const double* a) {
  DM<4, 4> m;
  const double r[16] = {a[3], -a[2], a[1], a[0], a[2], a[3], -a[0], a[1],
                        -a[1], a[0], a[3], a[2], -a[0], -a[1], -a[2], a[3]};
  for (int i = 0; i < 16; ++i) m.m[i] = r[i];
  return m;
}
BA_HD DM<4, 4> qR(const double* b) {
  DM<4, 4> m;
  const double r[16] = {b[3], b[2], -b[1], b[0], -b[2], b[3], b[0], b[1],
                        b[1], -b[0], b[3], b[2], -b[0], -b[1], -b[2], b[3]};
  for (int i = 0; i < 16; ++i) m.m[i] = r[i];
  return m;
}
// Utils.h:295-312: partial derivatives of the polynomial R(q) v
BA_HD DM<3, 4> dqx_dq(const double* q, V3 p) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  DM<3, 4> J;
  J(0, 0) = 2 * qy * p.y + 2 * qz * p.z;
  J(1, 0) = 2 * qy * p.x - 4 * qx * p.y - 2 * qw * p.z;
  J(2, 0) = 2 * qz * p.x + 2 * qw * p.y - 4 * qx * p.z;
  J(0, 1) = 2 * qx * p.y - 4 * qy * p.x + 2 * qw * p.z;
  J(1, 1) = 2 * qx * p.x + 2 * qz * p.z;
  J(2, 1) = 2 * qz * p.y - 2 * qw * p.x - 4 * qy * p.z;
  J(0, 2) = 2 * qx * p.z - 2 * qw * p.y - 4 * qz * p.x;
  J(1, 2) = 2 * qy * p.z + 2 * qw * p.x - 4 * qz * p.y;
  J(2, 2) = 2 * qy * p.y + 2 * qx * p.x;
  J(0, 3) = 2 * qy * p.z - 2 * qz * p.y;
  J(1, 3) = 2 * qz * p.x - 2 * qx * p.z;
  J(2, 3) = 2 * qx * p.y - 2 * qy * p.x;
  return J;
}
// Utils.h:137-185
BA_HD DM<3, 4> dlog_dq(const double* q) {
  const double v[3] = {q[0], q[1], q[2]};
  const double w = q[3];
  const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double n = sqrt(n2);
  DM<3, 4> J;
  if (n < 1e-9) {
    const double iw3 = 1.0 / (w * w * w);
    const double two_n2 = 2 * n2;
    const double dcol = (3 * two_n2) / (w * w * w * w) - 2 / (w * w);
    const double diag = 2 / w - two_n2 * iw3;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) J(i, j) = -4 * v[i] * v[j] * iw3;
      J(i, i) += diag;
      J(i, 3) = v[i] * dcol;
    }
  } else {
    const double c = 1.0 / (n2 / (w * w) + 1.0);
    const double at = atan(n / w);
    const double in3 = 1.0 / (n2 * n), in2 = 1.0 / n2, iw = 1.0 / w;
    const double f = 2 * at / n;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) J(i, j) = 2 * v[i] * v[j] * (c * in2 * iw - at * in3);
      J(i, i) += f;
      J(i, 3) = -(2 * v[i] * c) / (w * w);
    }
  }
  return J;
}
// Utils.h:252-266 (the reference's truncated series, kept as is)
BA_HD DM<4, 3> dq_exp_dw(V3 w) {
  const double wv[3] = {w.x, w.y, w.z};
  const double t = sqrt(dot(w, w));
  const double a = t / 20 - 1, b = t * t / 48 - 0.5, t2 = t * t;
  DM<4, 3> J;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) J(i, j) = (a * wv[i] * wv[j]) / 24;
    J(i, i) = (a * wv[i] * wv[i]) / 24 - t2 / 48 + 0.5;
    J(3, i) = (b * wv[i]) / 2;
  }
  return J;
}
BA_HD DM<4, 3> half_identity() {  // dq_exp_dw(0)
  DM<4, 3> E;
  E.zero();
  E(0, 0) = E(1, 1) = E(2, 2) = 0.5;
  return E;
}
// Utils.h:451-489
BA_HD DM<7, 6> dexp_decoupled_dx(const Tq& t) {
  DM<7, 6> J;
  J.zero();
  J(0, 0) = J(1, 1) = J(2, 2) = 1.0;
  set_block(J, 3, 3, mm(qL(t.q), half_identity()));
  return J;
}
// Utils.h:493-536
BA_HD DM<7, 6> dinv_exp_decoupled_dx(const Tq& t) {
  const double qi[4] = {-t.q[0], -t.q[1], -t.q[2], t.q[3]};
  const DM<4, 3> RE = mm(qR(qi), half_identity());
  DM<7, 6> J;
  J.zero();
  const M3 Rt = transpose(quat_to_rot(t.q[0], t.q[1], t.q[2], t.q[3]));
  set_block(J, 0, 0, dm3(Rt), -1.0);
  set_block(J, 0, 3, mm(dqx_dq(qi, v3(t.t[0], t.t[1], t.t[2])), RE));
  set_block(J, 3, 3, RE, -1.0);
  return J;
}
// Utils.h:587-639
BA_HD DM<7, 7> dt1_t2_dt1(const Tq& t1, const Tq& t2) {
  DM<7, 7> J;
  J.zero();
  J(0, 0) = J(1, 1) = J(2, 2) = 1.0;
  set_block(J, 0, 3, dqx_dq(t1.q, v3(t2.t[0], t2.t[1], t2.t[2])));
  set_block(J, 3, 3, qR(t2.q));
  return J;
}
// Utils.h:643-694
BA_HD DM<7, 7> dt1_t2_dt2(const Tq& t1) {
  DM<7, 7> J;
  J.zero();
  set_block(J, 0, 0, dm3(quat_to_rot(t1.q[0], t1.q[1], t1.q[2], t1.q[3])));
  set_block(J, 3, 3, qL(t1.q));
  return J;
}
// Utils.h:388-397
BA_HD DM<6, 7> dLog_decoupled_dt1(const Tq& t1, const Tq& t2) {
  DM<6, 7> J;
  J.zero();
  J(0, 0) = J(1, 1) = J(2, 2) = 1.0;
  const Tq t2i = tq_inv(t2);
  const Tq t12 = tq_mul(t1, t2i);
  set_block(J, 3, 3, mm(dlog_dq(t12.q), qR(t2i.q)));
  return J;
}
// Utils.h:401-447
BA_HD DM<6, 7> dlog_decoupled_dt2(const Tq& t1, const Tq& t2) {
  DM<6, 7> J;
  J.zero();
  J(0, 0) = J(1, 1) = J(2, 2) = -1.0;
  double qb[4] = {-t2.q[0], -t2.q[1], -t2.q[2], t2.q[3]}, ql[4];
  quat_mul(t1.q, qb, ql);
  quat_normalize(ql);
  DM<4, 4> dinv;
  dinv.zero();
  dinv(0, 0) = dinv(1, 1) = dinv(2, 2) = -1.0; dinv(3, 3) = 1.0;
  set_block(J, 3, 3, mm(mm(dlog_dq(ql), qL(t1.q)), dinv));
  return J;
}
// Utils.h:374-384
BA_HD DM<6, 6> dlog_decoupled_dx(const Tq& a, const Tq& b) {
  DM<6, 6> J;
  J.identity();
  const Tq bi = tq_inv(b);
  const Tq ab = tq_mul(a, bi);
  set_block(J, 3, 3, mm(mm(mm(dlog_dq(ab.q), qL(a.q)), qR(bi.q)), half_identity()));
  return J;
}

// ---- per-residual normal-equation blocks ---------------------------------------------
// Output of a pose-pose residual: Hessian blocks H11, H12, H22 (D x D, D <= 15), gradient
// g1, g2 and the error terms.  D is a runtime value; storage is 15-wide.
struct PPBlocks {
  DM<15, 15> h11, h12, h22;
  double g1[15], g2[15];
  double err_build;  // contribution to *_error_ as BuildProblem computes it
};

// unary (BundleAdjuster.cpp:1434-1482): r = log_decoupled(T_wp, T_prior),
// J = dlog_decoupled_dx; information = cov_inv * scale (scale carries the compounded
// Huber weights, :1469).  mahalanobis = r^T (cov_inv*scale) r.
BA_HD void unary_residual(const Tq& t_wp, const Tq& t_prior, int use_rotation, double* r6,
                          DM<6, 6>* J) {
  log_decoupled(t_wp, t_prior, r6);
  *J = dlog_decoupled_dx(t_wp, t_prior);
  if (!use_rotation)
    for (int i = 3; i < 6; ++i) {
      r6[i] = 0.0;
      for (int c = 0; c < 6; ++c) (*J)(i, c) = 0.0;
    }
}
BA_HD double quad6(const double* info36, const double* r6) {
  double s = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) s += r6[i] * info36[i * 6 + j] * r6[j];
  return s;
}
BA_HD void unary_blocks(const double* r6, const DM<6, 6>& J, const double* cov_inv36, double scale,
                        PPBlocks* o) {
  DM<6, 6> info;
  for (int i = 0; i < 36; ++i) info.m[i] = cov_inv36[i] * scale;
  const DM<6, 6> JtI = mTm(J, info);
  const DM<6, 6> H = mm(JtI, J);
  o->h11.zero(); o->h12.zero(); o->h22.zero();
  set_block(o->h11, 0, 0, H);
  for (int i = 0; i < 15; ++i) { o->g1[i] = 0.0; o->g2[i] = 0.0; }
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += JtI(i, k) * r6[k];
    o->g1[i] = s;
  }
  o->err_build = quad6(info.m, r6);
}

// binary (BundleAdjuster.cpp:1395-1428, insertion :1647-1669)
BA_HD void binary_blocks(const Tq& t_w1, const Tq& t_w2, const Tq& t_12, const double* cov_inv36,
                         const double* cov_inv_sqrt36, double weight, int use_rotation,
                         PPBlocks* o, double* eval_err) {
  const Tq t_1w = tq_inv(t_w1);
  const Tq t12 = tq_mul(t_1w, t_w2);
  double raw[6];
  log_decoupled(t12, t_12, raw);
  DM<6, 6> S;
  for (int i = 0; i < 36; ++i) S.m[i] = cov_inv_sqrt36[i];
  double rw[6];  // whitened residual
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += S(i, k) * raw[k];
    rw[i] = s;
  }
  const DM<6, 7> dl = dLog_decoupled_dt1(t12, t_12);
  DM<6, 6> dz1 = mm(mm(dl, dt1_t2_dt1(t_1w, t_w2)), dinv_exp_decoupled_dx(t_w1));
  DM<6, 6> dz2 = mm(mm(dl, dt1_t2_dt2(t_1w)), dexp_decoupled_dx(t_w2));
  if (!use_rotation)
    for (int i = 3; i < 6; ++i) {
      rw[i] = 0.0;
      for (int c = 0; c < 6; ++c) { dz1(i, c) = 0.0; dz2(i, c) = 0.0; }
    }
  // J = S dz ; J^T = dz^T S weight (:1662-1668)
  const DM<6, 6> j1 = mm(S, dz1), j2 = mm(S, dz2);
  DM<6, 6> jt1 = mTm(dz1, S), jt2 = mTm(dz2, S);
  for (int i = 0; i < 36; ++i) { jt1.m[i] *= weight; jt2.m[i] *= weight; }
  o->h11.zero(); o->h12.zero(); o->h22.zero();
  set_block(o->h11, 0, 0, mm(jt1, j1));
  set_block(o->h12, 0, 0, mm(jt1, j2));
  set_block(o->h22, 0, 0, mm(jt2, j2));
  for (int i = 0; i < 15; ++i) { o->g1[i] = 0.0; o->g2[i] = 0.0; }
  for (int i = 0; i < 6; ++i) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < 6; ++k) { s1 += jt1(i, k) * rw[k]; s2 += jt2(i, k) * rw[k]; }
    o->g1[i] = s1; o->g2[i] = s2;
  }
  // :1425-1427: the whitened residual weighted by cov_inv once more, times weight
  o->err_build = quad6(cov_inv36, rw) * weight;
  if (eval_err) {  // EvaluateResiduals (:207-223): |log_decoupled|^2 * weight
    double s = 0.0;
    for (int i = 0; i < 6; ++i) {
      const double v = (!use_rotation && i >= 3) ? 0.0 : raw[i];
      s += v * v;
    }
    *eval_err = s * weight;
  }
}

// ---- IMU pre-integration (Types.h:324-738) ------------------------------------------------
struct ImuState { double t[3]; double q[4]; double v[3]; };  // q is NOT renormalised (Types.h:336-339)

// Types.h:324-373
BA_HD ImuState integrate_pose(const ImuState& s, const double* k9, double dt, DM<10, 9>* dy_dk,
                              DM<4, 4>* dy_dy) {
  ImuState y = s;
  double qe[4];
  const V3 w = v3(k9[3] * dt, k9[4] * dt, k9[5] * dt);
  so3_exp(w, qe);
  for (int i = 0; i < 3; ++i) { y.t[i] += k9[i] * dt; y.v[i] += k9[6 + i] * dt; }
  quat_mul(qe, s.q, y.q);
  if (dy_dk) {
    dy_dk->zero();
    for (int i = 0; i < 3; ++i) { (*dy_dk)(i, i) = dt; (*dy_dk)(7 + i, 6 + i) = dt; }
    const DM<4, 3> dq = mm(qR(s.q), dq_exp_dw(w));
    set_block(*dy_dk, 3, 3, dq, dt);
  }
  if (dy_dy) *dy_dy = qL(qe);
  return y;
}
// Types.h:376-416
BA_HD void pose_derivative(const ImuState& s, const double* g, const double* z0, const double* z1,
                           const double* bg, const double* ba, double dt, double* k9,
                           DM<9, 6>* dk_db, DM<9, 10>* dk_dx) {
  const double alpha = (z1[6] - (z0[6] + dt)) / (z1[6] - z0[6]);
  V3 zg, za;
  zg = v3(z0[0] * alpha + z1[0] * (1.0 - alpha), z0[1] * alpha + z1[1] * (1.0 - alpha),
          z0[2] * alpha + z1[2] * (1.0 - alpha));
  za = v3(z0[3] * alpha + z1[3] * (1.0 - alpha), z0[4] * alpha + z1[4] * (1.0 - alpha),
          z0[5] * alpha + z1[5] * (1.0 - alpha));
  const M3 R = quat_to_rot(s.q[0], s.q[1], s.q[2], s.q[3]);  // Adj() and matrix() of the raw quaternion
  const V3 bgv = v3(bg[0], bg[1], bg[2]), bav = v3(ba[0], ba[1], ba[2]);
  const V3 wv = mul(R, zg + bgv);
  const V3 av = qrot(s.q, za + bav);
  k9[0] = s.v[0]; k9[1] = s.v[1]; k9[2] = s.v[2];
  k9[3] = wv.x; k9[4] = wv.y; k9[5] = wv.z;
  k9[6] = av.x - g[0]; k9[7] = av.y - g[1]; k9[8] = av.z - g[2];
  if (dk_db) {
    dk_db->zero();
    set_block(*dk_db, 3, 0, dm3(R));
    set_block(*dk_db, 6, 3, dm3(R));
  }
  if (dk_dx) {
    dk_dx->zero();
    (*dk_dx)(0, 7) = (*dk_dx)(1, 8) = (*dk_dx)(2, 9) = 1.0;
    set_block(*dk_dx, 3, 3, madd(dqx_dq(s.q, zg), dqx_dq(s.q, bgv)));
    set_block(*dk_dx, 6, 3, madd(dqx_dq(s.q, za), dqx_dq(s.q, bav)));
  }
}
BA_HD void add_ident(DM<10, 10>& m, const DM<4, 4>& dy_dy) {  // Types.h:488-490
  for (int i = 0; i < 3; ++i) { m(i, i) += 1.0; m(7 + i, 7 + i) += 1.0; }
  add_block(m, 3, 3, dy_dy);
}
// C <- F C F^T + G R G^T with the Jacobians of one integration step (Types.h:617-640)
BA_HD void imu_cov_update(const DM<10, 6>& dy_db, const DM<10, 10>& dy_dy0, DM<10, 10>* cov, const double* r6,
                          const WaveCtx* w = nullptr) {
  DM<10, 6> GR = dy_db;
  for (int r = 0; r < 10; ++r)
    for (int c = 0; c < 6; ++c) GR(r, c) *= r6[c];
  const DM<10, 10> prop = mmT(mm(dy_dy0, *cov, w), dy_dy0, w);
  *cov = madd(prop, mmT(GR, dy_db, w));
}
// Products with the two structured Jacobians of an RK4 stage — same sums in the same order as mm()
// minus the terms that are exactly zero by construction:
//   dy_dk (integrate_pose): rows 0-2 dt at (i, i); rows 3-6 the 4x3 block at columns 3-5; rows 7-9 dt at (7+i, 6+i)
//   dk_dy (pose_derivative): rows 0-2 a 1 at (i, 7+i); rows 3-5 and 6-8 a 3x4 block at columns 3-6
template <int N>
BA_HD DM<10, N> mm_dy_dk(const DM<10, 9>& a, const DM<9, N>& x) {
  DM<10, N> o;
  for (int c = 0; c < N; ++c) {
    for (int r = 0; r < 3; ++r) o(r, c) = 0.0 + a(r, r) * x(r, c);
    for (int r = 3; r < 7; ++r) {
      double s = 0.0;
      for (int k = 3; k < 6; ++k) s += a(r, k) * x(k, c);
      o(r, c) = s;
    }
    for (int r = 7; r < 10; ++r) o(r, c) = 0.0 + a(r, r - 1) * x(r - 1, c);
  }
  return o;
}
template <int N>
BA_HD DM<9, N> mm_dk_dy(const DM<9, 10>& a, const DM<10, N>& y) {
  DM<9, N> o;
  for (int c = 0; c < N; ++c) {
    for (int r = 0; r < 3; ++r) o(r, c) = 0.0 + a(r, 7 + r) * y(7 + r, c);
    for (int r = 3; r < 9; ++r) {
      double s = 0.0;
      for (int k = 3; k < 7; ++k) s += a(r, k) * y(k, c);
      o(r, c) = s;
    }
  }
  return o;
}
#if defined(__HIP_DEVICE_COMPILE__)
// Wave mode of the per-residual pass: the two accumulations of one integration step with everything
// resident in LDS (no private-memory copies of the 10x10 matrices): layout in w->lds
//   cov[100] | dpose_db[60] | t1[100] | step dy_db[60] | step dy_dy0[100]
// Element by element the expressions of imu_cov_update and of `madd(dy_db, mm(dy_dy, dpose_db))`.
BA_HD void imu_accumulate_wave(const double* __restrict__ st, bool update_cov, const double* r6, const WaveCtx* w) {
  double* cov = w->lds;
  double* dp = cov + 100;
  double* t1 = dp + 60;
  double* sdb = t1 + 100;
  double* sdy = sdb + 60;
  const int lane = w->lane;
  for (int e = lane; e < 160; e += 64) sdb[e] = st[e];  // (sdy follows sdb)
  __syncthreads();
  if (update_cov) {
    for (int e = lane; e < 100; e += 64) {
      const int r = e / 10, c = e - 10 * r;
      double s = 0.0;
      for (int k = 0; k < 10; ++k) s += sdy[r * 10 + k] * cov[k * 10 + c];
      t1[e] = s;
    }
    __syncthreads();
    for (int e = lane; e < 100; e += 64) {
      const int r = e / 10, c = e - 10 * r;
      double p = 0.0, g = 0.0;
      for (int k = 0; k < 10; ++k) p += t1[r * 10 + k] * sdy[c * 10 + k];
      for (int k = 0; k < 6; ++k) {
        const double gr = sdb[r * 6 + k] * r6[k];
        g += gr * sdb[c * 6 + k];
      }
      cov[e] = p + 1.0 * g;
    }
    __syncthreads();
  }
  for (int e = lane; e < 60; e += 64) {
    const int r = e / 6, c = e - 6 * r;
    double s = 0.0;
    for (int k = 0; k < 10; ++k) s += sdy[r * 10 + k] * dp[k * 6 + c];
    t1[e] = sdb[e] + 1.0 * s;
  }
  __syncthreads();
  for (int e = lane; e < 60; e += 64) dp[e] = t1[e];
  __syncthreads();
}
#endif
// Types.h:419-643, Jacobian branch, Euler covariance (C <- F C F^T + G R G^T)
BA_HD ImuState integrate_imu(const ImuState& s, const double* z0, const double* z1, const double* bg,
                             const double* ba, const double* g, bool jac, DM<10, 6>* dy_db,
                             DM<10, 10>* dy_dy0, DM<10, 10>* cov, const double* r6, const WaveCtx* w = nullptr) {
  const double dt = z1[6] - z0[6];
  if (dt == 0) {
    if (jac) { dy_db->zero(); dy_dy0->identity(); }
    return s;
  }
  double k1[9], k2[9], k3[9], k4[9], k[9];
  if (!jac) {
    pose_derivative(s, g, z0, z1, bg, ba, 0, k1, nullptr, nullptr);
    const ImuState y1 = integrate_pose(s, k1, dt * 0.5, nullptr, nullptr);
    pose_derivative(y1, g, z0, z1, bg, ba, dt / 2, k2, nullptr, nullptr);
    const ImuState y2 = integrate_pose(s, k2, dt * 0.5, nullptr, nullptr);
    pose_derivative(y2, g, z0, z1, bg, ba, dt / 2, k3, nullptr, nullptr);
    const ImuState y3 = integrate_pose(s, k3, dt, nullptr, nullptr);
    pose_derivative(y3, g, z0, z1, bg, ba, dt, k4, nullptr, nullptr);
    for (int i = 0; i < 9; ++i) k[i] = k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i];
    return integrate_pose(s, k, dt / 6.0, nullptr, nullptr);
  }
  DM<9, 6> dk_db;
  DM<9, 10> dk_dy;
  DM<10, 9> dy_dk;
  DM<4, 4> dyy;
  pose_derivative(s, g, z0, z1, bg, ba, 0, k1, &dk_db, &dk_dy);
  const DM<9, 6> dk1_db = dk_db;
  const DM<9, 10> dk1_dy = dk_dy;
  const ImuState y1 = integrate_pose(s, k1, dt * 0.5, &dy_dk, &dyy);
  *dy_db = mm_dy_dk(dy_dk, dk1_db);
  *dy_dy0 = mm_dy_dk(dy_dk, dk1_dy);
  add_ident(*dy_dy0, dyy);

  pose_derivative(y1, g, z0, z1, bg, ba, dt / 2, k2, &dk_db, &dk_dy);
  const DM<9, 6> dk2_db = madd(dk_db, mm_dk_dy(dk_dy, *dy_db));
  const DM<9, 10> dk2_dy = mm_dk_dy(dk_dy, *dy_dy0);
  const ImuState y2 = integrate_pose(s, k2, dt * 0.5, &dy_dk, &dyy);
  *dy_db = mm_dy_dk(dy_dk, dk2_db);
  *dy_dy0 = mm_dy_dk(dy_dk, dk2_dy);
  add_ident(*dy_dy0, dyy);

  pose_derivative(y2, g, z0, z1, bg, ba, dt / 2, k3, &dk_db, &dk_dy);
  const DM<9, 6> dk3_db = madd(dk_db, mm_dk_dy(dk_dy, *dy_db));
  const DM<9, 10> dk3_dy = mm_dk_dy(dk_dy, *dy_dy0);
  const ImuState y3 = integrate_pose(s, k3, dt, &dy_dk, &dyy);
  *dy_db = mm_dy_dk(dy_dk, dk3_db);
  *dy_dy0 = mm_dy_dk(dy_dk, dk3_dy);
  add_ident(*dy_dy0, dyy);

  pose_derivative(y3, g, z0, z1, bg, ba, dt, k4, &dk_db, &dk_dy);
  const DM<9, 6> dk4_db = madd(dk_db, mm_dk_dy(dk_dy, *dy_db));
  const DM<9, 10> dk4_dy = mm_dk_dy(dk_dy, *dy_dy0);

  for (int i = 0; i < 9; ++i) k[i] = k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i];
  DM<9, 6> dkt_db;
  DM<9, 10> dkt_dy;
  for (int i = 0; i < 54; ++i) dkt_db.m[i] = dk1_db.m[i] + 2 * dk2_db.m[i] + 2 * dk3_db.m[i] + dk4_db.m[i];
  for (int i = 0; i < 90; ++i) dkt_dy.m[i] = dk1_dy.m[i] + 2 * dk2_dy.m[i] + 2 * dk3_dy.m[i] + dk4_dy.m[i];
  const ImuState res = integrate_pose(s, k, dt / 6.0, &dy_dk, &dyy);
  *dy_db = mm_dy_dk(dy_dk, dkt_db);
  *dy_dy0 = mm_dy_dk(dy_dk, dkt_dy);
  add_ident(*dy_dy0, dyy);
  if (cov) imu_cov_update(*dy_db, *dy_dy0, cov, r6, w);
  return res;
}

// Inertial residual with Jacobians (parallel_algos.h:178-358).  pose state rows are
// [t(3) q(4) v(3) b(6)].  RS = residual size = pose dim (9 or 15).
struct ImuOut {
  double r[15];
  DM<15, 15> dz1, dz2, cov_inv;
};
// frozen_in / frozen_out (Options::calculate_inertial_covariance_once, parallel_algos.h:189-205):
// 160 doubles = the integration covariance (10x10) and the bias Jacobian of the integration
// (10x6).  frozen_in: integrate without Jacobians and take both from there; frozen_out: store
// what this call computed.
// steps (optional): the Jacobians of every integration step, computed beforehand (imu_step_jacobians:
// 160 doubles per sample, [dy_db 10x6 | dy_dy0 10x10], sample k holds the step k-1 -> k) — the state is
// then integrated without Jacobians and only the two accumulations run here, in the same order and
// with the same arithmetic as the fused form.
BA_HD void imu_residual(const double* p1, const double* p2, const double* meas, int nmeas,
                        const double* g, const double* r6, const double* rb6, int RS, bool jac,
                        ImuOut* o, const double* frozen_in = nullptr, double* frozen_out = nullptr,
                        const double* steps = nullptr, const WaveCtx* w = nullptr) {
  ImuState s;
  for (int i = 0; i < 3; ++i) { s.t[i] = p1[i]; s.v[i] = p1[7 + i]; }
  for (int i = 0; i < 4; ++i) s.q[i] = p1[3 + i];
  const double* bg = p1 + 10;
  const double* ba = p1 + 13;
  DM<10, 6> dpose_db, dy_db;
  DM<10, 10> dy_dy, cov;
  dpose_db.zero();
  cov.zero();
  const bool ijac = jac && !frozen_in;
#if defined(__HIP_DEVICE_COMPILE__)
  const bool lds_acc = ijac && steps && w;
  if (lds_acc) {
    for (int e = w->lane; e < 160; e += 64) w->lds[e] = 0.0;  // cov | dpose_db
    __syncthreads();
  }
#endif
  for (int i = 1; i < nmeas; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (lds_acc) {
      const double* z0 = meas + 7 * (i - 1);
      const double* z1 = meas + 7 * i;
      s = integrate_imu(s, z0, z1, bg, ba, g, false, nullptr, nullptr, nullptr, r6);
      imu_accumulate_wave(steps + (size_t)160 * i, z1[6] - z0[6] != 0, r6, w);
      continue;
    }
#endif
    if (ijac && steps) {
      const double* z0 = meas + 7 * (i - 1);
      const double* z1 = meas + 7 * i;
      s = integrate_imu(s, z0, z1, bg, ba, g, false, nullptr, nullptr, nullptr, r6);
      const double* st = steps + (size_t)160 * i;
      for (int k = 0; k < 60; ++k) dy_db.m[k] = st[k];
      for (int k = 0; k < 100; ++k) dy_dy.m[k] = st[60 + k];
      if (z1[6] - z0[6] != 0) imu_cov_update(dy_db, dy_dy, &cov, r6, w);  // (a zero step leaves the covariance alone)
    } else {
      s = integrate_imu(s, meas + 7 * (i - 1), meas + 7 * i, bg, ba, g, ijac, &dy_db, &dy_dy, &cov, r6);
    }
    if (ijac) dpose_db = madd(dy_db, mm(dy_dy, dpose_db, w));  // Types.h:712-714
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (lds_acc) {
    for (int k = 0; k < 100; ++k) cov.m[k] = w->lds[k];
    for (int k = 0; k < 60; ++k) dpose_db.m[k] = w->lds[100 + k];
    __syncthreads();
  }
#endif
  if (jac && frozen_in) {
    for (int k = 0; k < 100; ++k) cov.m[k] = frozen_in[k];
    for (int k = 0; k < 60; ++k) dpose_db.m[k] = frozen_in[100 + k];
  }
  if (jac && frozen_out) {
    for (int k = 0; k < 100; ++k) frozen_out[k] = cov.m[k];
    for (int k = 0; k < 60; ++k) frozen_out[100 + k] = dpose_db.m[k];
  }
  const Tq t_int = {{s.t[0], s.t[1], s.t[2]}, {s.q[0], s.q[1], s.q[2], s.q[3]}};
  const Tq t_w1 = tq_from7(p1), t_w2 = tq_from7(p2);
  for (int i = 0; i < 15; ++i) o->r[i] = 0.0;
  log_decoupled(t_int, t_w2, o->r);
  for (int i = 0; i < 3; ++i) o->r[6 + i] = s.v[i] - p2[7 + i];
  if (RS >= 15)
    for (int i = 0; i < 6; ++i) o->r[9 + i] = p1[10 + i] - p2[10 + i];
  if (!jac) return;
  const double total_dt = meas[7 * (nmeas - 1) + 6] - meas[6];
  // :217-227
  Tq t_12_0 = t_int;
  for (int i = 0; i < 3; ++i)
    t_12_0.t[i] -= (-g[i] * 0.5 * total_dt * total_dt + p1[7 + i] * total_dt);
  t_12_0 = tq_mul(tq_inv(t_w1), t_12_0);
  V3 v_12_0 = v3(s.v[0] - p1[7] + g[0] * total_dt, s.v[1] - p1[8] + g[1] * total_dt,
                 s.v[2] - p1[9] + g[2] * total_dt);
  {
    const double qi[4] = {-t_w1.q[0], -t_w1.q[1], -t_w1.q[2], t_w1.q[3]};
    v_12_0 = qrot(qi, v_12_0);
  }
  o->dz1.zero(); o->dz2.zero();
  for (int i = 0; i < 3; ++i) o->dz1(i, 6 + i) = total_dt;  // :238-239
  const M3 R1 = quat_to_rot(t_w1.q[0], t_w1.q[1], t_w1.q[2], t_w1.q[3]);
  for (int ii = 0; ii < 3; ++ii) {  // :240-244  R_w1 G_ii v_12_0 = R_w1 (e_ii x v)
    V3 e = v3(ii == 0, ii == 1, ii == 2);
    const V3 col = mul(R1, cross(e, v_12_0));
    o->dz1(6, 3 + ii) = col.x; o->dz1(7, 3 + ii) = col.y; o->dz1(8, 3 + ii) = col.z;
  }
  for (int i = 0; i < 3; ++i) o->dz1(6 + i, 6 + i) = 1.0;  // :247-248
  const DM<6, 7> dlog1 = dLog_decoupled_dt1(t_int, t_w2);
  set_block(o->dz1, 0, 0, mm(mm(dlog1, dt1_t2_dt1(t_w1, t_12_0)), dexp_decoupled_dx(t_w1)));  // :251-254
  set_block(o->dz2, 0, 0, mm(dlog_decoupled_dt2(t_int, t_w2), dexp_decoupled_dx(t_w2)));       // :258-260
  for (int i = 0; i < 3; ++i) o->dz2(6 + i, 6 + i) = -1.0;                                       // :264-265
  // covariance (:273-307)
  DM<9, 10> A;
  A.zero();
  set_block(A, 0, 0, dlog1);
  A(6, 7) = A(7, 8) = A(8, 9) = 1.0;
  const DM<9, 9> c9 = mmT(mm(A, cov, w), A, w);
  o->cov_inv.zero();
  for (int i = 0; i < 15; ++i) o->cov_inv(i, i) = 1.0;
  if (RS >= 15)
    for (int i = 0; i < 6; ++i) o->cov_inv(9 + i, 9 + i) = rb6[i] * total_dt;
  set_block(o->cov_inv, 0, 0, c9);
  invert_leading(o->cov_inv, RS, w);
  for (int i = RS; i < 15; ++i)
    for (int c = 0; c < 15; ++c) { o->cov_inv(i, c) = 0.0; o->cov_inv(c, i) = 0.0; }
  if (RS >= 15) {  // :313-337
    DM<15, 6> dz_db;
    dz_db.zero();
    set_block(dz_db, 0, 0, mm(dlog1, get_block<7, 6>(dpose_db, 0, 0)));
    set_block(dz_db, 6, 0, get_block<3, 6>(dpose_db, 7, 0));
    for (int i = 0; i < 6; ++i) dz_db(9 + i, i) = 1.0;
    set_block(o->dz1, 0, 9, dz_db);
    for (int i = 0; i < 6; ++i) o->dz2(9 + i, 9 + i) = -1.0;
  }
}
// The Jacobians of ONE integration step of a residual, sample k-1 -> k (k >= 1): the states up to
// sample k-1 are re-integrated without Jacobians (cheap), then the step with them.  One lane per
// (residual, sample): the steps of a residual are independent given the states, so the expensive part
// of the pre-integration — 80 % of its multiply-adds — runs in parallel over the samples.
BA_HD void imu_step_jacobians(const double* p1, const double* meas, int k, const double* g, double* out160,
                              const WaveCtx* w = nullptr) {
  ImuState s;
  for (int i = 0; i < 3; ++i) { s.t[i] = p1[i]; s.v[i] = p1[7 + i]; }
  for (int i = 0; i < 4; ++i) s.q[i] = p1[3 + i];
  const double* bg = p1 + 10;
  const double* ba = p1 + 13;
  for (int i = 1; i < k; ++i)
    s = integrate_imu(s, meas + 7 * (i - 1), meas + 7 * i, bg, ba, g, false, nullptr, nullptr, nullptr, nullptr);
  DM<10, 6> dy_db;
  DM<10, 10> dy_dy;
  (void)integrate_imu(s, meas + 7 * (k - 1), meas + 7 * k, bg, ba, g, true, &dy_db, &dy_dy, nullptr, nullptr, w);
  const int l0 = w ? w->lane : 0, ls = w ? 64 : 1;
  for (int i = l0; i < 60; i += ls) out160[i] = dy_db.m[i];
  for (int i = l0; i < 100; i += ls) out160[60 + i] = dy_dy.m[i];
}

#if defined(__HIPCC__)
// The step Jacobians of sample k with ONE WAVEFRONT per sample and the whole RK4 Jacobian chain resident in
// LDS (w->lds, 986 doubles): the four stage Jacobians dkN_db / dkN_dy, the running dy_db / dy_dy0 and the
// small constructed blocks (dk_db, dk_dy of pose_derivative; dy_dk, the 4x4 quaternion block of
// integrate_pose).  Every lane runs the states and the closed-form blocks (cheap, uniform), copies its share
// of them to LDS, and owns one or two elements of each structured product — the expressions of mm_dy_dk,
// mm_dk_dy, madd and add_ident of integrate_imu's Jacobian branch, element by element in the same order.
// The scalar form keeps ~700 doubles of these matrices in dynamically indexed private memory per lane.
__device__ __forceinline__ void imu_step_jacobians_wave(const double* p1, const double* meas, int k, const double* g,
                                                        double* __restrict__ out160, const WaveCtx* w) {
  constexpr int L_DKB = 0, L_DKY = 54, L_DYK = 144, L_DYY = 234, L_YB = 250, L_YY = 310, L_KB = 410, L_KY = 626;
  double* L = w->lds;
  const int lane = w->lane;
  ImuState s;
  for (int i = 0; i < 3; ++i) { s.t[i] = p1[i]; s.v[i] = p1[7 + i]; }
  for (int i = 0; i < 4; ++i) s.q[i] = p1[3 + i];
  const double* bg = p1 + 10;
  const double* ba = p1 + 13;
  for (int i = 1; i < k; ++i)
    s = integrate_imu(s, meas + 7 * (i - 1), meas + 7 * i, bg, ba, g, false, nullptr, nullptr, nullptr, nullptr);
  const double* z0 = meas + 7 * (k - 1);
  const double* z1 = meas + 7 * k;
  const double dt = z1[6] - z0[6];
  if (dt == 0) {  // integrate_imu: dy_db = 0, dy_dy0 = I
    for (int e = lane; e < 160; e += 64) out160[e] = (e >= 60 && (e - 60) % 11 == 0) ? 1.0 : 0.0;
    return;
  }
  auto put = [&](int off, const double* m, int n) { for (int e = lane; e < n; e += 64) L[off + e] = m[e]; };
  // o = mm_dy_dk(DYK, X) [+ identity blocks]: X at xo with N columns, result at oo
  auto prod_dy_dk = [&](int xo, int N, int oo, bool ident) {
    for (int e = lane; e < 10 * N; e += 64) {
      const int r = e / N, c = e - r * N;
      double v;
      if (r < 3) v = 0.0 + L[L_DYK + r * 9 + r] * L[xo + r * N + c];
      else if (r < 7) {
        v = 0.0;
        for (int kk = 3; kk < 6; ++kk) v += L[L_DYK + r * 9 + kk] * L[xo + kk * N + c];
      } else v = 0.0 + L[L_DYK + r * 9 + (r - 1)] * L[xo + (r - 1) * N + c];
      if (ident) {  // add_ident (Types.h:488-490)
        if ((r < 3 || r >= 7) && r == c) v += 1.0;
        if (r >= 3 && r < 7 && c >= 3 && c < 7) v += 1.0 * L[L_DYY + (r - 3) * 4 + (c - 3)];
      }
      L[oo + e] = v;
    }
  };
  // o = [DKB +] mm_dk_dy(DKY, Y): Y at yo with N columns, result at oo
  auto prod_dk_dy = [&](int yo, int N, int oo, bool add_dkb) {
    for (int e = lane; e < 9 * N; e += 64) {
      const int r = e / N, c = e - r * N;
      double v;
      if (r < 3) v = 0.0 + L[L_DKY + r * 10 + 7 + r] * L[yo + (7 + r) * N + c];
      else {
        v = 0.0;
        for (int kk = 3; kk < 7; ++kk) v += L[L_DKY + r * 10 + kk] * L[yo + kk * N + c];
      }
      L[oo + e] = add_dkb ? L[L_DKB + e] + 1.0 * v : v;
    }
  };
  double k1[9], k2[9], k3[9], k4[9], kt[9];
  DM<9, 6> dk_db;
  DM<9, 10> dk_dy;
  DM<10, 9> dy_dk;
  DM<4, 4> dyy;
  // stage 1
  pose_derivative(s, g, z0, z1, bg, ba, 0, k1, &dk_db, &dk_dy);
  const ImuState y1 = integrate_pose(s, k1, dt * 0.5, &dy_dk, &dyy);
  put(L_KB, dk_db.m, 54); put(L_KY, dk_dy.m, 90); put(L_DYK, dy_dk.m, 90); put(L_DYY, dyy.m, 16);
  __syncthreads();
  prod_dy_dk(L_KB, 6, L_YB, false); prod_dy_dk(L_KY, 10, L_YY, true);
  __syncthreads();
  // stages 2 .. 4: dkN = dk + dk_dy . (dy of the previous stage); then the stage's own dy
  ImuState yprev = y1;
  for (int stg = 1; stg < 4; ++stg) {
    double* kk = stg == 1 ? k2 : stg == 2 ? k3 : k4;
    pose_derivative(yprev, g, z0, z1, bg, ba, stg == 3 ? dt : dt / 2, kk, &dk_db, &dk_dy);
    put(L_DKB, dk_db.m, 54); put(L_DKY, dk_dy.m, 90);
    __syncthreads();
    prod_dk_dy(L_YB, 6, L_KB + stg * 54, true); prod_dk_dy(L_YY, 10, L_KY + stg * 90, false);
    if (stg == 3) break;
    yprev = integrate_pose(s, kk, stg == 1 ? dt * 0.5 : dt, &dy_dk, &dyy);
    __syncthreads();  // the products above read YB / YY and wrote KB / KY; DYK / DYY are free
    put(L_DYK, dy_dk.m, 90); put(L_DYY, dyy.m, 16);
    __syncthreads();
    prod_dy_dk(L_KB + stg * 54, 6, L_YB, false); prod_dy_dk(L_KY + stg * 90, 10, L_YY, true);
    __syncthreads();
  }
  __syncthreads();
  // total derivative k1 + 2 k2 + 2 k3 + k4 and its Jacobians (into DKB / DKY), the final step
  for (int i = 0; i < 9; ++i) kt[i] = k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i];
  for (int e = lane; e < 54; e += 64)
    L[L_DKB + e] = L[L_KB + e] + 2 * L[L_KB + 54 + e] + 2 * L[L_KB + 108 + e] + L[L_KB + 162 + e];
  for (int e = lane; e < 90; e += 64)
    L[L_DKY + e] = L[L_KY + e] + 2 * L[L_KY + 90 + e] + 2 * L[L_KY + 180 + e] + L[L_KY + 270 + e];
  (void)integrate_pose(s, kt, dt / 6.0, &dy_dk, &dyy);
  put(L_DYK, dy_dk.m, 90); put(L_DYY, dyy.m, 16);
  __syncthreads();
  prod_dy_dk(L_DKB, 6, L_YB, false); prod_dy_dk(L_DKY, 10, L_YY, true);
  __syncthreads();
  for (int e = lane; e < 160; e += 64) out160[e] = L[L_YB + e];  // YB (60) | YY (100) are adjacent
  __syncthreads();
}
#endif

// blocks J^T S^-1 J, J^T S^-1 r with S^-1 = cov_inv * weight (BundleAdjuster.cpp:1526)
#if defined(__HIPCC__)
// Wave mode of imu_blocks: operands in LDS (info | dz1 | dz2 | j1t | j2t, 5 x 225 doubles), the three
// Hessian blocks and the two gradients go straight from the lanes to global memory.  Same sums in the
// same order as the scalar form.  Returns the error term.
__device__ __forceinline__ double imu_blocks_wave(const ImuOut& io, double weight, const WaveCtx* w, double* __restrict__ h675,
                             double* __restrict__ g30) {
  double* info = w->lds;
  double* z1 = info + 225;
  double* z2 = z1 + 225;
  double* j1 = z2 + 225;
  double* j2 = j1 + 225;
  const int lane = w->lane;
  for (int e = lane; e < 225; e += 64) { info[e] = io.cov_inv.m[e] * weight; z1[e] = io.dz1.m[e]; z2[e] = io.dz2.m[e]; }
  __syncthreads();
  for (int e = lane; e < 225; e += 64) {
    const int r = e / 15, c = e - 15 * r;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < 15; ++k) { s1 += z1[k * 15 + r] * info[k * 15 + c]; s2 += z2[k * 15 + r] * info[k * 15 + c]; }
    j1[e] = s1; j2[e] = s2;
  }
  __syncthreads();
  for (int e = lane; e < 225; e += 64) {
    const int r = e / 15, c = e - 15 * r;
    double a = 0.0, b = 0.0, d = 0.0;
    for (int k = 0; k < 15; ++k) {
      a += j1[r * 15 + k] * z1[k * 15 + c];
      b += j1[r * 15 + k] * z2[k * 15 + c];
      d += j2[r * 15 + k] * z2[k * 15 + c];
    }
    h675[e] = a; h675[225 + e] = b; h675[450 + e] = d;
  }
  double* q = z1;  // (reused after the barrier below)
  __syncthreads();
  if (lane < 15) {
    double s1 = 0.0, s2 = 0.0, qq = 0.0;
    for (int k = 0; k < 15; ++k) { s1 += j1[lane * 15 + k] * io.r[k]; s2 += j2[lane * 15 + k] * io.r[k]; qq += info[lane * 15 + k] * io.r[k]; }
    g30[lane] = s1; g30[15 + lane] = s2;
    q[lane] = qq;
  }
  __syncthreads();
  double e = 0.0;
  for (int i = 0; i < 15; ++i) e += io.r[i] * q[i];
  __syncthreads();
  return e;
}
#endif
BA_HD void imu_blocks(const ImuOut& io, double weight, PPBlocks* o, const WaveCtx* w = nullptr) {
  DM<15, 15> info = io.cov_inv;
  for (int i = 0; i < 225; ++i) info.m[i] *= weight;
  const DM<15, 15> j1t = mTm(io.dz1, info, w), j2t = mTm(io.dz2, info, w);
  o->h11 = mm(j1t, io.dz1, w);
  o->h12 = mm(j1t, io.dz2, w);
  o->h22 = mm(j2t, io.dz2, w);
  double e = 0.0;
  for (int i = 0; i < 15; ++i) {
    double s1 = 0.0, s2 = 0.0, q = 0.0;
    for (int k = 0; k < 15; ++k) { s1 += j1t(i, k) * io.r[k]; s2 += j2t(i, k) * io.r[k]; q += info(i, k) * io.r[k]; }
    o->g1[i] = s1; o->g2[i] = s2;
    e += io.r[i] * q;
  }
  o->err_build = e;
}

}  // namespace bad
