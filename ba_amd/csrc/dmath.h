// Device/host math shared by the gfx950 kernels of the projection path.
//
// Closed forms of the reference's Jacobian chains.  The reference multiplies dense
// 2x4 . 4x7 . 7x7 . 7x6 matrices that are mostly zeros
// (/root/reference/include/ba/parallel_algos.h:91-110 with the helpers of
// /root/reference/include/ba/Utils.h:295-312,451-694); with unit quaternions those
// products collapse algebraically to
//   dz_dx_meas = dpi . [ rho R_sw ,  -R_sv [R_pw (Xw - rho t_wp)]x ]
//   dz_dx_ref  =       [ -rho dpi R_sw_m ,  dpi R_a [Y]x ]          (LmDim == 1)
//   dz_dlm     = -dpi t (LmDim 1)   |   -dpi R_sw (LmDim 3)
// where dpi is the pinhole derivative at P, R_a = R_sw_m R_wp_ref, Y = R_vs_r ray +
// rho t_vs_r (derivation: DESIGN.md §Kernels).  BA_HD lets the same code be compiled
// for the host so tests can compare it with the oracle without a GPU.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BA_HD __host__ __device__ __forceinline__
#else
#define BA_HD inline
#endif

namespace bad {

struct V3 { double x, y, z; };

BA_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
BA_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
BA_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
BA_HD V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
BA_HD V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
BA_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// 3x3 row-major
struct M3 { double m[9]; };
BA_HD V3 mul(const M3& R, V3 v) {
  return v3(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z,
            R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z);
}
BA_HD V3 mulT(const M3& R, V3 v) {  // R^T v
  return v3(R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z,
            R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
            R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z);
}
BA_HD M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C.m[r * 3 + c] = A.m[r * 3] * B.m[c] + A.m[r * 3 + 1] * B.m[3 + c] + A.m[r * 3 + 2] * B.m[6 + c];
  return C;
}
BA_HD M3 transpose(const M3& A) {
  M3 T;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T.m[r * 3 + c] = A.m[c * 3 + r];
  return T;
}
// rotation matrix of a quaternion (x,y,z,w), Eigen's polynomial form
BA_HD M3 quat_to_rot(double x, double y, double z, double w) {
  M3 R;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
  R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
  return R;
}
// Hamilton product, (x,y,z,w) storage
BA_HD void quat_mul(const double* a, const double* b, double* o) {
  const double ox = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double oy = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double oz = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  const double ow = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = ox; o[1] = oy; o[2] = oz; o[3] = ow;
}
BA_HD void quat_normalize(double* q) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
// Sophus SO3::exp (small-angle series below 1e-10), then normalised as the SO3
// constructor does.
BA_HD void so3_exp(V3 w, double* q) {
  const double th2 = dot(w, w);
  const double th = sqrt(th2);
  double imag, real;
  if (th < 1e-10) {
    const double th4 = th2 * th2;
    imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    real = 1.0 - 0.5 * th2 + (1.0 / 384.0) * th4;
  } else {
    imag = sin(0.5 * th) / th;
    real = cos(0.5 * th);
  }
  q[0] = imag * w.x; q[1] = imag * w.y; q[2] = imag * w.z; q[3] = real;
  quat_normalize(q);
}
// Sophus SO3::log
BA_HD V3 so3_log(const double* q) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const double n = sqrt(n2);
  const double w = q[3];
  double f;
  if (n < 1e-10) {
    f = 2.0 / w - 2.0 * n2 / (w * w * w);
  } else if (fabs(w) < 1e-10) {
    f = (w > 0 ? M_PI : -M_PI) / n;
  } else {
    f = 2.0 * atan(n / w) / n;
  }
  return v3(f * q[0], f * q[1], f * q[2]);
}

// Rigid transform as rotation matrix + translation: y = R x + t
struct Rt { M3 R; V3 t; };
BA_HD Rt compose(const Rt& a, const Rt& b) {  // a * b
  Rt c;
  c.R = mul(a.R, b.R);
  c.t = mul(a.R, b.t) + a.t;
  return c;
}
BA_HD Rt inverse(const Rt& a) {
  Rt c;
  c.R = transpose(a.R);
  c.t = mulT(a.R, a.t) * -1.0;
  return c;
}

// Camera model: calibu::LinearCamera (model 0: fx, fy, u0, v0) or calibu::FovCamera (model 1: + w, the
// five parameters of the reference's CalibSize = 5 instantiations, BundleAdjuster.cpp:1816-1826).  Calibu
// is not in the reference tree: the FOV model is the published one (Devernay & Faugeras 2001,
// r_d = atan(2 r_u tan(w/2)) / w) applied as a factor on the normalised point, with its limits for a
// vanishing radius / vanishing w — the same restatement as oracle/outils.h, operation for operation.
// Kernels built for pinhole-only rigs construct Cam with model = 0 as a literal: the FOV branches fold away.
struct Cam { double fx, fy, u0, v0, w; int model; };
constexpr double kFovSmall = 1e-5;

// distortion factor f = r_d / r_u with df/dr, df/dw;  inverse factor g = r_u / r_d with dg/dr, dg/dw
BA_HD void fov_factor(double w, double r, double* f, double* df_dr, double* df_dw) {
  *f = 1.0; *df_dr = 0.0; *df_dw = 0.0;
  if (w * w <= kFovSmall) return;
  const double th = tan(0.5 * w), m = 2.0 * th, dm = 1.0 + th * th;
  if (r * r < kFovSmall) {
    *f = m / w;
    *df_dw = dm / w - m / (w * w);
    return;
  }
  const double at = atan(r * m), den = 1.0 + r * r * m * m;
  *f = at / (r * w);
  *df_dr = m / (den * r * w) - at / (r * r * w);
  *df_dw = dm / (den * w) - at / (r * w * w);
}
BA_HD void fov_factor_inv(double w, double rd, double* g, double* dg_dr, double* dg_dw) {
  *g = 1.0; *dg_dr = 0.0; *dg_dw = 0.0;
  if (w * w <= kFovSmall) return;
  const double th = tan(0.5 * w), m = 2.0 * th, dm = 1.0 + th * th;
  if (rd * rd < kFovSmall) {
    *g = w / m;
    *dg_dw = 1.0 / m - w * dm / (m * m);
    return;
  }
  const double tn = tan(rd * w), sec2 = 1.0 + tn * tn;
  *g = tn / (rd * m);
  *dg_dr = w * sec2 / (rd * m) - tn / (rd * rd * m);
  *dg_dw = sec2 / m - tn * dm / (rd * m * m);
}

// pi(P) and its 2x3 derivative rows (d0 = d u/dP, d1 = d v/dP)
BA_HD void project(const Cam& c, V3 P, double* u, double* v) {
  const double iz = 1.0 / P.z;
  if (c.model == 1) {
    const double px = P.x / P.z, py = P.y / P.z;
    double f, dr, dw;
    fov_factor(c.w, sqrt(px * px + py * py), &f, &dr, &dw);
    *u = c.fx * (f * px) + c.u0;
    *v = c.fy * (f * py) + c.v0;
    return;
  }
  *u = c.fx * P.x * iz + c.u0;
  *v = c.fy * P.y * iz + c.v0;
}
BA_HD void dproject(const Cam& c, V3 P, V3* d0, V3* d1) {
  const double iz = 1.0 / P.z;
  if (c.model == 1) {
    const double px = P.x * iz, py = P.y * iz, r = sqrt(px * px + py * py);
    double f, dr, dw;
    fov_factor(c.w, r, &f, &dr, &dw);
    const double k = r > 0.0 ? dr / r : 0.0;  // d (f p) / d p = f I + (df/dr) p p^T / r
    const double a00 = f + k * px * px, a01 = k * px * py, a11 = f + k * py * py;
    *d0 = v3(c.fx * a00 * iz, c.fx * a01 * iz, -c.fx * (a00 * px + a01 * py) * iz);
    *d1 = v3(c.fy * a01 * iz, c.fy * a11 * iz, -c.fy * (a01 * px + a11 * py) * iz);
    return;
  }
  *d0 = v3(c.fx * iz, 0.0, -c.fx * P.x * iz * iz);
  *d1 = v3(0.0, c.fy * iz, -c.fy * P.y * iz * iz);
}
// Unproject(pix): the ray with z = 1
BA_HD V3 unproject(const Cam& c, double u, double v) {
  double x = (u - c.u0) / c.fx, y = (v - c.v0) / c.fy;
  if (c.model == 1) {
    double g, dr, dw;
    fov_factor_inv(c.w, sqrt(x * x + y * y), &g, &dr, &dw);
    x *= g; y *= g;
  }
  return v3(x, y, 1.0);
}

// One projection residual with its Jacobians.
//   LM == 1: x = (ray, rho) in the reference sensor frame; LM == 3: x = homogeneous
//   world point.  Inputs: T_sw_m (sensor<-world of the measuring pose/camera), for
//   LM == 1 T_ws_r (world<-reference sensor); T_wp of the measuring pose and, for the
//   reference-pose block, T_wp_r and T_vs_r; T_sv_m = T_vs_m^-1.
// Outputs (row-major): r[2], jm[2][6], jr[2][6], jl[2][LM].
template <int LM>
struct ProjJac {
  double r[2];
  double jm[12];
  double jr[12];
  double jl[2 * (LM > 0 ? LM : 1)];
};

// residual only (BundleAdjuster.cpp:155-176, parallel_algos.h:59-64)
template <int LM>
BA_HD V3 proj_point(const Rt& t_sw_m, const Rt& t_ws_r, const double* x) {
  const V3 xv = v3(x[0], x[1], x[2]);
  if (LM == 1) {
    const Rt T = compose(t_sw_m, t_ws_r);
    return mul(T.R, xv) + T.t * x[3];
  }
  return mul(t_sw_m.R, xv) + t_sw_m.t * x[3];
}

template <int LM>
BA_HD void proj_jacobians(const Cam& cam, const double* z, const double* x,
                          const Rt& t_sw_m, const Rt& t_ws_r, const Rt& t_wp_m,
                          const Rt& t_sv_m, const Rt& t_wp_r, const Rt& t_vs_r,
                          bool same_pose, ProjJac<LM>* out) {
  const V3 xv = v3(x[0], x[1], x[2]);
  const double rho = x[3];
  Rt T = t_sw_m;
  if (LM == 1) T = compose(t_sw_m, t_ws_r);
  const V3 P = mul(T.R, xv) + T.t * rho;
  double u, v;
  project(cam, P, &u, &v);
  out->r[0] = z[0] - u;
  out->r[1] = z[1] - v;
  V3 d0, d1;
  dproject(cam, P, &d0, &d1);
  // landmark block: -dpi [R | t] columns (parallel_algos.h:76-84)
  if (LM == 1) {
    out->jl[0] = -dot(d0, T.t);
    out->jl[1] = -dot(d1, T.t);
  } else if (LM == 3) {
    const V3 a0 = mulT(T.R, d0), a1 = mulT(T.R, d1);  // rows of dpi R
    out->jl[0] = -a0.x; out->jl[1] = -a0.y; out->jl[2] = -a0.z;
    out->jl[3] = -a1.x; out->jl[4] = -a1.y; out->jl[5] = -a1.z;
  }
  if (same_pose) {  // parallel_algos.h:97-99,111-113
    for (int i = 0; i < 12; ++i) { out->jm[i] = 0.0; out->jr[i] = 0.0; }
    return;
  }
  // measuring pose block (parallel_algos.h:91-96)
  {
    // world point (scaled by rho for LM == 1): Xw = R_ws_r ray + rho t_ws_r
    V3 Xw = xv;
    if (LM == 1) Xw = mul(t_ws_r.R, xv) + t_ws_r.t * rho;
    // point in the vehicle frame of the measuring pose, R_pw (Xw - rho t_wp)
    const V3 Pp = mulT(t_wp_m.R, Xw - t_wp_m.t * rho);
    // translation columns: rho dpi R_sw
    const V3 a0 = mulT(t_sw_m.R, d0), a1 = mulT(t_sw_m.R, d1);
    out->jm[0] = rho * a0.x; out->jm[1] = rho * a0.y; out->jm[2] = rho * a0.z;
    out->jm[6] = rho * a1.x; out->jm[7] = rho * a1.y; out->jm[8] = rho * a1.z;
    // rotation columns: -dpi R_sv [Pp]x ; row^T [Pp]x = (row x ... ) -> -(b x Pp)... :
    // (b^T [p]x)_j = (p x b)_j * -1 = (b x p)_j  =>  -b^T [p]x = p x b
    const V3 b0 = mulT(t_sv_m.R, d0), b1 = mulT(t_sv_m.R, d1);  // rows of dpi R_sv
    const V3 c0 = cross(Pp, b0), c1 = cross(Pp, b1);
    out->jm[3] = c0.x; out->jm[4] = c0.y; out->jm[5] = c0.z;
    out->jm[9] = c1.x; out->jm[10] = c1.y; out->jm[11] = c1.z;
  }
  if (LM == 1) {
    // reference pose block (parallel_algos.h:104-110)
    const V3 a0 = mulT(t_sw_m.R, d0), a1 = mulT(t_sw_m.R, d1);  // rows of dpi R_sw_m
    out->jr[0] = -rho * a0.x; out->jr[1] = -rho * a0.y; out->jr[2] = -rho * a0.z;
    out->jr[6] = -rho * a1.x; out->jr[7] = -rho * a1.y; out->jr[8] = -rho * a1.z;
    // rows of dpi R_a, R_a = R_sw_m R_wp_r ; then (row^T [Y]x) = Y-cross: b^T [y]x = (b x y)... sign below
    const V3 e0 = mulT(t_wp_r.R, a0), e1 = mulT(t_wp_r.R, a1);
    const V3 Y = mul(t_vs_r.R, xv) + t_vs_r.t * rho;
    // b^T [y]x = -(y x b)^T ... using (b^T [y]x)_j = sum_i b_i eps_{i k j} ... = (b x y)_j
    const V3 f0 = cross(e0, Y), f1 = cross(e1, Y);
    out->jr[3] = f0.x; out->jr[4] = f0.y; out->jr[5] = f0.z;
    out->jr[9] = f1.x; out->jr[10] = f1.y; out->jr[11] = f1.z;
  } else {
    for (int i = 0; i < 12; ++i) out->jr[i] = 0.0;
  }
}

// Same residual and Jacobians from fewer inputs — what k_linearize uses (one thread per
// observation: every transform held in registers costs occupancy).  Identities used:
//   point in the measuring vehicle frame   R_pw (Xw - rho t_wp) = R_vs (P - rho t_sv),  P = sensor-frame point
//     =>  rotation columns of dz_dx_meas:   [.]x (R_vs d) = R_vs ((P - rho t_sv) x d)
//   point in the reference vehicle frame   R_vs_r ray + rho t_vs_r = R_pw_r (Xw - rho t_wp_r)
//     =>  rotation columns of dz_dx_ref:    R_pw_r (a x (Xw - rho t_wp_r)),  a = R_sw_m^T d
// so T_wp of the measuring pose and T_vs of the reference camera are not needed.
//   LM == 1: x = (ray, rho) in the reference sensor frame, t_ws_r / t_wp_r of the reference pose;
//   LM == 3: x = homogeneous world point, t_ws_r / t_wp_r unused.
//
// jk (CAL, LM == 1): the two rows of dz_dtvs (parallel_algos.h:120-131), the derivative w.r.t. the
// decoupled update of the camera mount, T_vs <- (R_vs exp(w), t_vs + dt), entering through both the
// measuring and the reference camera: with T = T_vs_m^-1 A T_vs_r, A = T_wp_m^-1 T_wp_r,
//   dP/d(dt) = rho R_vs_m^T (R_A - I),   dP/dw = [P]x - R_T [ray]x
//   =>  row = -(dP/d.)^T d :   translation  -rho (R_wp_r^T a - R_vs_m d),   a = R_sw_m^T d
//                              rotation      P x d - ray x (R_ws_r^T a)
// (identically zero when both poses coincide; `keep` makes that exact).
template <int LM, bool CAL = false>
BA_HD void proj_linearize(const Cam& cam, const double* z, const double* x, const Rt& t_sw_m, const M3& R_vs_m,
                          V3 t_sv_m, const Rt& t_ws_r, const Rt& t_wp_r, bool same_pose, ProjJac<LM>* out,
                          double* jk = nullptr) {
  const V3 xv = v3(x[0], x[1], x[2]);
  const double rho = x[3];
  V3 Xw = xv;  // world point, scaled by rho for LM == 1
  if (LM == 1) Xw = mul(t_ws_r.R, xv) + t_ws_r.t * rho;
  const V3 P = mul(t_sw_m.R, Xw) + t_sw_m.t * rho;
  double u, v;
  project(cam, P, &u, &v);
  out->r[0] = z[0] - u;
  out->r[1] = z[1] - v;
  V3 d0, d1;
  dproject(cam, P, &d0, &d1);
  const V3 a0 = mulT(t_sw_m.R, d0), a1 = mulT(t_sw_m.R, d1);  // rows of dpi R_sw_m
  if (LM == 1) {
    const V3 tt = mul(t_sw_m.R, t_ws_r.t) + t_sw_m.t;  // translation of T_sw_m T_ws_r
    out->jl[0] = -dot(d0, tt);
    out->jl[1] = -dot(d1, tt);
  } else if (LM == 3) {
    out->jl[0] = -a0.x; out->jl[1] = -a0.y; out->jl[2] = -a0.z;
    out->jl[3] = -a1.x; out->jl[4] = -a1.y; out->jl[5] = -a1.z;
  }
  // same_pose (parallel_algos.h:97-99,111-113): both pose blocks are zero — applied as a factor
  // (0 or 1) at the end rather than an early return, so that every output stays in registers
  const double keep = same_pose ? 0.0 : 1.0;
  const double rk = rho * keep;
  {
    out->jm[0] = rk * a0.x; out->jm[1] = rk * a0.y; out->jm[2] = rk * a0.z;
    out->jm[6] = rk * a1.x; out->jm[7] = rk * a1.y; out->jm[8] = rk * a1.z;
    const V3 q = (P - t_sv_m * rho) * keep;
    const V3 c0 = mul(R_vs_m, cross(q, d0)), c1 = mul(R_vs_m, cross(q, d1));
    out->jm[3] = c0.x; out->jm[4] = c0.y; out->jm[5] = c0.z;
    out->jm[9] = c1.x; out->jm[10] = c1.y; out->jm[11] = c1.z;
  }
  if (LM == 1) {
    out->jr[0] = -rk * a0.x; out->jr[1] = -rk * a0.y; out->jr[2] = -rk * a0.z;
    out->jr[6] = -rk * a1.x; out->jr[7] = -rk * a1.y; out->jr[8] = -rk * a1.z;
    const V3 y = (Xw - t_wp_r.t * rho) * keep;
    const V3 f0 = mulT(t_wp_r.R, cross(a0, y)), f1 = mulT(t_wp_r.R, cross(a1, y));
    out->jr[3] = f0.x; out->jr[4] = f0.y; out->jr[5] = f0.z;
    out->jr[9] = f1.x; out->jr[10] = f1.y; out->jr[11] = f1.z;
    if (CAL) {
      const V3 g0 = (mulT(t_wp_r.R, a0) - mul(R_vs_m, d0)) * (-rk), g1 = (mulT(t_wp_r.R, a1) - mul(R_vs_m, d1)) * (-rk);
      const V3 h0 = (cross(P, d0) - cross(xv, mulT(t_ws_r.R, a0))) * keep;
      const V3 h1 = (cross(P, d1) - cross(xv, mulT(t_ws_r.R, a1))) * keep;
      jk[0] = g0.x; jk[1] = g0.y; jk[2] = g0.z; jk[3] = h0.x; jk[4] = h0.y; jk[5] = h0.z;
      jk[6] = g1.x; jk[7] = g1.y; jk[8] = g1.z; jk[9] = h1.x; jk[10] = h1.y; jk[11] = h1.z;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) out->jr[i] = 0.0;
  }
}

// dz_dcam_params (CalibSize instantiations, LM == 1; parallel_algos.h:114-118):
//   -dTransfer_dparams(T_sw_m T_ws_r, z_ref, rho),  Transfer(T, pix, rho) = Project(R Unproject(pix) + rho t),
// over the camera parameters (pinhole: fx, fy, u0, v0; FOV camera: + w) — both the un-projection of the
// reference pixel and the projection depend on them.  The ray is Unproject(z_ref) (z = 1), NOT the landmark's x_s ray, paired
// with the landmark's rho as it stands: the reference's call, kept as written.
// jk: two rows of six (pinhole: columns 4, 5 zero; FOV camera: column 5 zero).
BA_HD void proj_intrinsics_rows(const Cam& cam, const double* z_ref, double rho, const Rt& t_sw_m, const Rt& t_ws_r,
                                double keep, double* jk) {
  const V3 ray = unproject(cam, z_ref[0], z_ref[1]);
  const V3 P = mul(t_sw_m.R, mul(t_ws_r.R, ray) + t_ws_r.t * rho) + t_sw_m.t * rho;
  V3 d0, d1;
  dproject(cam, P, &d0, &d1);
  // columns 0 and 1 of R_T = R_sw_m R_ws_r
  const V3 c0 = mul(t_sw_m.R, v3(t_ws_r.R.m[0], t_ws_r.R.m[3], t_ws_r.R.m[6]));
  const V3 c1 = mul(t_sw_m.R, v3(t_ws_r.R.m[1], t_ws_r.R.m[4], t_ws_r.R.m[7]));
  const double a00 = dot(d0, c0), a01 = dot(d0, c1), a10 = dot(d1, c0), a11 = dot(d1, c1);
  const double iz = 1.0 / P.z;
  if (cam.model == 1) {
    // FOV camera, five columns: dProject_dparams(P) + [dProject_dP R](:, 0:2) dUnproject_dparams(z_ref)
    const double dx = (z_ref[0] - cam.u0) / cam.fx, dy = (z_ref[1] - cam.v0) / cam.fy;
    const double rd = sqrt(dx * dx + dy * dy);
    double g, dg_dr, dg_dw;
    fov_factor_inv(cam.w, rd, &g, &dg_dr, &dg_dw);
    const double k = rd > 0.0 ? dg_dr / rd : 0.0;
    const double b00 = g + k * dx * dx, b01 = k * dx * dy, b11 = g + k * dy * dy;  // d ray.xy / d (dx, dy)
    const double du0[5] = {b00 * (-dx / cam.fx), b01 * (-dy / cam.fy), b00 * (-1.0 / cam.fx), b01 * (-1.0 / cam.fy), dx * dg_dw};
    const double du1[5] = {b01 * (-dx / cam.fx), b11 * (-dy / cam.fy), b01 * (-1.0 / cam.fx), b11 * (-1.0 / cam.fy), dy * dg_dw};
    const double px = P.x / P.z, py = P.y / P.z;
    double f, df_dr, df_dw;
    fov_factor(cam.w, sqrt(px * px + py * py), &f, &df_dr, &df_dw);
    const double dq0[5] = {f * px, 0.0, 1.0, 0.0, cam.fx * px * df_dw};
    const double dq1[5] = {0.0, f * py, 0.0, 1.0, cam.fy * py * df_dw};
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      jk[c] = -keep * (dq0[c] + (a00 * du0[c] + a01 * du1[c]));
      jk[6 + c] = -keep * (dq1[c] + (a10 * du0[c] + a11 * du1[c]));
    }
    jk[5] = 0.0; jk[11] = 0.0;
    return;
  }
  const double dx_dfx = -ray.x / cam.fx, dy_dfy = -ray.y / cam.fy;  // d ray / d fx = -(u - u0) / fx^2
  jk[0] = -keep * (a00 * dx_dfx + P.x * iz); jk[1] = -keep * (a01 * dy_dfy);
  jk[2] = -keep * (1.0 - a00 / cam.fx);      jk[3] = -keep * (-a01 / cam.fy);
  jk[4] = 0.0; jk[5] = 0.0;
  jk[6] = -keep * (a10 * dx_dfx);            jk[7] = -keep * (a11 * dy_dfy + P.y * iz);
  jk[8] = -keep * (-a10 / cam.fx);           jk[9] = -keep * (1.0 - a11 / cam.fy);
  jk[10] = 0.0; jk[11] = 0.0;
}

// ---- the reference's Jacobian chains with TWO extrinsics (DoTvs instantiations after a rejected step) ----
// After a rejected calibration step the reference's poses carry cached T_sw built with the T_vs BEFORE
// the step while the rig already holds the step (BundleAdjuster.cpp:72-83 vs :1060-1068, Types.h:61-70):
// until the next applied step its chains (parallel_algos.h:91-131) read cached transforms in some
// factors and the rig's T_vs in others.  The closed forms above assume one T_vs; this is the chain
// factor by factor — block products of the 4x7 . 7x7 . 7x6 matrices written out with quaternion
// products — with every factor taking the T_vs the reference takes.
//   d(R(q) v)/dq . dq : Utils.h:295-312 (the partial derivatives of the polynomial form, as written there)
BA_HD V3 dqx_dq_apply(const double* q, V3 p, const double* dq) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3], x = p.x, y = p.y, z = p.z;
  const double j00 = 2 * qy * y + 2 * qz * z, j10 = 2 * qy * x - 4 * qx * y - 2 * qw * z, j20 = 2 * qz * x + 2 * qw * y - 4 * qx * z;
  const double j01 = 2 * qx * y - 4 * qy * x + 2 * qw * z, j11 = 2 * qx * x + 2 * qz * z, j21 = 2 * qz * y - 2 * qw * x - 4 * qy * z;
  const double j02 = 2 * qx * z - 2 * qw * y - 4 * qz * x, j12 = 2 * qy * z + 2 * qw * x - 4 * qz * y, j22 = 2 * qy * y + 2 * qx * x;
  const double j03 = 2 * qy * z - 2 * qz * y, j13 = 2 * qz * x - 2 * qx * z, j23 = 2 * qx * y - 2 * qy * x;
  return v3(j00 * dq[0] + j01 * dq[1] + j02 * dq[2] + j03 * dq[3], j10 * dq[0] + j11 * dq[1] + j12 * dq[2] + j13 * dq[3],
            j20 * dq[0] + j21 * dq[1] + j22 * dq[2] + j23 * dq[3]);
}
// t_wp_m7 / t_wp_r7: the two poses (t, q);  t_vs_rig7: the rig's T_vs;  t_vs_cache7: the T_vs the cached T_sw
// were built with.  x = (ray, rho) in the reference sensor frame.  Outputs the three 2x6 blocks (row-major):
// dz_dx_meas, dz_dx_ref, dz_dtvs; same_pose zeroes the two pose blocks as the reference does.
BA_HD void proj_chain_two_tvs(const Cam& cam, const double* x, const double* t_wp_m7, const double* t_wp_r7,
                              const double* t_vs_rig7, const double* t_vs_cache7, bool same_pose, double* jm,
                              double* jr, double* jk) {
  const V3 ray = v3(x[0], x[1], x[2]);
  const double rho = x[3];
  const double* qp = t_wp_m7 + 3;
  const double* qr = t_wp_r7 + 3;
  const double* qv = t_vs_rig7 + 3;
  const V3 tp = v3(t_wp_m7[0], t_wp_m7[1], t_wp_m7[2]), tr = v3(t_wp_r7[0], t_wp_r7[1], t_wp_r7[2]);
  const V3 tv = v3(t_vs_rig7[0], t_vs_rig7[1], t_vs_rig7[2]), tv0 = v3(t_vs_cache7[0], t_vs_cache7[1], t_vs_cache7[2]);
  // cached transforms: T_ws = T_wp T_vs(cache) (quaternion product renormalised), T_sw its inverse
  double q_ws_m[4], q_ws_r[4], q_s[4], q_sr[4];
  quat_mul(qp, t_vs_cache7 + 3, q_ws_m); quat_normalize(q_ws_m);
  quat_mul(qr, t_vs_cache7 + 3, q_ws_r); quat_normalize(q_ws_r);
  q_s[0] = -q_ws_m[0]; q_s[1] = -q_ws_m[1]; q_s[2] = -q_ws_m[2]; q_s[3] = q_ws_m[3];
  q_sr[0] = -q_ws_r[0]; q_sr[1] = -q_ws_r[1]; q_sr[2] = -q_ws_r[2]; q_sr[3] = q_ws_r[3];
  (void)q_sr;
  const M3 R_p = quat_to_rot(qp[0], qp[1], qp[2], qp[3]), R_r = quat_to_rot(qr[0], qr[1], qr[2], qr[3]);
  const M3 R_ws_m = quat_to_rot(q_ws_m[0], q_ws_m[1], q_ws_m[2], q_ws_m[3]);
  const M3 R_ws_r = quat_to_rot(q_ws_r[0], q_ws_r[1], q_ws_r[2], q_ws_r[3]);
  const V3 t_ws_m = mul(R_p, tv0) + tp, t_ws_r = mul(R_r, tv0) + tr;
  const M3 R_s = transpose(R_ws_m);
  const V3 t_s = mul(R_s, t_ws_m) * -1.0;
  // world point (scaled by rho) and the point in the measuring sensor frame, both from the caches
  const V3 Xw = mul(R_ws_r, ray) + t_ws_r * rho;
  const V3 P = mul(R_s, Xw) + t_s * rho;
  V3 d0, d1;
  dproject(cam, P, &d0, &d1);
  // rig: T_sv = T_vs^-1
  const double q_c[4] = {-qv[0], -qv[1], -qv[2], qv[3]};
  const M3 R_v = quat_to_rot(qv[0], qv[1], qv[2], qv[3]);
  const M3 R_c = transpose(R_v);
  const double q_pw[4] = {-qp[0], -qp[1], -qp[2], qp[3]};
  const M3 R_pw = transpose(R_p);
  const double keep = same_pose ? 0.0 : 1.0;
  auto put = [&](double* j, int col, V3 v) { j[col] = -dot(d0, v); j[6 + col] = -dot(d1, v); };
  for (int k = 0; k < 3; ++k) {
    const V3 ek = v3(k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0);
    const double dk[4] = {0.5 * ek.x, 0.5 * ek.y, 0.5 * ek.z, 0.0};  // dq_exp_dw(0) e_k
    // ---- dz_dx_meas = -dpi . dt_x_dt(T_sw[cache], pw) . dt1_t2_dt2(T_sv[rig]) . dinv_exp_decoupled_dx(T_wp)
    {
      double dq_pw[4], br[4], t1[4];
      quat_mul(dk, q_pw, dq_pw);                                  // R(q_pw) E e_k
      br[0] = -dq_pw[0]; br[1] = -dq_pw[1]; br[2] = -dq_pw[2]; br[3] = -dq_pw[3];
      const V3 trk = dqx_dq_apply(q_pw, tp, dq_pw);               // translation of T_pw w.r.t. the rotation update
      quat_mul(q_c, br, t1);                                      // L(q_c) R(q_pw) (-E) e_k
      const V3 rot = mul(R_c, trk) * rho + dqx_dq_apply(q_s, Xw, t1);
      const V3 trn = mul(R_c, mul(R_pw, ek)) * -rho;
      put(jm, k, trn * keep);
      put(jm, 3 + k, rot * keep);
    }
    // ---- dz_dx_ref = -dpi . dt_x_dt(T_sw_m[cache] T_wp_r, T_vs[rig] x) . dt1_t2_dt2(T_sw_m[cache]) . dexp_decoupled_dx(T_wp_r)
    {
      double q_a[4], t1[4], t2[4];
      quat_mul(q_s, qr, q_a); quat_normalize(q_a);
      const V3 Y = mul(R_v, ray) + tv * rho;
      quat_mul(qr, dk, t1);
      quat_mul(q_s, t1, t2);
      put(jr, k, mul(R_s, ek) * (rho * keep));
      put(jr, 3 + k, dqx_dq_apply(q_a, Y, t2) * keep);
    }
    // ---- dz_dtvs = -dpi . dt_x_dt(T_sw_m T_ws_r [caches], x) . ( dt1_t2_dt2(T_sv) dt1_t2_dt2(D) dexp_decoupled_dx(T_vs)
    //                                                             + dt1_t2_dt1(T_sv, D T_vs) dinv_exp_decoupled_dx(T_vs) ),  D = T_wp_m^-1 T_wp_r
    {
      double q_T[4], q_D[4], q_2[4], a1[4], a2[4], a3[4], dq_c[4], brv[4], b1[4];
      quat_mul(q_s, q_ws_r, q_T); quat_normalize(q_T);
      quat_mul(q_pw, qr, q_D); quat_normalize(q_D);
      const M3 R_D = quat_to_rot(q_D[0], q_D[1], q_D[2], q_D[3]);
      const V3 t_D = mul(R_pw, tr - tp);
      quat_mul(q_D, qv, q_2); quat_normalize(q_2);
      const V3 t_2 = mul(R_D, tv) + t_D;
      quat_mul(qv, dk, a1); quat_mul(q_D, a1, a2); quat_mul(q_c, a2, a3);   // L(q_c) L(q_D) L(q_v) E e_k
      quat_mul(dk, q_c, dq_c);                                               // R(q_c) E e_k
      brv[0] = -dq_c[0]; brv[1] = -dq_c[1]; brv[2] = -dq_c[2]; brv[3] = -dq_c[3];
      const V3 trv = dqx_dq_apply(q_c, tv, dq_c);
      quat_mul(brv, q_2, b1);                                                // R(q_2) brv
      const double sum4[4] = {a3[0] + b1[0], a3[1] + b1[1], a3[2] + b1[2], a3[3] + b1[3]};
      const V3 rot = dqx_dq_apply(q_T, ray, sum4) + (trv + dqx_dq_apply(q_c, t_2, brv)) * rho;
      const V3 trn = (mul(R_c, mul(R_D, ek)) - mul(R_c, ek)) * rho;
      put(jk, k, trn);
      put(jk, 3 + k, rot);
    }
  }
}

}  // namespace bad
