// Host build of the device math (dmath.h) so that `-m "not gpu"` tests can compare the
// kernels' closed-form Jacobians with the oracle without a GPU.  Test hook only: the
// product never calls this library.
#include "dmath.h"
using namespace bad;

static Rt rt_from7(const double* p) {
  Rt t;
  t.R = quat_to_rot(p[3], p[4], p[5], p[6]);
  t.t = v3(p[0], p[1], p[2]);
  return t;
}

extern "C" void ba_hostcheck_proj_jacobians(int lm_dim, const double* cam4, const double* z,
                                            const double* x, const double* t_wp_m7,
                                            const double* t_vs_m7, const double* t_wp_r7,
                                            const double* t_vs_r7, int same_pose, double* r2,
                                            double* jm12, double* jr12, double* jl) {
  Cam cam = {cam4[0], cam4[1], cam4[2], cam4[3]};
  const Rt t_wp_m = rt_from7(t_wp_m7), t_vs_m = rt_from7(t_vs_m7);
  const Rt t_wp_r = rt_from7(t_wp_r7), t_vs_r = rt_from7(t_vs_r7);
  const Rt t_sw_m = inverse(compose(t_wp_m, t_vs_m));
  const Rt t_ws_r = compose(t_wp_r, t_vs_r);
  const Rt t_sv_m = inverse(t_vs_m);
  if (lm_dim == 1) {
    ProjJac<1> o;
    proj_jacobians<1>(cam, z, x, t_sw_m, t_ws_r, t_wp_m, t_sv_m, t_wp_r, t_vs_r, same_pose != 0, &o);
    for (int i = 0; i < 2; ++i) r2[i] = o.r[i];
    for (int i = 0; i < 12; ++i) { jm12[i] = o.jm[i]; jr12[i] = o.jr[i]; }
    for (int i = 0; i < 2; ++i) jl[i] = o.jl[i];
  } else {
    ProjJac<3> o;
    proj_jacobians<3>(cam, z, x, t_sw_m, t_ws_r, t_wp_m, t_sv_m, t_wp_r, t_vs_r, same_pose != 0, &o);
    for (int i = 0; i < 2; ++i) r2[i] = o.r[i];
    for (int i = 0; i < 12; ++i) { jm12[i] = o.jm[i]; jr12[i] = o.jr[i]; }
    for (int i = 0; i < 6; ++i) jl[i] = o.jl[i];
  }
}

// ---- pose-pose residuals (dpose.h) ---------------------------------------------------
#include "dpose.h"

extern "C" void ba_hostcheck_unary(const double* t_wp7, const double* t_prior7, int use_rotation,
                                   double* r6, double* J36) {
  DM<6, 6> J;
  unary_residual(tq_from7(t_wp7), tq_from7(t_prior7), use_rotation, r6, &J);
  for (int i = 0; i < 36; ++i) J36[i] = J.m[i];
}

extern "C" void ba_hostcheck_binary(const double* t_w1, const double* t_w2, const double* t_12,
                                    const double* cov_inv36, const double* cov_inv_sqrt36,
                                    double weight, int use_rotation, double* h11, double* h12,
                                    double* h22, double* g1, double* g2, double* err_build,
                                    double* err_eval) {
  PPBlocks o;
  binary_blocks(tq_from7(t_w1), tq_from7(t_w2), tq_from7(t_12), cov_inv36, cov_inv_sqrt36, weight,
                use_rotation, &o, err_eval);
  for (int i = 0; i < 225; ++i) { h11[i] = o.h11.m[i]; h12[i] = o.h12.m[i]; h22[i] = o.h22.m[i]; }
  for (int i = 0; i < 15; ++i) { g1[i] = o.g1[i]; g2[i] = o.g2[i]; }
  *err_build = o.err_build;
}

extern "C" void ba_hostcheck_imu(const double* p1_16, const double* p2_16, const double* meas,
                                 int nmeas, const double* g3, const double* r6, const double* rb6,
                                 int RS, double* r15, double* dz1, double* dz2, double* cov_inv) {
  ImuOut o;
  imu_residual(p1_16, p2_16, meas, nmeas, g3, r6, rb6, RS, true, &o);
  for (int i = 0; i < 15; ++i) r15[i] = o.r[i];
  for (int i = 0; i < 225; ++i) { dz1[i] = o.dz1.m[i]; dz2[i] = o.dz2.m[i]; cov_inv[i] = o.cov_inv.m[i]; }
}
