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
