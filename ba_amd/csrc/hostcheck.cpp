// Host build of the device math (dmath.h) so that `-m "not gpu"` tests can compare the
// kernels' closed-form Jacobians with the oracle without a GPU.  Test hook only: the
// product never calls this library.
#include <vector>
#include "dmath.h"
using namespace bad;

static Rt rt_from7(const double* p) {
  Rt t;
  t.R = quat_to_rot(p[3], p[4], p[5], p[6]);
  t.t = v3(p[0], p[1], p[2]);
  return t;
}

// camera model of the calls below: w != 0 makes cam4 the first four parameters of a FOV camera
static double g_hostcheck_fov_w = 0.0;
extern "C" void ba_hostcheck_set_fov(double w) { g_hostcheck_fov_w = w; }
static Cam cam_from4(const double* cam4) {
  Cam c = {cam4[0], cam4[1], cam4[2], cam4[3], g_hostcheck_fov_w, g_hostcheck_fov_w != 0.0 ? 1 : 0};
  return c;
}

static int g_hostcheck_variant = 1;  // 1 = proj_linearize (what k_linearize calls), 0 = proj_jacobians
extern "C" void ba_hostcheck_set_variant(int v) { g_hostcheck_variant = v; }

extern "C" void ba_hostcheck_proj_jacobians(int lm_dim, const double* cam4, const double* z,
                                            const double* x, const double* t_wp_m7,
                                            const double* t_vs_m7, const double* t_wp_r7,
                                            const double* t_vs_r7, int same_pose, double* r2,
                                            double* jm12, double* jr12, double* jl) {
  Cam cam = cam_from4(cam4);
  const Rt t_wp_m = rt_from7(t_wp_m7), t_vs_m = rt_from7(t_vs_m7);
  const Rt t_wp_r = rt_from7(t_wp_r7), t_vs_r = rt_from7(t_vs_r7);
  const Rt t_sw_m = inverse(compose(t_wp_m, t_vs_m));
  const Rt t_ws_r = compose(t_wp_r, t_vs_r);
  const Rt t_sv_m = inverse(t_vs_m);
  // the kernels' form (proj_linearize: fewer transforms) must agree with the literal closed form
  // (proj_jacobians) — both are compared with the oracle by tests/test_hostcheck.py (variant flag)
  if (lm_dim == 1) {
    ProjJac<1> o;
    if (g_hostcheck_variant == 0) proj_jacobians<1>(cam, z, x, t_sw_m, t_ws_r, t_wp_m, t_sv_m, t_wp_r, t_vs_r, same_pose != 0, &o);
    else proj_linearize<1>(cam, z, x, t_sw_m, t_vs_m.R, t_sv_m.t, t_ws_r, t_wp_r, same_pose != 0, &o);
    for (int i = 0; i < 2; ++i) r2[i] = o.r[i];
    for (int i = 0; i < 12; ++i) { jm12[i] = o.jm[i]; jr12[i] = o.jr[i]; }
    for (int i = 0; i < 2; ++i) jl[i] = o.jl[i];
  } else {
    ProjJac<3> o;
    if (g_hostcheck_variant == 0) proj_jacobians<3>(cam, z, x, t_sw_m, t_ws_r, t_wp_m, t_sv_m, t_wp_r, t_vs_r, same_pose != 0, &o);
    else proj_linearize<3>(cam, z, x, t_sw_m, t_vs_m.R, t_sv_m.t, t_ws_r, t_wp_r, same_pose != 0, &o);
    for (int i = 0; i < 2; ++i) r2[i] = o.r[i];
    for (int i = 0; i < 12; ++i) { jm12[i] = o.jm[i]; jr12[i] = o.jr[i]; }
    for (int i = 0; i < 6; ++i) jl[i] = o.jl[i];
  }
}

// dz_dtvs rows of the DoTvs instantiations as k_linearize<.., CAL> evaluates them (LmSize 1)
extern "C" void ba_hostcheck_proj_tvs_jacobian(const double* cam4, const double* z, const double* x,
                                               const double* t_wp_m7, const double* t_vs_m7, const double* t_wp_r7,
                                               const double* t_vs_r7, int same_pose, double* jk12) {
  Cam cam = cam_from4(cam4);
  const Rt t_wp_m = rt_from7(t_wp_m7), t_vs_m = rt_from7(t_vs_m7);
  const Rt t_wp_r = rt_from7(t_wp_r7), t_vs_r = rt_from7(t_vs_r7);
  const Rt t_sw_m = inverse(compose(t_wp_m, t_vs_m));
  const Rt t_ws_r = compose(t_wp_r, t_vs_r);
  const Rt t_sv_m = inverse(t_vs_m);
  ProjJac<1> o;
  proj_linearize<1, true>(cam, z, x, t_sw_m, t_vs_m.R, t_sv_m.t, t_ws_r, t_wp_r, same_pose != 0, &o, jk12);
}

// dz_dcam_params rows of the CalibSize instantiations (proj_intrinsics_rows, what k_linearize<.., 2> calls)
extern "C" void ba_hostcheck_proj_intrinsics_jacobian(const double* cam4, const double* z_ref, double rho,
                                                      const double* t_wp_m7, const double* t_vs_m7,
                                                      const double* t_wp_r7, const double* t_vs_r7, double* jk12) {
  Cam cam = cam_from4(cam4);
  const Rt t_sw_m = inverse(compose(rt_from7(t_wp_m7), rt_from7(t_vs_m7)));
  const Rt t_ws_r = compose(rt_from7(t_wp_r7), rt_from7(t_vs_r7));
  proj_intrinsics_rows(cam, z_ref, rho, t_sw_m, t_ws_r, 1.0, jk12);
}

// the reference's chains with cached and rig extrinsics apart (proj_chain_two_tvs)
extern "C" void ba_hostcheck_proj_chain_two_tvs(const double* cam4, const double* x, const double* t_wp_m7,
                                                const double* t_wp_r7, const double* t_vs_rig7,
                                                const double* t_vs_cache7, int same_pose, double* jm12,
                                                double* jr12, double* jk12) {
  Cam cam = cam_from4(cam4);
  proj_chain_two_tvs(cam, x, t_wp_m7, t_wp_r7, t_vs_rig7, t_vs_cache7, same_pose != 0, jm12, jr12, jk12);
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

// the two-launch form of the device (k_imu_steps + k_imu): step Jacobians per sample first, then the
// sequential part — must reproduce the fused form bit for bit
extern "C" void ba_hostcheck_imu_split(const double* p1_16, const double* p2_16, const double* meas,
                                       int nmeas, const double* g3, const double* r6, const double* rb6,
                                       int RS, double* r15, double* dz1, double* dz2, double* cov_inv) {
  std::vector<double> steps((size_t)160 * (nmeas > 0 ? nmeas : 1), 0.0);
  for (int k = 1; k < nmeas; ++k) imu_step_jacobians(p1_16, meas, k, g3, &steps[(size_t)160 * k]);
  ImuOut o;
  imu_residual(p1_16, p2_16, meas, nmeas, g3, r6, rb6, RS, true, &o, nullptr, nullptr, steps.data());
  for (int i = 0; i < 15; ++i) r15[i] = o.r[i];
  for (int i = 0; i < 225; ++i) { dz1[i] = o.dz1.m[i]; dz2[i] = o.dz2.m[i]; cov_inv[i] = o.cov_inv.m[i]; }
}

// ---- static structure (structure.h) --------------------------------------------------------------
// Builds the lists of ba_hip_finalize for a graph and evaluates them on the CPU exactly as the
// device kernels do — k_linearize's row layout and landmark blocks, k_assemble_tiles' tile
// references, k_pose_blocks' per-pose terms — from per-residual Jacobians supplied by the caller.
// tests/test_structure_lists.py compares the result with a dense brute-force Schur complement, so
// the index logic is validated without a GPU.
#include "structure.h"
#include <cstring>

extern "C" int ba_hostcheck_schur_lists(
    int LM, int D, uint32_t P, const uint8_t* pose_active, uint32_t L, const uint8_t* lm_active,
    const uint32_t* lm_ref_pose, uint32_t O, const uint32_t* proj_pose, const uint32_t* proj_lm,
    const double* jm12, const double* jr12, const double* jl, const double* r2, const double* w,
    double* S_lower /* ld x ld, lower storage as on the device */, double* rhs_p, double* rhs_sc,
    double* vinv /* [L][LM*LM] */, double* bl /* [L][LM] */, uint32_t* out_ld, uint32_t* out_counts /* [8] */) {
  using namespace bae;
  Problem pb;
  pb.num_cams = 1; pb.num_poses = P; pb.num_lms = L; pb.num_proj = O;
  pb.pose_active.assign(pose_active, pose_active + P);
  pb.lm_active.assign(lm_active, lm_active + L);
  pb.lm_ref_pose.assign(lm_ref_pose, lm_ref_pose + L);
  pb.lm_ref_cam.assign(L, 0);
  pb.proj_pose.assign(proj_pose, proj_pose + O);
  pb.proj_lm.assign(proj_lm, proj_lm + O);
  pb.proj_cam.assign(O, 0);
  pb.proj_z.assign(2 * (size_t)O, 0.0);
  pb.proj_w.assign(O, 1.0);
  Lists st;
  std::string err;
  if (!build_lists(pb, LM, D, st, err)) return -1;
  const uint32_t R = st.R, WO = (uint32_t)w_row_offset(LM), ld = st.ld;
  *out_ld = ld;
  out_counts[0] = st.n_chunks; out_counts[1] = st.n_pairs; out_counts[2] = (uint32_t)st.n_pair_entries;
  out_counts[3] = (uint32_t)st.n_tile_refs; out_counts[4] = (uint32_t)st.n_pose_entries; out_counts[5] = st.n_inc;
  out_counts[6] = st.n_rows; out_counts[7] = st.Pact;
  // linearisation waves: whole landmarks, every observation covered exactly once, big ones last
  {
    std::vector<uint8_t> seen(st.O, 0);
    for (uint32_t c = 0; c < st.n_chunks; ++c) {
      const uint32_t a0 = st.wave_rng[c].x, a1 = st.wave_rng[c].y;
      if (a1 <= a0 || a1 > st.O) return -2;
      const uint32_t l0 = st.obs_lm[a0], l1 = st.obs_lm[a1 - 1];
      if (st.lm_ptr[l0] != a0 || st.lm_ptr[l1 + 1] != a1) return -3;
      const bool big = c >= st.n_chunks - st.n_big_chunks;
      if (big != (a1 - a0 > 64) || (big && l0 != l1)) return -4;
      for (uint32_t a = a0; a < a1; ++a) { if (seen[a]) return -5; seen[a] = 1; }
    }
    for (uint32_t a = 0; a < st.O; ++a) if (!seen[a]) return -5;
  }
  // ---- k_linearize, emulated per landmark ------------------------------------------------------
  std::vector<double> frow((size_t)st.n_rows * 6, 0.0), scal(st.n_scalars, 0.0);
  const int LL = LM > 0 ? LM : 1;
  for (uint32_t l = 0; l < L; ++l) {
    double V[9] = {0}, b[3] = {0}, Wr[6] = {0};
    const bool act = st.lm_opt[l] >= 0;
    for (uint32_t s = st.lm_ptr[l]; s < st.lm_ptr[l + 1]; ++s) {
      const uint32_t a = st.obs_rid[s];
      const bool same = LM == 1 && st.obs_pose[s] == lm_ref_pose[l];  // parallel_algos.h:97-99,111-113
      double Jm[12], Jr[12];
      for (int i = 0; i < 12; ++i) { Jm[i] = same ? 0.0 : jm12[12 * (size_t)a + i]; Jr[i] = (same || LM != 1) ? 0.0 : jr12[12 * (size_t)a + i]; }
      const double* Jl = jl + 2 * LL * (size_t)a;
      const double ww = w[a], sw = std::sqrt(ww);
      scal[2 * (size_t)s] = r2[2 * (size_t)a] * sw;
      scal[2 * (size_t)s + 1] = r2[2 * (size_t)a + 1] * sw;
      double* rows = &frow[(size_t)s * R * 6];
      for (int i = 0; i < 12; ++i) rows[i] = Jm[i] * sw;
      if (LM == 1) for (int i = 0; i < 12; ++i) rows[12 + i] = Jr[i] * sw;
      if (!act) continue;
      for (int p = 0; p < LM; ++p) {
        for (int q = 0; q < LM; ++q) V[p * LM + q] += (Jl[p] * Jl[q] + Jl[LM + p] * Jl[LM + q]) * ww;
        b[p] += (Jl[p] * r2[2 * (size_t)a] + Jl[LM + p] * r2[2 * (size_t)a + 1]) * ww;
      }
      for (int k = 0; k < LM; ++k)
        for (int x = 0; x < 6; ++x) rows[(WO + k) * 6 + x] = (Jm[x] * Jl[k] + Jm[6 + x] * Jl[LM + k]) * ww;
      if (LM == 1) for (int x = 0; x < 6; ++x) Wr[x] += (Jr[x] * Jl[0] + Jr[6 + x] * Jl[1]) * ww;
    }
    if (!act) continue;
    double Vi[9];
    if (LM == 1) {
      if (std::fabs(V[0]) < 1e-6) V[0] += 1e-6;
      Vi[0] = 1.0 / V[0];
    } else {
      double nrm = 0;
      for (int i = 0; i < 9; ++i) nrm += V[i] * V[i];
      if (std::sqrt(nrm) < 1e-6) { V[0] += 1e-6; V[4] += 1e-6; V[8] += 1e-6; }
      const double a = V[0], bb = V[1], c = V[2], d = V[3], e = V[4], f = V[5], g = V[6], h = V[7], i = V[8];
      const double A00 = e * i - f * h, A01 = c * h - bb * i, A02 = bb * f - c * e;
      const double A10 = f * g - d * i, A11 = a * i - c * g, A12 = c * d - a * f;
      const double A20 = d * h - e * g, A21 = bb * g - a * h, A22 = a * e - bb * d;
      const double id = 1.0 / (a * A00 + bb * A10 + c * A20);
      const double t[9] = {A00 * id, A01 * id, A02 * id, A10 * id, A11 * id, A12 * id, A20 * id, A21 * id, A22 * id};
      std::memcpy(Vi, t, sizeof(t));
    }
    for (int i = 0; i < LM * LM; ++i) vinv[(size_t)l * LM * LM + i] = Vi[i];
    for (int i = 0; i < LM; ++i) { bl[(size_t)l * LM + i] = b[i]; scal[2 * (size_t)st.O + (size_t)l * LM + i] = b[i]; }
    auto nwv = [&](const double* Wrows, double* out) {  // out[c] = -(W Vi)[:, c]
      for (int c = 0; c < LM; ++c)
        for (int x = 0; x < 6; ++x) {
          double sacc = 0;
          for (int k = 0; k < LM; ++k) sacc += Wrows[k * 6 + x] * Vi[k * LM + c];
          out[c * 6 + x] = -sacc;
        }
    };
    for (uint32_t s = st.lm_ptr[l]; s < st.lm_ptr[l + 1]; ++s) {
      double* rows = &frow[(size_t)s * R * 6];
      nwv(rows + WO * 6, rows + (WO + LM) * 6);
    }
    if (LM == 1) {
      double* lr = &frow[((size_t)st.lrow_base + 2 * l) * 6];
      for (int x = 0; x < 6; ++x) lr[x] = Wr[x];
      nwv(lr, lr + 6);
    }
  }
  // ---- k_assemble_tiles ------------------------------------------------------------------------------
  std::memset(S_lower, 0, sizeof(double) * (size_t)ld * ld);
  const uint32_t nt = ld / 64;
  uint64_t t = 0;
  for (uint32_t tr = 0; tr < nt; ++tr)
    for (uint32_t tc = 0; tc <= tr; ++tc, ++t)
      for (uint32_t q = st.tile_ptr[t]; q < st.tile_ptr[t + 1]; ++q) {
        const U2 ref = st.tile_ref[q];
        const uint32_t cnt = ref.y >> 14;
        const int ro = (int)((ref.y >> 7) & 127) - kRefBias, co = (int)(ref.y & 127) - kRefBias;
        double acc[36] = {0};
        for (uint32_t e = ref.x; e < ref.x + cnt; ++e) {
          const double* a = &frow[(size_t)st.pair_ent[e].x * 6];
          const double* bb = &frow[(size_t)st.pair_ent[e].y * 6];
          for (int x = 0; x < 6; ++x)
            for (int y = 0; y < 6; ++y) acc[x * 6 + y] += a[x] * bb[y];
        }
        for (int x = 0; x < 6; ++x)
          for (int y = 0; y < 6; ++y) {
            const int rr = ro + y, cc = co + x;  // block (i,j), i < j, is stored transposed
            if (rr < 0 || rr >= 64 || cc < 0 || cc >= 64) continue;
            double& dst = S_lower[((size_t)tr * 64 + rr) * ld + (size_t)tc * 64 + cc];
            if (dst != 0.0) return -6;  // two blocks must never overlap
            dst = acc[x * 6 + y];
          }
      }
  // ---- k_pose_blocks -----------------------------------------------------------------------------------
  for (uint32_t p = 0; p < st.Pact; ++p) {
    double blk[36] = {0}, ga[6] = {0}, gb[6] = {0};
    for (uint32_t e = st.pose_ptr[p]; e < st.pose_ptr[p + 1]; ++e) {
      const U3 en = st.pose_ent[e];
      const double* a = &frow[(size_t)en.a * 6];
      const double* bb = &frow[(size_t)en.b * 6];
      const double sc = scal[en.s];
      for (int x = 0; x < 6; ++x) {
        for (int y = 0; y < 6; ++y) blk[x * 6 + y] += a[x] * bb[y];
        (e < st.pose_mid[p] ? ga : gb)[x] += a[x] * sc;
      }
    }
    for (int x = 0; x < 6; ++x) {
      for (int y = 0; y < 6; ++y) S_lower[((size_t)p * D + x) * ld + (size_t)p * D + y] = blk[x * 6 + y];
      rhs_p[(size_t)p * D + x] = ga[x];
      rhs_sc[(size_t)p * D + x] = ga[x] + gb[x];
    }
  }
  return 0;
}
