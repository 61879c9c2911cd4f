/*
 * ba_hip.h — C-ABI of the MI355X (gfx950) engine behind ba::BundleAdjuster<>::Solve().
 *
 * The reference (arpg/ba) has no FFI layer: its boundary is the C++ class template
 * ba::BundleAdjuster (/root/reference/include/ba/BundleAdjuster.h:111-753) whose only
 * non-inline member is Solve() (/root/reference/src/BundleAdjuster.cpp:278-705).  This
 * header is the boundary inserted *inside* Solve(): the host class (include/ba/
 * BundleAdjuster.h in this repo) marshals its problem graph into flat SoA arrays once
 * per Solve() and then drives one Gauss-Newton / dogleg iteration as a short sequence of
 * the calls below; only scalars come back per iteration.  Each entry point names the
 * reference code it replaces.
 *
 * Conventions
 *   - plain C, no C++ or torch types; every pointer argument is caller-owned host memory
 *     that is copied during the call unless stated otherwise;
 *   - every function returns int: 0 = ok, <0 = HIP/runtime failure (ba_hip_last_error()
 *     gives text), >0 = numeric status mirroring ba::OptimizationResult
 *     (BundleAdjuster.h:38-46): BA_HIP_FACTORIZATION_ERROR;
 *   - never throws; no global state; one host thread per engine; calls are synchronous
 *     on return (the engine's stream is drained) unless suffixed _async;
 *   - rigid transforms are 7 doubles [tx,ty,tz,qx,qy,qz,qw]; ids are dense uint32 in
 *     insertion order exactly as the reference's Add* calls return them;
 *   - multi-GPU: one engine per device, each holding ALL poses and its shard of
 *     landmarks/projection residuals; the per-iteration sums that cross shards go
 *     through the caller-supplied all-reduce hook (ba_hip_set_allreduce), e.g. RCCL; with
 *     the collectives hook (ba_hip_set_collectives) the reduced solve itself is distributed.
 */
#ifndef BA_HIP_H
#define BA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BA_HIP_OK 0
#define BA_HIP_FACTORIZATION_ERROR 4 /* ba::FactorizationError, BundleAdjuster.h:44 */
#define BA_HIP_SOLVER_ERROR 5        /* ba::SolverError,        BundleAdjuster.h:45 */

typedef struct ba_hip_engine ba_hip_engine;

/* Options the device phases need (subset of ba::Options, BundleAdjuster.h:72-107). */
typedef struct {
  double projection_outlier_threshold;            /* BundleAdjuster.cpp:184 */
  int32_t use_robust_norm_for_proj_residuals;     /* BundleAdjuster.cpp:1377 */
  int32_t use_robust_norm_for_inertial_residuals; /* BundleAdjuster.cpp:1513 */
  int32_t use_triangular_matrices;                /* only affects debug downloads of S */
  int32_t keep_reduced_system;                    /* keep a copy of S for ba_hip_get_S after the
                                                     in-place factorisation (the reference's
                                                     write_reduced_camera_matrix, BundleAdjuster.cpp:600-627) */
  double gyro_sigma, accel_sigma, gyro_bias_sigma, accel_bias_sigma; /* BundleAdjuster.h:204-218 */
  /* Rank-deficiency guard of the reduced solve (extension; 0 = off = the reference's behaviour).
   * The reference reports FactorizationError only for a pivot that is EXACTLY zero (Eigen's LDLT /
   * SimplicialLDLT info(), BundleAdjuster.cpp:756-759) — on a numerically singular S whether an
   * elimination cancels to exactly zero is rounding-order luck.  With tol > 0 a pivot d_j with
   * |d_j| < tol * |S_jj| (S_jj = the diagonal entry before elimination) raises
   * BA_HIP_FACTORIZATION_ERROR instead of letting an arbitrary step through. */
  double pivot_rel_tolerance;
} ba_hip_options;

/* The four error sums of EvaluateResiduals / BuildProblem (BundleAdjuster.cpp:144-274,
 * 1340-1537; BundleAdjuster.h:593-602). */
typedef struct {
  double proj_error, binary_error, unary_error, inertial_error;
} ba_hip_errors;

/* Scalars of the dogleg step (BundleAdjuster.cpp:858-1017): squared norms and dot
 * products over the pose (p) and landmark (l) parts of rhs (= J^T r, unreduced) and of
 * the Gauss-Newton step, and the steepest-descent denominator ||J rhs||^2. */
typedef struct {
  double rhs_p_sq, rhs_l_sq;   /* :858 numerator */
  double j_rhs_sq;             /* :906-910 denominator */
  double gn_p_sq, gn_l_sq;     /* :971-973 */
  double rhs_gn_p, rhs_gn_l;   /* dot products for a, b of :994-998 */
  /* calibration part (ba_hip_set_calibration; zeros otherwise): |rhs_k|^2 (:859), |delta_k|^2 of
   * the Gauss-Newton step (:972), their dot product (:997).  j_rhs_sq already includes
   * |J_k rhs_k|^2 (:883-886, 910). */
  double rhs_k_sq, gn_k_sq, rhs_gn_k;
} ba_hip_dogleg_scalars;

/* Norms of the step formed by ba_hip_compose_step (summary_.delta_norm is their sum,
 * BundleAdjuster.cpp:26). */
typedef struct {
  double step_p_norm, step_l_norm;
} ba_hip_step_norms;

/* Per-phase device time of the last iteration, milliseconds (HIP events on the engine's
 * stream); names follow the reference's PrintTimer sites (Utils.h:51-62). */
typedef struct {
  double j_evaluation, robust_weights, jtj_schur, solve, back_substitution,
         evaluate_residuals, apply_update;
} ba_hip_timers;

/* ---- lifetime ------------------------------------------------------------------ */
/* lm_dim in {0,1,3}, pose_dim in {6,9,15} (the reference's LmSize / PoseSize template
 * parameters, BundleAdjuster.h:111-134).  stream: a hipStream_t to run on, or NULL for
 * an engine-owned stream.  Fails (<0) when no HIP device is usable: there is no CPU
 * fallback. */
int ba_hip_create(int lm_dim, int pose_dim, int device, void* stream, ba_hip_engine** out);
void ba_hip_destroy(ba_hip_engine* e);
const char* ba_hip_last_error(const ba_hip_engine* e);
int ba_hip_set_options(ba_hip_engine* e, const ba_hip_options* o);
/* The reference's CalibSize / DoTvs template parameters (BundleAdjuster.h:110-134).  do_tvs != 0:
 * the extrinsics T_vs of camera 0 become six more unknowns BEHIND the pose unknowns of the reduced
 * system (kCalibDim = 6, kTvsOffset = 0; BundleAdjuster.cpp:316-322, 493-583): every vector the
 * calls below size with ba_hip_num_pose_params() grows by ba_hip_num_calib_params() trailing
 * entries (rhs, Gauss-Newton delta, step), ba_hip_get_S returns the bordered (n + 6)^2 matrix, and
 * ba_hip_apply_step moves T_vs by exp_decoupled(T_vs, -delta_k) (:72-83) — read it back with
 * ba_hip_get_cameras.  As in the reference a rolled-back step does NOT restore T_vs (:1060-1068).
 * calib_size = 4 (with do_tvs = 0): the pinhole parameters (fx, fy, u0, v0) of camera 0 become four
 * unknowns the same way (kCamParamsInCalib; BundleAdjuster.cpp:46-69, parallel_algos.h:114-118:
 * dz_dcam_params = -dTransfer_dparams(T_sw_m T_ws_r, z_ref, rho)); needs the reference pixel of every
 * landmark (ba_hip_set_landmark_ref_pixels).  ba_hip_apply_step moves the parameters by -delta_k and
 * re-derives every x_s ray from its reference pixel (:57-68); a rollback restores them (:1066, :1147);
 * read them back with ba_hip_get_camera_params.  calib_size = 5: the same for a FovCamera 0
 * (fx, fy, u0, v0, w; ba_hip_set_camera_models) — the reference's SelfCalBundleAdjuster
 * (BundleAdjuster.h:758-759).  calib_size must equal the parameter count of camera 0 (checked by
 * ba_hip_finalize; the reference's fixed-size assignment at parallel_algos.h:115-118).
 * Both at once is refused (the reference's T_vs block wipes the intrinsics columns it shares a
 * j_kpr_ entry with, :1775-1783), as is any other size.  LmSize 1 only (parallel_algos.h:102-131).
 * Structural: call before ba_hip_finalize. */
int ba_hip_set_calibration(ba_hip_engine* e, int calib_size, int do_tvs);
/* LandmarkT::z_ref (BundleAdjuster.h:476-483): the pixel of every landmark in its reference
 * camera, two doubles per landmark id.  Only the intrinsics calibration reads it. */
int ba_hip_set_landmark_ref_pixels(ba_hip_engine* e, uint32_t n, const double* z_ref2);
/* [fx, fy, u0, v0] of every camera as the engine currently holds them; n = the camera count of
 * ba_hip_set_cameras (a mismatch is an error, nothing is written) */
int ba_hip_get_camera_params(ba_hip_engine* e, uint32_t n, double* params4);

/* ---- problem upload (replaces the AoS graph of Types.h:41-321) -------------------- */
/* calibu::Rig cameras: pinhole params [fx,fy,u0,v0] and T_vs (BundleAdjuster.h:259-263) */
int ba_hip_set_cameras(ba_hip_engine* e, uint32_t n, const double* params4, const double* t_vs7);
/* Camera model of every camera of ba_hip_set_cameras (call after it; it resets them to 0):
 * model 0 = calibu::LinearCamera (pinhole), 1 = calibu::FovCamera with distortion parameter w[c]
 * (pix = K (f(r) p), p = P.xy / P.z, f(r) = atan(2 r tan(w/2)) / (r w) — Devernay & Faugeras 2001; Calibu is
 * not in the reference tree, see oracle/outils.h).  w is read for model 1 only. */
int ba_hip_set_camera_models(ba_hip_engine* e, uint32_t n, const int32_t* model, const double* w);
/* w of every camera (0 for a LinearCamera) as the engine currently holds it; n as above */
int ba_hip_get_camera_fov(ba_hip_engine* e, uint32_t n, double* w);
/* Options::use_per_pose_cam_params (BundleAdjuster.h:96; parallel_algos.h:54-57,
 * BundleAdjuster.cpp:162-176): every projection residual is evaluated with the pinhole
 * intrinsics [fx,fy,u0,v0] of its MEASUREMENT pose (PoseT::cam_params, Types.h:46) instead of the
 * rig camera's.  n = number of poses (checked at ba_hip_begin_solve), n = 0 switches back to the
 * rig intrinsics.  May be called at any time before ba_hip_linearize; not part of the structure. */
int ba_hip_set_pose_cam_params(ba_hip_engine* e, uint32_t n, const double* params4);
/* poses_ (BundleAdjuster.h:292-323); v_w/b may be NULL (zeros) */
int ba_hip_set_poses(ba_hip_engine* e, uint32_t n, const double* t_wp7, const double* v_w3,
                     const double* b6, const uint8_t* is_active);
/* landmarks_ (BundleAdjuster.h:326-367): homogeneous world point, reference pose/camera */
int ba_hip_set_landmarks(ba_hip_engine* e, uint32_t n, const double* x_w4,
                         const uint32_t* ref_pose_id, const uint32_t* ref_cam_id,
                         const uint8_t* is_active);
/* ACCEPTED projection residuals in residual-id order (BundleAdjuster.h:459-513) */
int ba_hip_set_projection_residuals(ba_hip_engine* e, uint32_t n, const double* z2,
                                    const uint32_t* meas_pose_id, const uint32_t* landmark_id,
                                    const uint32_t* cam_id, const double* weight);
/* Projection residuals that are "conditioning" (reference pose inactive, measuring pose active,
 * BundleAdjuster.h:503-510): ids into the list above.  Only SolutionSummary::cond_proj_error uses
 * them; ba_hip_get_conditioning_error returns the sum of |residual|^2 over them at the current state
 * (BundleAdjuster.cpp:692-703), formed on the device. */
int ba_hip_set_conditioning_residuals(ba_hip_engine* e, uint32_t n, const uint32_t* residual_id);
int ba_hip_get_conditioning_error(ba_hip_engine* e, double* proj_sq_sum);
/* unary_residuals_ (BundleAdjuster.h:377-407): prior pose and cov^-1 (6x6 row-major) */
int ba_hip_set_unary_residuals(ba_hip_engine* e, uint32_t n, const uint32_t* pose_id,
                               const double* t_wp7, const double* cov_inv36,
                               const uint8_t* use_rotation);
/* binary_residuals_ (BundleAdjuster.h:425-456): cov^-1 and its square root as computed
 * at Add time, weight */
int ba_hip_set_binary_residuals(ba_hip_engine* e, uint32_t n, const uint32_t* pose1_id,
                                const uint32_t* pose2_id, const double* t_12_7,
                                const double* cov_inv36, const double* cov_inv_sqrt36,
                                const double* weight, const uint8_t* use_rotation);
/* inertial_residuals_ (BundleAdjuster.h:516-546): CSR over the sample table,
 * samples are rows [wx,wy,wz,ax,ay,az,time] (Types.h:222-244) */
int ba_hip_set_imu_residuals(ba_hip_engine* e, uint32_t n, const uint32_t* pose1_id,
                             const uint32_t* pose2_id, const uint32_t* meas_ptr /* n+1 */,
                             const double* meas7, const double* weight);
/* ImuResidualT::IntegrateResidual without Jacobians (Types.h:662-738): RK4 integration of `nmeas`
 * IMU samples [wx,wy,wz,ax,ay,az,time] from the state (t_wp7, v_w3) with biases bg3 / ba3 and
 * gravity g3 — host code, the same source as the device kernels (no engine, no GPU needed).
 * states10 receives max(nmeas, 1) rows [t(3) q(4) v(3)]: the start state, then the state at every
 * later sample (the reference's `poses` vector).  Returns 0, or -1 on a NULL argument. */
int ba_hip_integrate_imu(const double t_wp7[7], const double v_w3[3], const double bg3[3], const double ba3[3],
                         const double g3[3], const double* meas7, uint32_t nmeas, double* states10);
/* The same with the Jacobian outputs of the reference's signature (Types.h:662-738), all optional:
 * dpose_db60 (10 x 6, row-major: state [t q v] over the gyro / accelerometer biases), dpose_dpose100
 * (10 x 10, over the start state) and the covariance c_res100 (10 x 10, in/out: C <- F C F^T + G R G^T
 * per step, r6 = diagonal of R).  As in the reference they are formed only when one of the two
 * Jacobians is asked for and r6 is not NULL. */
int ba_hip_integrate_imu_jacobians(const double t_wp7[7], const double v_w3[3], const double bg3[3],
                                   const double ba3[3], const double g3[3], const double* meas7, uint32_t nmeas,
                                   const double r6[6], double* states10, double* dpose_db60,
                                   double* dpose_dpose100, double* c_res100);
/* ImuResidualT::GetPoseDerivative (Types.h:376-416): k9 = [v; R (w + b_g); R (a + b_a) - g] of the
 * state10 = [t(3) q(4) v(3)] with the two samples interpolated at z_start.time + dt; dk_db54 (9 x 6)
 * and dk_dx90 (9 x 10) may be NULL.  ImuResidualT::IntegratePose (Types.h:324-373): out10 = the state
 * advanced by k9 * dt (q <- exp(k_w dt) q, not renormalised); dy_dk90 (10 x 9) and the quaternion
 * block dy_dy16 (4 x 4) may be NULL.  Host code, no GPU. */
int ba_hip_imu_pose_derivative(const double state10[10], const double g3[3], const double z_start7[7],
                               const double z_end7[7], const double bg3[3], const double ba3[3], double dt,
                               double k9[9], double* dk_db54, double* dk_dx90);
int ba_hip_imu_integrate_pose(const double state10[10], const double k9[9], double dt, double out10[10],
                              double* dy_dk90, double* dy_dy16);
/* The Lie-group / quaternion helpers of the reference's include/ba/Utils.h, host code shared with the kernels
 * (include/ba/Utils.h wraps this entry with the reference's names).  Transforms travel as [t(3) q(4)],
 * quaternions as x,y,z,w; results are row-major.  op: 1 dlog_dq(q) 3x4 | 2 dq_exp_dw(w) 4x3 | 3 dq1q2_dq1(q2) 4x4 |
 * 4 dq1q2_dq2(q1) 4x4 | 5 dqx_dq(q, x3) 3x4 | 6 dqx_dx(q) 3x3 | 7 log_decoupled(a, b) 6 | 8 exp_decoupled(a, x6) 7 |
 * 9 dlog_decoupled_dx(a, b) 6x6 | 10 dLog_decoupled_dt1(t1, t2) 6x7 | 11 dlog_decoupled_dt2(t1, t2) 6x7 |
 * 12 dexp_decoupled_dx(t) 7x6 | 13 dinv_exp_decoupled_dx(t) 7x6 | 14 dt_x_dt(t, x4) 4x7 | 15 dt1_t2_dt1(t1, t2) 7x7 |
 * 16 dt1_t2_dt2(t1) 7x7 | 17 MultHomogeneous(t, x4) 4.  Returns the number of doubles written, -1 on a bad call. */
int ba_hip_lie(int op, const double* a, const double* b, double* out);
/* ImuCalibrationT::r and r_b (Types.h:112-159): diagonal of the IMU measurement noise (gyro x3,
 * accelerometer x3) and of the bias random walk, as parallel_algos.h:204,288 read them from imu_.
 * NULL pointers: derive both from the sigmas of ba_hip_options (what Init() does,
 * BundleAdjuster.h:204-218).  Like the gravity vector it is not part of the structure: it takes
 * effect at the next ba_hip_begin_solve, no ba_hip_finalize needed. */
int ba_hip_set_imu_noise(ba_hip_engine* e, const double r6[6], const double rb6[6]);
/* Options::calculate_inertial_covariance_once (BundleAdjuster.h:106, parallel_algos.h:189-205):
 * the integration covariance and the bias Jacobian of an inertial residual are computed in its
 * first linearisation and reused afterwards (they survive later ba_hip_set_imu_residuals calls
 * as long as the residual list only grows).  reset != 0 forgets the stored ones (Init()). */
int ba_hip_set_inertial_covariance_once(ba_hip_engine* e, int on, int reset);
int ba_hip_set_gravity(ba_hip_engine* e, const double g3[3]); /* BundleAdjuster.h:243-252 */
/* Build the device-side structure: observation list sorted by landmark (CSR),
 * pose-landmark incidences, per-pose-pair gather lists for the reduced matrix. */
int ba_hip_finalize(ba_hip_engine* e);

/* ---- one Solve() ------------------------------------------------------------------ */
/* BundleAdjuster.cpp:288-296: x_s = T_sw(ref) x_w, normalised (lm_dim == 1).  May be called
 * again on a finalized engine whose graph did not change (the reference's "Solve may be called
 * repeatedly", BundleAdjuster.h:549-551): the state, Huber-compounded unary weights, reliability
 * flags and frozen inertial covariances the previous Solve() left on the device are kept. */
int ba_hip_begin_solve(ba_hip_engine* e);
/* BundleAdjuster.cpp:1237-1330 decides the masks on the host; bit i of masks[p] set =
 * parameter i of pose p is regularised (Jacobian column zeroed, S(idx,idx) = 1e6,
 * BundleAdjuster.cpp:587-598,1622-1629). */
int ba_hip_set_pose_masks(ba_hip_engine* e, uint32_t n, const uint16_t* masks);
/* BuildProblem + J^T J + Schur complement (BundleAdjuster.cpp:1166-1803, 327-598):
 * leaves S, rhs_p_sc, rhs_p, rhs_l, V^-1, W on the device; returns the error sums
 * BuildProblem computes (proj_error_ etc.). */
int ba_hip_linearize(ba_hip_engine* e, ba_hip_errors* out);
/* CalculateGn + GetLandmarkDelta (BundleAdjuster.cpp:748-833, 709-744): dense Cholesky
 * of S, delta_p, then delta_l = V^-1 (rhs_l - W^T delta_p). */
int ba_hip_solve_gn(ba_hip_engine* e);
/* BundleAdjuster.cpp:858-925 + the norms/dots of :971-1004 */
int ba_hip_dogleg_terms(ba_hip_engine* e, int gn_available, ba_hip_dogleg_scalars* out);
/* step = coef_rhs * (rhs_p_, rhs_l_) + coef_gn * delta_gn  (BundleAdjuster.cpp:923-925,
 * 947-950, 985, 1015-1017, 1108-1110) */
int ba_hip_compose_step(ba_hip_engine* e, double coef_rhs, double coef_gn, ba_hip_step_norms* out);
/* ApplyUpdate (BundleAdjuster.cpp:21-140) into the alternate state buffer; the previous
 * state is kept as the rollback snapshot (replaces the deep copies of :1022-1028). */
int ba_hip_apply_step(ba_hip_engine* e);
/* restore the snapshot (BundleAdjuster.cpp:1060-1068, 1139-1149) */
int ba_hip_rollback(ba_hip_engine* e);
/* EvaluateResiduals (BundleAdjuster.cpp:144-274) at the current state */
int ba_hip_eval_residuals(ba_hip_engine* e, ba_hip_errors* out);
/* BundleAdjuster.cpp:672-678: x_w = T_ws(ref) x_s (lm_dim == 1) */
int ba_hip_end_solve(ba_hip_engine* e);

/* ---- results / debug taps ----------------------------------------------------------- */
int ba_hip_get_poses(ba_hip_engine* e, double* t_wp7, double* v_w3, double* b6);
int ba_hip_get_landmarks(ba_hip_engine* e, double* x_w4);
int ba_hip_get_landmark_flags(ba_hip_engine* e, uint8_t* is_reliable, uint32_t* num_outliers);
uint32_t ba_hip_num_pose_params(const ba_hip_engine* e);   /* PoseSize * active poses */
uint32_t ba_hip_num_calib_params(const ba_hip_engine* e);  /* 0, or 6 with ba_hip_set_calibration(.., do_tvs) */
/* Options::calculate_calibration_marginals (BundleAdjuster.cpp:771-784): the K x K block of S^-1
 * that belongs to the calibration unknowns (row-major), from the factor left by the last
 * ba_hip_solve_gn — no extra solves.  Replicated / single-shard solve only. */
int ba_hip_get_calibration_marginals(ba_hip_engine* e, double* cov_kxk);
/* T_vs of every camera (7 doubles each) as the engine currently holds them: the values of
 * ba_hip_set_cameras, moved by the calibration steps applied since. */
int ba_hip_get_cameras(ba_hip_engine* e, uint32_t n, double* t_vs7);  /* n = camera count, checked */
uint32_t ba_hip_num_lm_params(const ba_hip_engine* e);
/* s_ as the reference leaves it (BundleAdjuster.cpp:473-477,587-598): dense n x n
 * row-major (n = pose + calibration unknowns); block (i,j) kept only for i <= j when
 * use_triangular_matrices (the calibration border counts as the last block: S_pk is kept, S_kp not,
 * :513-518).  The vectors of get_rhs / get_delta_gn / get_step sized by the pose unknowns carry the
 * calibration entries behind them. */
int ba_hip_get_S(ba_hip_engine* e, double* s_nxn);
int ba_hip_get_rhs(ba_hip_engine* e, double* rhs_p_sc, double* rhs_p, double* rhs_l);
int ba_hip_get_delta_gn(ba_hip_engine* e, double* delta_p, double* delta_l);
int ba_hip_get_step(ba_hip_engine* e, double* delta_p, double* delta_l);
int ba_hip_get_proj_weights(ba_hip_engine* e, double* weight); /* per residual id */
/* residual vectors z - pi (2 doubles per residual id) at the current state, i.e. what
 * ProjectionResidual::residual holds after a Solve() (BundleAdjuster.cpp:155-181) */
int ba_hip_get_proj_residuals(ba_hip_engine* e, double* residual2);
/* The weighted Jacobian blocks and residuals of the last ba_hip_linearize as the reference stores
 * them in j_pr_, j_l_ and r_pr_ (BundleAdjuster.cpp:1636-1642, 1795-1796, 1384-1385): per residual id
 * sqrt(w) dz_dx_meas (2x6), sqrt(w) dz_dx_ref (2x6, LmSize 1), sqrt(w) dz_dlm (2xLm) and
 * sqrt(w) r (2), masked columns zeroed.  Read back from the factor rows — the device never holds a
 * Jacobian MATRIX; the host class writes the reference's j_pr.txt / j_l.txt / r_pr.txt from this
 * (write_reduced_camera_matrix, BundleAdjuster.cpp:608-616).  Any pointer may be NULL. */
int ba_hip_get_proj_jacobians(ba_hip_engine* e, double* j_meas12, double* j_ref12, double* j_lm, double* r2);
/* Calibration instantiations: sqrt(w) dz_dtvs (2x6) per residual id, what the reference stores in
 * j_kpr_ (BundleAdjuster.cpp:1769-1783) and writes to j_kpr.txt (:619-622). */
int ba_hip_get_calib_jacobians(ba_hip_engine* e, double* j_k12);
int ba_hip_get_timers(ba_hip_engine* e, ba_hip_timers* t);
/* Test tap for systems too large to download (S is 28.8 GB at BASELINE.json configs[3]): forms
 * || S delta_gn - rhs_p_sc || and || rhs_p_sc || ON THE DEVICE from the copy of S kept before the
 * in-place factorisation (needs ba_hip_options.keep_reduced_system) and the Gauss-Newton pose
 * step of the last ba_hip_solve_gn — i.e. checks what CalculateGn promises,
 * S delta = rhs (BundleAdjuster.cpp:748-833).  Single shard only. */
int ba_hip_check_solve(ba_hip_engine* e, double* residual_norm, double* rhs_norm);
/* Residual vectors of the inertial residuals at the current state (ImuResidualT::residual after
 * EvaluateResiduals, BundleAdjuster.cpp:225-256): 15 doubles per residual in residual-id order,
 * the first PoseSize of them used (9: translation, rotation, velocity; 15: + biases). */
int ba_hip_get_imu_residuals(ba_hip_engine* e, double* residual15);
/* Mahalanobis distance of every inertial residual at the last evaluation (ImuResidualT::
 * mahalanobis_distance, BundleAdjuster.cpp:252-254), residual-id order; summed over the
 * conditioning residuals by SolutionSummary::cond_inertial_error (:680-690). */
int ba_hip_get_imu_errors(ba_hip_engine* e, double* mahalanobis);
/* cumulative Huber scale of every unary residual's cov^-1 (the reference multiplies
 * cov_inv in place every BuildProblem, BundleAdjuster.cpp:1469) */
int ba_hip_get_unary_scales(ba_hip_engine* e, double* scale);

/* Per-kernel device time, accumulated since ba_hip_set_profiling(e, 1): HIP events on
 * the engine's stream around every launch of the three hot kernels (used by bench.py's
 * roofline; costs two events per launch, so it is off by default). */
typedef struct {
  uint32_t syrk_launches, gather_launches /* k_assemble_tiles */, landmarks_launches /* k_linearize */, imu_launches;
  double syrk_ms, gather_ms, landmarks_ms;
  double syrk_flops;       /* algorithmic flops of those k_syrk launches */
  double imu_ms;           /* k_imu, the BuildProblem launch (parallel_algos.h:178-358) */
  double pose_blocks_ms;   /* k_pose_blocks: diagonal blocks + right-hand sides */
  uint32_t pose_blocks_launches, reserved;
} ba_hip_kernel_stats;
/* Sizes of the static structure ba_hip_finalize built (for byte accounting in benchmarks). */
typedef struct {
  uint64_t poses_active, landmarks_active, observations;
  uint64_t incidences;        /* (active pose, active landmark) pairs */
  uint64_t factor_rows;       /* 48-byte rows written by the linearisation kernel per iteration */
  uint64_t pair_blocks;       /* off-diagonal pose-pair blocks of S with at least one term */
  uint64_t pair_entries;      /* rank-1 terms summed into those blocks */
  uint64_t tiles_lower;       /* 64x64 tiles of the lower triangle incl. diagonal */
  uint64_t tiles_S, tiles_L;  /* of those: structurally nonzero in S / in its factor (after fill) */
  uint64_t tile_refs;         /* (tile, block) references of the tile assembly (straddling blocks count twice) */
  uint64_t pose_entries;      /* terms of the diagonal blocks / right-hand sides (12 bytes each) */
  uint64_t linearize_waves;   /* wavefronts of the linearisation kernel */
  uint64_t factor_tile_products; /* 64x64x64 tile products of the tile-sparse LDL^T on the factor's pattern (trailing
                                    updates + substitutions; x 2 * 64^3 flop each): the flops of one reduced solve */
} ba_hip_structure_stats;
int ba_hip_get_structure_stats(ba_hip_engine* e, ba_hip_structure_stats* out);
/* Experiment knobs for scratch/ micro-benchmarks (kernel variants with identical results): key 1 =
 * variant of the tile assembly kernel (0, 1, 2), key 2 = launch order of its tiles (0 row-major,
 * 1 XCD-aware columns), key 3 = 1: write every lower tile instead of the factor's pattern only,
 * key 4 = linearisation variant (0 LDS-staged rows, 1 direct stores), key 5 = 1: build the static
 * lists on the host (structure.h) instead of on the device at the next ba_hip_finalize, key 6 = variant
 * of the inertial linearisation (-1 chosen by residual count, 0 one lane per sample / per residual, 1 a
 * wavefront per residual over the lane-per-sample step pass, 2 the single-pass form with the step Jacobians
 * inside, 4 a wavefront per residual AND per sample). */
int ba_hip_debug_set(ba_hip_engine* e, int key, int value);
int ba_hip_set_profiling(ba_hip_engine* e, int enable);
int ba_hip_get_kernel_stats(ba_hip_engine* e, ba_hip_kernel_stats* out);

/* ---- raw device access for drivers that own streams/collectives ---------------------- */
/* Device pointer + element count of the buffers whose cross-shard SUM defines the
 * iteration: 0 = S (lower storage incl. rhs row), 1 = scalar block. */
int ba_hip_device_buffer(ba_hip_engine* e, int which, void** dev_ptr, size_t* num_doubles);
/* All-reduce hook: called on the engine's host thread, after the engine has drained its
 * stream, for every buffer that must be summed over shards (doubles or uint64 counts).
 * dtype: 0 = f64, 1 = u64.  Must return 0 on success.  NULL = single shard. */
typedef int (*ba_hip_allreduce_fn)(void* ctx, void* dev_ptr, size_t count, int dtype);
int ba_hip_set_allreduce(ba_hip_engine* e, ba_hip_allreduce_fn fn, void* ctx, int rank, int nranks);
/* Cross-shard SUM of `count` host values (dtype 0 = f64, 1 = u64) through the installed all-reduce
 * (hook or native communicator); a no-op on a single shard.  Used by the host class for the few
 * graph statistics the gauge masks depend on (BundleAdjuster.cpp:1237-1330): per-pose residual
 * counts must be GLOBAL counts when the residuals are sharded. */
int ba_hip_allreduce_host(ba_hip_engine* e, void* host, size_t count, int dtype);
/* Collectives hook (optional, on top of the all-reduce hook): with it the dense reduced solve
 * is DISTRIBUTED over the shards instead of replicated:
 * S is cut into blocks of G x G 64-tiles (G = 4 / 8 / 16 by system size) owned by the ranks through a small class
 * table (ba_amd/csrc/dist_plan.h: "tri" / "grid" / "col" / "row" layouts); the partial S of every shard reaches the
 * block owners point to point (only the rectangles the shard's own pattern touches), the owner of a diagonal block
 * factorises the panel's square and broadcasts it, the block rows under it travel point to point to the ranks that
 * multiply them, and each rank applies the trailing updates to the tiles it owns.
 *   op 1 = broadcast `count` doubles at dev_ptr from rank `root`;
 *   op 2 = reduce-scatter (sum), in place: dev_ptr holds nranks chunks of `count` doubles, on
 *          return chunk `rank` (at dev_ptr + rank * count) holds the sum over ranks of that chunk
 *          (BA_HIP_DENSE_SCATTER=1 and single-rank communicators only);
 * Same calling conventions as the all-reduce hook.  NULL = replicated solve. */
typedef int (*ba_hip_collective_fn)(void* ctx, int op, void* dev_ptr, size_t count, int root);
/* (ops 3 / 4 / 5 of the hook, used by the distributed solve since round 3: op 3 = send `count` doubles at
 * dev_ptr to rank `root`, must not block on the receiver (copy or defer); op 4 = receive `count` doubles from rank
 * `root`; op 5 (dev_ptr NULL) = end of one exchange: every send handed over since the last op 5 has been started.
 * The engine issues all sends of an exchange, then all receives, then op 5.) */
int ba_hip_set_collectives(ba_hip_engine* e, ba_hip_collective_fn fn, void* ctx);
/* Native communicator: the engine loads librccl itself (one process per GPU, RCCL over xGMI) and
 * runs every cross-shard sum and the collectives of the distributed reduced solve on an
 * engine-owned ncclComm — a C++ user of ba::BundleAdjuster needs no torch and no hooks.
 *   ba_hip_comm_unique_id  rank 0 creates the 128-byte id and hands it to the other ranks out of band
 *   ba_hip_comm_init       collective over all ranks (ncclCommInitRank on the engine's device);
 *                          installs the native all-reduce + collectives (replacing any hook) and sets
 *                          rank / nranks.  nranks == 1 is allowed and still drives the sharded code
 *                          paths through RCCL (test on a one-GPU box)
 *   ba_hip_comm_destroy    back to a single unsharded engine
 * With it the per-panel broadcast of the distributed solve is enqueued in stream order (no host
 * round trip per panel). */
int ba_hip_comm_unique_id(void* id128);
int ba_hip_comm_init(ba_hip_engine* e, const void* id128, int rank, int nranks);
int ba_hip_comm_destroy(ba_hip_engine* e);
/* Bytes this rank moved through the communicator(s) since the last reset: the chain stream carries what
 * the next panel waits for (factorised squares, the block row under them, the backward substitution's
 * partial sums), the side stream the rest of every panel (consumed by the bulk trailing updates). */
typedef struct {
  double chain_bytes_sent, chain_bytes_recv;   /* square broadcasts (root: sent, others: received) + urgent rows */
  double side_bytes_sent, side_bytes_recv;     /* the remaining rows of every panel, point to point */
  double reduce_scatter_bytes;                 /* partial S onto the tile owners (send buffer size per call) */
  double allreduce_bytes;                      /* every all-reduced buffer (rhs, scalars, histograms, patterns) */
  uint64_t chain_messages, side_messages, factorisations;
} ba_hip_comm_stats;
int ba_hip_get_comm_stats(ba_hip_engine* e, ba_hip_comm_stats* out);
int ba_hip_reset_comm_stats(ba_hip_engine* e);
/* Byte accounting of the distributed solve's message plan WITHOUT a device (pure host): nz_lower is the
 * nblk x nblk row-major byte pattern of the factor's 64x64 tiles (NULL = dense), layout one of "auto",
 * "tri", "grid", "col", "row" (ba_amd/csrc/dist_plan.h), kout the panel width in tiles (0 = the engine's
 * choice for nblk).  Returns 0, or -1 if the layout does not exist for nranks. */
typedef struct {
  double factor_bytes;                         /* the whole factor: what a 1-D panel broadcast hands to every rank */
  double chain_recv_max, chain_recv_total;     /* per factorisation: busiest receiver / sum over ranks */
  double side_recv_max, side_recv_total;
  double chain_sent_total, side_sent_total;
  double recv_max;                             /* chain + side of the busiest receiver */
  double backward_allreduce_bytes;
  uint32_t messages_chain, messages_side, panels, ranks, classes, kout;
} ba_hip_dist_plan_stats_t;
int ba_hip_dist_plan_stats(uint32_t nblk, const uint8_t* nz_lower, int nranks, const char* layout, uint32_t kout,
                           ba_hip_dist_plan_stats_t* out);
/* Tile pattern of the factor as the engine holds it (nblk x nblk bytes, lower): for the accounting above. */
int ba_hip_get_factor_tile_pattern(ba_hip_engine* e, uint32_t nblk, uint8_t* nz_lower);
/* 1 if the next ba_hip_solve_gn will run the distributed solve, 0 if replicated / single. */
int ba_hip_solve_is_distributed(ba_hip_engine* e);

/* ---- stand-alone kernels exposed for tests and benchmarks ------------------------- */
/* Dense Cholesky solve of an SPD system given by its LOWER triangle (row-major n x n,
 * host memory): x = A^-1 b.  Runs the same kernels ba_hip_solve_gn uses. */
int ba_hip_dense_solve(ba_hip_engine* e, uint32_t n, const double* a_lower, const double* b, double* x);
/* exact k-th smallest (0-based) of n non-negative doubles — the device selection behind
 * the Huber sigma (std::nth_element at floor(N/2), BundleAdjuster.cpp:1356-1358) */
int ba_hip_select_kth(ba_hip_engine* e, uint32_t n, const double* values, uint32_t k, double* out);

#ifdef __cplusplus
}
#endif
#endif /* BA_HIP_H */
