/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement of arpg/ba's BundleAdjuster<> Gauss-Newton
 * path (oracle/ba_oracle.cpp).  It mirrors the reference's public API
 * (/root/reference/include/ba/BundleAdjuster.h:177-631) call for call so that
 * parity tests read like reference usage.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (ba_amd/, include/) never does.
 *
 * PARITY UNPINNED: the reference cannot be built in this image (Eigen, Sophus,
 * Calibu, TBB absent) and holds no golden numeric vectors; the restatement is
 * pinned by finite-difference Jacobian checks at the reference's thresholds and by
 * dense-algebra identities (tests/test_oracle_*.py).
 *
 * 7-vectors describing a rigid transform are [tx,ty,tz,qx,qy,qz,qw]
 * (translation, then quaternion coefficients in Eigen order).
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ba orc_ba;

/* ba::Options<double>, BundleAdjuster.h:72-107 (same defaults via orc_default_options). */
typedef struct {
  double trust_region_size;
  double gyro_sigma, accel_sigma, gyro_bias_sigma, accel_bias_sigma;
  double projection_outlier_threshold;
  double error_change_threshold, param_change_threshold;
  uint32_t dogleg_max_inner_iterations;
  int apply_results, use_dogleg, use_triangular_matrices, use_sparse_solver;
  int regularize_biases_in_batch, enable_auto_regularization;
  int use_robust_norm_for_proj_residuals, use_robust_norm_for_inertial_residuals;
} orc_options;

/* ba::SolutionSummary<double>, BundleAdjuster.h:48-70 (+ the four error sums of GetErrors). */
typedef struct {
  uint32_t num_proj_residuals, num_inertial_residuals;
  uint32_t num_cond_proj_residuals, num_cond_inertial_residuals;
  double cond_proj_error, cond_inertial_error, proj_error, inertial_error;
  double delta_norm, pre_solve_norm, post_solve_norm;
  int result; /* OptimizationResult */
  double unary_error, binary_error;
  uint32_t iterations_run;
  double trust_region_size;
} orc_summary;

void orc_default_options(orc_options* o);

orc_ba* orc_create(int lm_dim, int pose_dim);
/* BundleAdjuster<Scalar, LmSize, PoseSize, CalibSize, DoTvs> (BundleAdjuster.h:110-134).  Restated:
   calib_size = 0 with do_tvs = 0 | 1, and calib_size = 4 (the pinhole parameters fx, fy, u0, v0 of
   camera 0, BundleAdjuster.cpp:46-69) or 5 (fx, fy, u0, v0, w of a FOV camera 0 — the reference's
   SelfCalBundleAdjuster, BundleAdjuster.h:758-759) with do_tvs = 0 — LmSize 1 only; anything else
   returns NULL.
   The reduced system is (n + kCalibDim)^2: the calibration unknowns follow the pose unknowns
   (BundleAdjuster.cpp:316-322, 493-583); with do_tvs they are the decoupled update of the
   extrinsics T_vs of camera 0 (:72-83). */
orc_ba* orc_create_calib(int lm_dim, int pose_dim, int calib_size, int do_tvs);
void orc_destroy(orc_ba* h);
void orc_init(orc_ba* h, const orc_options* o);
void orc_set_gravity(orc_ba* h, const double g[3]);
uint32_t orc_add_camera(orc_ba* h, const double params[4], const double t_vs[7]);
uint32_t orc_add_pose(orc_ba* h, const double t_wp[7], const double v_w[3],
                      const double b[6], int is_active, double time);
uint32_t orc_add_landmark(orc_ba* h, const double x_w[4], uint32_t ref_pose_id,
                          uint32_t ref_cam_id, int is_active);
uint32_t orc_add_projection_residual(orc_ba* h, const double z[2], uint32_t meas_pose_id,
                                     uint32_t landmark_id, uint32_t cam_id, double weight);
uint32_t orc_add_unary_constraint(orc_ba* h, uint32_t pose_id, const double t_wv[7],
                                  const double cov[36], int use_rotation);
uint32_t orc_add_binary_constraint(orc_ba* h, uint32_t pose1_id, uint32_t pose2_id,
                                   const double t_12[7], const double cov[36],
                                   double weight, int use_rotation);
/* meas: n rows of [wx,wy,wz,ax,ay,az,time] */
uint32_t orc_add_imu_residual(orc_ba* h, uint32_t pose1_id, uint32_t pose2_id,
                              const double* meas, uint32_t n, double weight);
void orc_regularize_pose(orc_ba* h, uint32_t pose_id, int translation, int gravity,
                         int bias, int rotation);
void orc_set_root_pose_id(orc_ba* h, uint32_t id);

/* bulk adders (same semantics as n single calls; ids returned in out_ids if non-null) */
/* Options::use_per_pose_cam_params (BundleAdjuster.h:96, parallel_algos.h:54-57,
 * BundleAdjuster.cpp:162-176): every projection residual is evaluated with the pinhole
 * intrinsics stored on its MEASUREMENT pose (AddPose overload with cam_params, BundleAdjuster.h:292).
 * orc_set_use_per_pose_cam_params returns 1 if a pose has no parameters. */
/* Options::calculate_inertial_covariance_once (BundleAdjuster.h:106, parallel_algos.h:189-205) */
/* SetImuCalibration (BundleAdjuster.h:567): the noise diagonals imu_.r / imu_.r_b (after orc_init) */
void orc_set_imu_noise(orc_ba* h, const double r6[6], const double rb6[6]);
void orc_set_calculate_inertial_covariance_once(orc_ba* h, int on);
void orc_set_pose_cam_params(orc_ba* h, uint32_t pose_id, const double params4[4]);
int orc_set_use_per_pose_cam_params(orc_ba* h, int on);
void orc_add_poses(orc_ba* h, uint32_t n, const double* t_wp, const double* v_w,
                   const double* b, const uint8_t* is_active, const double* time);
void orc_add_landmarks(orc_ba* h, uint32_t n, const double* x_w, const uint32_t* ref_pose_id,
                       const uint32_t* ref_cam_id, const uint8_t* is_active);
void orc_add_projection_residuals(orc_ba* h, uint32_t n, const double* z,
                                  const uint32_t* meas_pose_id, const uint32_t* landmark_id,
                                  const uint32_t* cam_id, const double* weight,
                                  uint32_t* out_ids);

void orc_solve(orc_ba* h, uint32_t max_iter, double gn_damping, int error_increase_allowed);

/* results */
uint32_t orc_num_poses(const orc_ba* h);
uint32_t orc_num_landmarks(const orc_ba* h);
uint32_t orc_num_proj_residuals(const orc_ba* h);
void orc_get_pose(const orc_ba* h, uint32_t id, double t_wp[7], double v_w[3], double b[6]);
void orc_get_poses(const orc_ba* h, double* t_wp, double* v_w, double* b);
void orc_get_landmark(const orc_ba* h, uint32_t id, double x_w[4]);
void orc_get_landmarks(const orc_ba* h, double* x_w);
int orc_is_landmark_reliable(const orc_ba* h, uint32_t id);
double orc_landmark_outlier_ratio(const orc_ba* h, uint32_t id);
void orc_get_summary(const orc_ba* h, orc_summary* s);

/* parity/debug taps: state of the LAST executed Solve iteration */
uint32_t orc_num_pose_params(const orc_ba* h);   /* n = PoseDim * active poses */
uint32_t orc_num_lm_params(const orc_ba* h);     /* LmDim * active landmarks */
void orc_get_S(const orc_ba* h, double* s_nxn);  /* reduced matrix as the reference
                                                    leaves it in s_ (block-upper when
                                                    use_triangular_matrices) */
void orc_get_rhs(const orc_ba* h, double* rhs_n);        /* rhs_p_sc */
void orc_get_rhs_p(const orc_ba* h, double* rhs_n);      /* rhs_p_ (before Schur) */
void orc_get_rhs_l(const orc_ba* h, double* rhs_l);
void orc_get_delta_p(const orc_ba* h, double* d);        /* applied pose step */
void orc_get_delta_l(const orc_ba* h, double* d);
uint32_t orc_num_calib_params(const orc_ba* h);          /* kCalibDim: orc_get_S / orc_get_rhs are (n + kCalibDim) wide */
void orc_get_delta_k(const orc_ba* h, double* d);        /* applied calibration step */
void orc_get_rhs_k(const orc_ba* h, double* r);          /* rhs_k_ (before Schur) */
void orc_get_camera_pose(const orc_ba* h, uint32_t cam_id, double t_vs[7]);
void orc_get_proj_tvs_jacobians(const orc_ba* h, double* j_tvs); /* dz_dtvs 2x6 per residual id */
int orc_get_calibration_marginals(const orc_ba* h, double* cov_kxk); /* returns kCalibDim */
void orc_get_camera_params(const orc_ba* h, uint32_t cam_id, double params[4]);
void orc_get_proj_calib_jacobians(const orc_ba* h, double* j_k); /* 2 x kCalibDim per residual id, j_kpr_ layout */
/* Transfer(T_ba, pix, rho) = Project(R Unproject(pix) + rho t) of the pinhole model and its 2x4
   parameter Jacobian (jac8 may be NULL) — finite-difference pin of dTransfer_dparams */
void orc_math_transfer(const double params[4], const double t_ba[7], const double pix[2], double rho,
                       double out[2], double jac8[8]);
/* the same for the FOV camera (fx, fy, u0, v0, w): 2x5 Jacobian; and Project / dProject_dP /
   Unproject(Project(P)) of that model */
void orc_math_transfer_fov(const double params[5], const double t_ba[7], const double pix[2], double rho,
                           double out[2], double jac10[10]);
void orc_math_fov_project(const double params[5], const double P[3], double pix[2], double dpix_dP[6], double ray[3]);
/* camera cam_id becomes a calibu::FovCamera with distortion parameter w (default: LinearCamera) */
void orc_set_camera_fov(orc_ba* h, uint32_t cam_id, double w);
double orc_get_camera_fov(const orc_ba* h, uint32_t cam_id);
void orc_get_proj_weights(const orc_ba* h, double* w);   /* per residual id */
void orc_get_proj_residuals(const orc_ba* h, double* r2);/* per residual id, 2 each */
void orc_get_imu_residuals(const orc_ba* h, double* r15);/* ImuResidualT::residual, 15 each (first PoseSize used) */
uint32_t orc_num_imu_residuals(const orc_ba* h);
/* per-residual Jacobians of the last BuildProblem (dz_dx_meas 2x6, dz_dx_ref 2x6,
   dz_dlm 2xLm), row-major, unmasked/unweighted as stored in the residual. */
void orc_get_proj_jacobians(const orc_ba* h, double* j_meas, double* j_ref, double* j_lm);
void orc_get_imu_jacobians(const orc_ba* h, uint32_t id, double* dz_dx1, double* dz_dx2,
                           double* cov_inv, double* residual); /* 15x15,15x15,15x15,15 */
void orc_get_binary_jacobians(const orc_ba* h, uint32_t id, double* dz_dx1, double* dz_dx2,
                              double* residual);
void orc_get_unary_jacobian(const orc_ba* h, uint32_t id, double* dz_dx, double* residual);

/* phase timers of the last Solve (seconds, summed over iterations), named after the
   reference's StartTimer/PrintTimer sites (Utils.h:51-62). */
typedef struct {
  double build_problem, j_evaluation_proj, jtj, schur_complement, solve,
         back_substitution, evaluate_residuals, apply_update, total;
} orc_timers;
void orc_get_timers(const orc_ba* h, orc_timers* t);

/* stand-alone math taps for finite-difference tests */
void orc_math_dlog_dq(const double q[4], double out3x4[12]);
void orc_math_so3_log(const double q[4], double out[3]);
void orc_math_so3_exp(const double w[3], double q[4]);
void orc_math_exp_decoupled(const double t[7], const double x[6], double out[7]);
void orc_math_log_decoupled(const double a[7], const double b[7], double out[6]);
void orc_math_se3_mul(const double a[7], const double b[7], double out[7]);
void orc_math_se3_inv(const double a[7], double out[7]);
void orc_math_dense_solve_upper(uint32_t n, const double* s, const double* rhs, double* x);
/* threads of the dense LDL^T (process-wide).  1 = reference-faithful (Eigen's LDLT / SimplicialLDLT
   run on one thread, BundleAdjuster.cpp:752-799); > 1 = the "best-effort CPU" baseline mode. */
void orc_set_num_threads(int n);
int orc_get_num_threads(void);
/* IntegrateResidual (Types.h:662-738): pose [t7,v3], returns [t7,v3] and optional
   10x6 bias Jacobian and 10x10 covariance. */
void orc_math_integrate(const double pose_t[7], const double v[3], const double* meas,
                        uint32_t n, const double bg[3], const double ba[3],
                        const double g[3], const double r6[6], double out_t[7],
                        double out_v[3], double* dpose_db_10x6, double* c_10x10);
/* the same integration returning dpose_db (10x6), dpose_dpose (10x10, Types.h:716-718) and the covariance */
void orc_math_integrate_jacobians(const double pose_t[7], const double v[3], const double* meas, uint32_t n,
                                  const double bg[3], const double ba[3], const double g[3], const double r6[6],
                                  double* dpose_db, double* dpose_dpose, double* c);
/* the Utils.h helpers by the op codes of ba_hip_lie (include/ba_hip.h); returns the doubles written */
int orc_math_lie(int op, const double* a, const double* b, double* out);

#ifdef __cplusplus
}
#endif
#endif
