/*
 * ba_capi.h — flat C view of the C++ host class ba::BundleAdjuster<> (include/ba/
 * BundleAdjuster.h) for callers without a C++ toolchain: Python/ctypes in this repo's
 * tests and bench.py.  One handle = one ba::BundleAdjuster<double, lm_dim, pose_dim, 0>.
 * Every call forwards to the member of the same name, whose semantics follow the
 * reference API (/root/reference/include/ba/BundleAdjuster.h:177-631).  7-vectors are
 * [tx,ty,tz,qx,qy,qz,qw].
 */
#ifndef BA_CAPI_H
#define BA_CAPI_H
#include <stdint.h>
#include "ba_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_adjuster ba_adjuster;

typedef struct { /* ba::Options<double>, reference BundleAdjuster.h:72-107 */
  double trust_region_size;
  double gyro_sigma, accel_sigma, gyro_bias_sigma, accel_bias_sigma;
  double projection_outlier_threshold;
  double error_change_threshold, param_change_threshold;
  uint32_t dogleg_max_inner_iterations;
  int32_t apply_results, use_dogleg, use_triangular_matrices, use_sparse_solver;
  int32_t regularize_biases_in_batch, enable_auto_regularization;
  int32_t use_robust_norm_for_proj_residuals, use_robust_norm_for_inertial_residuals;
  int32_t write_reduced_camera_matrix; /* 1: keep S readable through the taps; 2: also write s.txt / rhs.txt
                                          (the reference's dump, BundleAdjuster.cpp:600-606) */
  int32_t device;
  double factorization_pivot_tolerance; /* extension, 0 = off: ba::Options::factorization_pivot_tolerance */
  int32_t calculate_calibration_marginals; /* reference BundleAdjuster.h:95 (do_tvs adjusters) */
  int32_t reserved;
} ba_options;

typedef struct { /* ba::SolutionSummary<double> + GetErrors, reference :48-70,593-602 */
  uint32_t num_proj_residuals, num_inertial_residuals;
  uint32_t num_cond_proj_residuals, num_cond_inertial_residuals;
  double proj_error, inertial_error, unary_error, binary_error;
  double delta_norm, pre_solve_norm, post_solve_norm;
  int32_t result; /* ba::OptimizationResult */
  uint32_t iterations_run;
  double trust_region_size;
} ba_summary;

void ba_default_options(ba_options* o);
ba_adjuster* ba_adjuster_create(int lm_dim, int pose_dim);
/* ba::BundleAdjuster<double, lm_dim, pose_dim, calib_size, do_tvs>, lm_dim 1: do_tvs (the extrinsics
 * of camera 0 become six more unknowns) or calib_size 4 / 5 (the parameters of camera 0 — a LinearCamera
 * / a FovCamera — become four / five more), not both.  NULL otherwise. */
ba_adjuster* ba_adjuster_create_calib(int lm_dim, int pose_dim, int calib_size, int do_tvs);
void ba_adjuster_destroy(ba_adjuster* a);
void ba_adjuster_init(ba_adjuster* a, const ba_options* o);
void ba_adjuster_set_gravity(ba_adjuster* a, const double g[3]);
uint32_t ba_adjuster_add_camera(ba_adjuster* a, const double params[4], const double t_vs[7]);
/* a calibu::FovCamera: params (fx, fy, u0, v0, w) */
uint32_t ba_adjuster_add_camera_fov(ba_adjuster* a, const double params[5], const double t_vs[7]);
uint32_t ba_adjuster_add_pose(ba_adjuster* a, const double t_wp[7], const double v_w[3],
                              const double b[6], int is_active, double time);
uint32_t ba_adjuster_add_landmark(ba_adjuster* a, const double x_w[4], uint32_t ref_pose_id,
                                  uint32_t ref_cam_id, int is_active);
uint32_t ba_adjuster_add_projection_residual(ba_adjuster* a, const double z[2], uint32_t meas_pose_id,
                                             uint32_t landmark_id, uint32_t cam_id, double weight);
uint32_t ba_adjuster_add_unary_constraint(ba_adjuster* a, uint32_t pose_id, const double t_wv[7],
                                          const double cov[36], int use_rotation);
uint32_t ba_adjuster_add_binary_constraint(ba_adjuster* a, uint32_t pose1_id, uint32_t pose2_id,
                                           const double t_12[7], const double cov[36], double weight,
                                           int use_rotation);
uint32_t ba_adjuster_add_imu_residual(ba_adjuster* a, uint32_t pose1_id, uint32_t pose2_id,
                                      const double* meas7, uint32_t n, double weight);
void ba_adjuster_regularize_pose(ba_adjuster* a, uint32_t pose_id, int translation, int gravity,
                                 int bias, int rotation);
void ba_adjuster_set_root_pose_id(ba_adjuster* a, uint32_t id);
/* bulk adders = n single calls */
/* Options::use_per_pose_cam_params + PoseT::cam_params (reference BundleAdjuster.h:96,292-323):
 * pinhole [fx,fy,u0,v0] for every pose added so far (n must equal the number of poses) and the
 * option switched on; n = 0 switches it off.  Returns 0 on success. */
int ba_adjuster_set_pose_cam_params(ba_adjuster* a, uint32_t n, const double* params4);
/* Options::calculate_inertial_covariance_once (reference BundleAdjuster.h:106); call after init */
void ba_adjuster_set_calculate_inertial_covariance_once(ba_adjuster* a, int on);
/* SetImuCalibration with the noise diagonals r / r_b replaced (reference BundleAdjuster.h:566-567) */
void ba_adjuster_set_imu_noise(ba_adjuster* a, const double r6[6], const double rb6[6]);
void ba_adjuster_add_poses(ba_adjuster* a, uint32_t n, const double* t_wp, const double* v_w,
                           const double* b, const uint8_t* is_active, const double* time);
void ba_adjuster_add_landmarks(ba_adjuster* a, uint32_t n, const double* x_w,
                               const uint32_t* ref_pose_id, const uint32_t* ref_cam_id,
                               const uint8_t* is_active);
void ba_adjuster_add_projection_residuals(ba_adjuster* a, uint32_t n, const double* z,
                                          const uint32_t* meas_pose_id, const uint32_t* landmark_id,
                                          const uint32_t* cam_id, const double* weight,
                                          uint32_t* out_ids);
void ba_adjuster_solve(ba_adjuster* a, uint32_t max_iter, double gn_damping, int error_increase_allowed);
uint32_t ba_adjuster_num_poses(const ba_adjuster* a);
uint32_t ba_adjuster_num_landmarks(const ba_adjuster* a);
uint32_t ba_adjuster_num_proj_residuals(const ba_adjuster* a);
void ba_adjuster_get_poses(const ba_adjuster* a, double* t_wp, double* v_w, double* b);
void ba_adjuster_get_landmarks(const ba_adjuster* a, double* x_w);
int ba_adjuster_is_landmark_reliable(const ba_adjuster* a, uint32_t id);
double ba_adjuster_landmark_outlier_ratio(const ba_adjuster* a, uint32_t id);
/* GetProjectionResidual(id): out11 = z(2), residual(2), weight, orig_weight, mahalanobis_distance,
 * x_meas_id, x_ref_id, landmark_id, cam_id */
void ba_adjuster_get_projection_residual(const ba_adjuster* a, uint32_t id, double* out11);
/* GetImuResidual(id): out19 = pose1_id, pose2_id, weight, number of measurements, residual(15: the
 * first PoseSize entries are used); returns the number of measurements (0 for a bad id) */
uint32_t ba_adjuster_get_imu_residual(const ba_adjuster* a, uint32_t id, double* out19);
void ba_adjuster_get_summary(const ba_adjuster* a, ba_summary* s);
/* SolutionSummary::cond_proj_error, cond_inertial_error (reference BundleAdjuster.cpp:680-704) */
void ba_adjuster_get_cond_errors(const ba_adjuster* a, double out2[2]);
void ba_adjuster_get_timers(const ba_adjuster* a, ba_hip_timers* t);
/* the engine behind the adjuster (valid after the first Solve) for the debug taps of ba_hip.h */
ba_hip_engine* ba_adjuster_engine(ba_adjuster* a);
/* rig()->cameras_[cam_id]->Pose(): with do_tvs camera 0 moves with every applied step */
void ba_adjuster_get_camera_pose(const ba_adjuster* a, uint32_t cam_id, double t_vs[7]);
/* rig()->cameras_[cam_id]->GetParams(): with calib_size 4 / 5 camera 0's move with every applied step */
void ba_adjuster_get_camera_params(const ba_adjuster* a, uint32_t cam_id, double params[4]);
/* the fifth parameter w of a FovCamera (0 for a LinearCamera) */
double ba_adjuster_get_camera_fov(const ba_adjuster* a, uint32_t cam_id);
/* SolutionSummary::calibration_marginals (6 x 6, row-major) of the last iteration; returns its
 * dimension (0 when the option was off or the adjuster has no calibration unknowns) */
uint32_t ba_adjuster_get_calibration_marginals(const ba_adjuster* a, double cov[36]);
/* GetLastStep().delta_k (zeros without do_tvs) */
void ba_adjuster_get_last_calib_step(const ba_adjuster* a, double delta_k[6]);
void ba_adjuster_set_allreduce(ba_adjuster* a, ba_hip_allreduce_fn fn, void* ctx, int rank, int nranks);
/* ba::BundleAdjuster::SetCommunicator / ClearCommunicator (id128 = NULL): the engine-owned RCCL communicator;
 * the next Solve() joins it (collective).  distributed_solve 1 = distributed reduced solve, 0 = replicated. */
void ba_adjuster_set_communicator(ba_adjuster* a, const void* id128, int rank, int nranks, int distributed_solve);
int ba_adjuster_solve_is_distributed(ba_adjuster* a);
/* ba::BundleAdjuster::SetCollectives: the collectives hook on top of the all-reduce hook (distributed reduced solve) */
void ba_adjuster_set_collectives(ba_adjuster* a, ba_hip_collective_fn fn, void* ctx);

#ifdef __cplusplus
}
#endif
#endif
