// Internal declarations of the gfx950 engine behind include/ba_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ba_hip.h"
#include "structure.h"
#include "dist_plan.h"

namespace bae {

// ---- small RAII-less device buffer (engine owns and frees explicitly) -------------
template <typename T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    if (count <= n && p) return hipSuccess;
    release();
    if (count == 0) { n = 0; return hipSuccess; }
    hipError_t err = hipMalloc((void**)&p, count * sizeof(T));
    if (err == hipSuccess) n = count;
    return err;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  size_t bytes() const { return n * sizeof(T); }
};

// Rigid transform in matrix form as stored on the device: R row-major (9) then t (3).
static const int kRt = 12;
static const int kCamRec = 37;  // camera record: params(4) | T_vs Rt(12) | T_sv Rt(12) | T_vs as t,q (7) | w | model
// Pose state row: t(3) q(4) v(3) b(6)
static const int kPoseState = 16;
// one factor row: 6 doubles (see DESIGN.md "factor rows")
static const int kRow = 6;

// Host-built structure (ba_hip_finalize): the lists of structure.h + the pose-pose scatter count
struct Structure : Lists {
  uint32_t n_pp_entries = 0;  // (pose, residual side) entries of the pose-pose scatter
};

struct Engine {
  int lm_dim = 1, pose_dim = 6, device = 0;
  // Calibration unknowns behind the pose unknowns of the reduced system (ba_hip_set_calibration):
  // 0, or 6 = the decoupled update of T_vs of camera 0 (the reference's DoTvs instantiations).
  int calib_dim = 0;
  // calib_tvs: the six unknowns are T_vs; otherwise (calib_dim == 4 | 5) the parameters of camera 0 — (fx, fy,
  // u0, v0) of a LinearCamera, (fx, fy, u0, v0, w) of a FovCamera (CalibSize; BundleAdjuster.cpp:46-69, parallel_algos.h:114-118)
  bool calib_tvs = false;
  std::vector<double> cam_params_prev;   // params_backup of SolveInternal (:1025-1028): restored by a rollback
  std::vector<double> cam_w_prev;        // ... its fifth entry for a FOV camera
  bool has_fov = false;                  // some camera of the rig is a FovCamera: the FOV kernel instantiations run
  DBuf<double> lm_zref;                  // [L][2] reference pixel of every landmark (LandmarkT::z_ref)
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // bulk trailing updates of the factorisation (look-ahead)
  const double* A_cleared = nullptr;  // the allocation of A whose whole square has been zeroed once
  std::vector<hipEvent_t> ev_panel, ev_bulk;
  hipEvent_t ev_imu_start = nullptr, ev_imu_done = nullptr;  // k_imu on stream2 (launch_imu_early)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;           // k_pose_blocks on stream2 (launch_gather_S)
  bool imu_early_pending = false;
  bool own_stream = false;
  std::string err;
  ba_hip_options opt;
  Problem prob;
  Structure st;
  bool finalized = false;
  int cur = 0;  // current state buffer (0/1)
  bool has_snapshot = false;
  bool factored = false;  // A currently holds the Cholesky factor, not S
  ba_hip_timers timers;
  ba_hip_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  ba_hip_collective_fn coll = nullptr;   // broadcast / reduce-scatter: distributed reduced solve
  void* coll_ctx = nullptr;
  int rank = 0, nranks = 1;
  // native RCCL communicator (ba_hip_comm_init, comm.hip); comm_force: run the sharded code paths
  // even with one rank (exercises the RCCL calls on a one-GPU box)
  void* comm = nullptr;
  bool comm_force = false;
  bool sharded() const { return allreduce && (nranks > 1 || comm_force); }

  // ---- device: static problem data
  DBuf<double> cam;                 // [C][4 + 12 (T_vs) + 12 (T_sv) + 7 (T_vs as t, q)]
  // With T_vs a parameter the reference reads it in two places: the rig (Jacobian chains) and the
  // per-pose caches of T_sw (PoseT::GetTsw, Types.h:61-70), and a rejected step restores the caches
  // but not the rig (BundleAdjuster.cpp:1060-1068).  `cam` is the rig; cam_eval feeds k_pose_prep
  // (the caches) and differs from it only between a rollback and the next accepted/applied step.
  DBuf<double> cam_eval;
  std::vector<double> tvs_eval, tvs_eval_prev;   // host copies, [C][7]
  const double* cam_eval_ptr() const { return calib_dim ? cam_eval.p : cam.p; }
  DBuf<double> pose_cam;            // [P][4] per-pose intrinsics (Options::use_per_pose_cam_params)
  const double* pose_cam_ptr() const { return prob.pose_cam_params.empty() ? nullptr : pose_cam.p; }
  DBuf<int32_t> pose_opt, lm_opt;
  DBuf<uint16_t> pose_mask;
  DBuf<uint32_t> lm_ref_pose, lm_ref_cam;
  DBuf<uint32_t> lm_ptr;            // [L+1] CSR over sorted observations
  DBuf<double> obs_z;               // [O][2]
  DBuf<uint32_t> obs_pose, obs_cam, obs_lm, obs_rid;
  DBuf<double> obs_w0;
  DBuf<uint8_t> obs_cond;                // 1 = conditioning residual (ba_hip_set_conditioning_residuals)
  // static lists of structure.h
  DBuf<uint2> wave_rng;                  // [n_chunks] observation ranges of the linearisation waves
  DBuf<uint32_t> tile_ptr;               // [tiles_lower+1]
  DBuf<uint2> tile_ref;                  // (first term, count << 14 | offsets) per block overlapping a tile
  DBuf<uint32_t> tile_order;             // launch order of the tile assembly (build_tile_order)
  DBuf<uint4> tile_desc;                 // per launch entry: (tile, first ref, end ref, row << 16 | column) (k_tile_desc)
  uint32_t n_tile_order = 0;
  uint64_t tile_order_version = ~0ull;
  uint64_t tile_order_plan = ~0ull;      // distributed solve: the launch list holds the owned + sent tiles of plan version ..
  int dbg_assemble_variant = 5;          // k_assemble_tiles<VAR> (ba_hip_debug_set key 1)
  int dbg_host_structure = 0;            // 1: build the static lists on the host (structure.h) (key 5)
  int dbg_linearize_variant = 0;         // 0 LDS-staged factor rows, 1 direct stores (key 4)
  int dbg_tile_order = 0;                // 0 row-major tiles, 1 XCD-aware columns (key 2)
  int dbg_imu_wave = -1;                 // k_imu: -1 by count, 0 lanes, 1 wavefront per residual, 2 fused single pass, 4 wavefront per residual and per sample (key 6)
  int dbg_all_tiles = 0;                 // 1: assemble / zero every lower tile, not only the factor's pattern (key 3)
  DBuf<uint2> pair_ent;                  // (rowA, rowB) rank-1 terms of the off-diagonal blocks
  DBuf<uint32_t> pose_ptr, pose_mid;     // [Pact+1], [Pact]
  DBuf<uint32_t> pose_ent;               // 3 x n_pose_entries: (rowA, rowB, scalar index)
  DBuf<uint8_t> nzL;                     // tile pattern of the factor L (nt x nt bytes, lower), see k_chol.hip
  bool nzL_valid = false;
  std::vector<uint8_t> nzL_host;         // host copy (flop accounting of the profiled launches)
  uint64_t nzL_version = 0;
  // distributed solve: per panel the row tiles with a structurally nonzero tile (message rows)
  std::vector<uint8_t> nzS_host;         // tile pattern of S itself (union over the shards), before fill
  DBuf<uint32_t> dist_srows;             // reduce-scatter: row tiles of every panel with a nonzero S tile
  std::vector<uint32_t> dist_srows_off;
  uint64_t dist_srows_version = ~0ull;
  DBuf<uint32_t> dist_rows;
  std::vector<uint32_t> dist_rows_off;
  uint64_t dist_rows_version = ~0ull;
  DBuf<double> dist_msg;                 // square broadcast message (distributed solve)
  // plan of the distributed solve (dist_plan.h): ownership map, per-panel row lists and messages
  DistPlan dist_plan;
  uint64_t dist_plan_version = ~0ull;
  std::string dist_layout_name;
  DBuf<uint32_t> dist_tiles;             // DistPlan::tiles on the device
  // sparse exchange of S (dist_scatter_S_sparse): rectangles per peer (prefix offsets [N + 1]) and their staging offsets
  uint64_t dist_sp_version = ~0ull;
  std::vector<uint32_t> dist_sp_send, dist_sp_recv;
  std::vector<size_t> dist_sp_send_off, dist_sp_recv_off;
  DBuf<uint4> dist_sp_srect, dist_sp_rrect;
  DBuf<uint2> dist_pairs;                // block pairs this rank owns, by block column (k_update128<false, true>)
  std::vector<uint32_t> dist_pair_first; // per panel J: first pair with block column >= J + 2
  uint32_t dist_npairs = 0;
  DBuf<uint32_t> dist_sq_list;           // per panel: the tiles of its square, then the rhs row (row lists of the chain kernels)
  DBuf<double> dist_usend, dist_urecv;   // urgent point-to-point staging (chain stream)
  DBuf<double> dist_ssend, dist_srecv;   // side point-to-point staging (side stream)
  DBuf<double> dist_back;                // backward substitution: partial sums of a panel + their reduction
  hipStream_t stream3 = nullptr;         // distributed solve: bulk trailing updates
  hipStream_t stream4 = nullptr;         // distributed solve: side communicator
  void* comm2 = nullptr;                 // side communicator (ncclCommSplit of comm)
  std::vector<hipEvent_t> ev_dist;       // 5 events per panel: urgent, packed, side, next, bulk
  ba_hip_comm_stats cstats = {};
  DBuf<double> packed;                   // packed lower triangle + rhs row (all-reduce staging)

  // pose-pose residuals (unary | binary | imu slots)
  DBuf<uint8_t> pose_active;
  DBuf<uint32_t> un_pose; DBuf<double> un_t, un_cov_inv, un_scale; DBuf<uint8_t> un_rot;
  DBuf<uint32_t> bin_p1, bin_p2; DBuf<double> bin_t, bin_cov_inv, bin_cov_inv_sqrt, bin_w;
  DBuf<uint8_t> bin_rot;
  DBuf<uint32_t> imu_p1, imu_p2, imu_ptr; DBuf<double> imu_meas, imu_consts, imu_cov_inv;
  // Options::calculate_inertial_covariance_once: per residual 10x10 integration covariance + 10x6
  // bias Jacobian of its first linearisation, and a "done" flag
  bool imu_cov_once = false;
  DBuf<double> imu_frozen;
  DBuf<double> imu_steps;                // [samples][160] step Jacobians of the pre-integration (k_imu_steps)
  DBuf<uint8_t> imu_cov_done;
  uint32_t imu_cov_count = 0;
  DBuf<double> pp_h, pp_g, pp_dz, pp_info, pp_err;
  DBuf<uint32_t> pp_ptr, pp_res_p1, pp_res_p2;
  DBuf<uint4> pp_ent;

  // ---- device: state (double buffered)
  DBuf<double> pose_state[2];            // [P][16]
  DBuf<double> lm_x[2];                  // [L][4]  x_s (lm_dim 1) or x_w (lm_dim 3)
  DBuf<uint8_t> lm_reliable[2];
  DBuf<double> lm_xw;                    // [L][4]  world points (upload / download, lm_dim 1)
  DBuf<double> tsw, tws, twp;            // [P*C][12], [P*C][12], [P][12] for state `cur`
  DBuf<uint32_t> lm_outliers;

  // ---- device: per-iteration
  DBuf<double> obs_e, obs_w;             // error for the median, robust weight
  // err_cache: |r|^2 * original weight per observation at the state held in buffer 0 / 1, left behind by the last
  // EvaluateResiduals at that state (k_residuals mode 1); ba_hip_linearize takes its Huber median from it instead of
  // running the error pass again.  Invalidated by everything that changes a state buffer or the cameras; off with
  // calibration unknowns (the camera moves with the step) and on request (BA_HIP_NO_ERR_CACHE).
  DBuf<double> obs_e_state[2];
  bool obs_e_valid[2] = {false, false};
  bool err_cache_on() const { return calib_dim == 0 && !err_cache_off; }
  bool err_cache_off = false;
  void err_cache_clear() { obs_e_valid[0] = obs_e_valid[1] = false; }
  DBuf<double> obs_jl;                   // [O][2*lm] sqrt(w) * dz_dlm (dogleg J_l * rhs_l)
  DBuf<double> diag_blocks;              // [Pact][36] diagonal blocks of S (k_pose_blocks -> k_write_diag)
  DBuf<double> crow;                     // [n_scalars][6] calibration rows, indexed like `scal`: sqrt(w) dz_dtvs
                                         // (u, v row) of observation a at 2a, 2a+1; E_l = sum w Jl^T Jk of
                                         // landmark l at 2O + l; the last row stays zero
  DBuf<double> border_blocks;            // [Pact][36] S_pk blocks (k_pose_border -> k_write_border)
  DBuf<double> calib_partials;           // block partials of k_calib_reduce
  DBuf<double> frow;                     // [n_rows][6] observation-major factor rows (structure.h)
  DBuf<double> scal;                     // scalars: [2*O] sqrt(w) r, then [L*lm] b_l, then one 0
  DBuf<double> lm_vinv, lm_bl;           // [L][lm*lm], [L][lm]
  DBuf<double> A;                        // [(n+1)][ld] lower storage + rhs row
  DBuf<double> A_keep;                   // copy of A before factorisation (debug option)
  DBuf<double> rhs_p, rhs_sc;            // [n] unreduced / reduced (copy of A's last row)
  DBuf<double> gn_p, gn_l, step_p, step_l;
  DBuf<double> invdiag;                  // inverse diagonal tiles of the Cholesky factor
  DBuf<double> partials;                 // reduction scratch
  DBuf<double> scalars_out;              // small result block (device) + host mirror
  // Deferred scalars: between defer_begin() and defer_flush() the reductions of sum_partials / sum_small
  // leave their results in scalars_out[16 ..) and only note where the host wants them; the flush is ONE
  // device-to-host copy and ONE synchronisation for the whole phase call (a small window pays ~30 us per
  // round trip: four of them made ba_hip_dogleg_terms 140 us).  Off for sharded engines, whose sums go
  // through the all-reduce hook one by one.
  std::vector<hipEvent_t> timer_events;  // pool of the phase timers (engine.hip: EventTimer)
  bool defer_active = false;
  int defer_n = 0;
  double* defer_host[40];
  double dog_h[8];                       // landing place of the dogleg sums (ba_hip_dogleg_terms)
  bool dog_jrhs_valid = false;           // dog_h[6], dog_h[7] (|J rhs|^2: projection / pose-pose part) belong to the current linearisation
  double eval_h[4];                      // ... of the evaluation sums (ba_hip_eval_residuals)
  DBuf<unsigned long long> hist;         // selection histograms
  DBuf<int32_t> flags;                   // factorisation status block (k_chol.hip: setup_status_block)
  bool square_attr_set = false;          // k_square / k_rowpanel: dynamic LDS limit raised on this engine's device
  DBuf<double> pivot_floor;              // tol * |S_jj| per row (ba_hip_options::pivot_rel_tolerance)

  // optional per-kernel timing (ba_hip_set_profiling)
  bool profiling = false;
  ba_hip_kernel_stats kstats = {};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_syrk, ev_gather, ev_landmarks, ev_imu, ev_pose;
  void prof_begin(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, hipStream_t s = nullptr) {
    if (!profiling) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s ? s : stream);
    v.push_back({a, b});
  }
  void prof_end(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, hipStream_t s = nullptr) {
    if (!profiling) return;
    (void)hipEventRecord(v.back().second, s ? s : stream);
  }
  void prof_collect();

  int fail(hipError_t e, const char* what);
  int fail_msg(const char* what);
};

// the installed all-reduce (hook or native), with the byte counter of ba_hip_get_comm_stats
inline int shard_allreduce(Engine* e, void* dev_ptr, size_t count, int dtype) {
  e->cstats.allreduce_bytes += 8.0 * (double)count;
  return e->allreduce(e->allreduce_ctx, dev_ptr, count, dtype);
}

#define BAE_HIP(call)                                                    \
  do {                                                                   \
    hipError_t _e = (call);                                              \
    if (_e != hipSuccess) return e->fail(_e, #call);                     \
  } while (0)

// ---- kernel launchers (defined in k_*.hip); all enqueue on e->stream -------------
int launch_pose_prep(Engine* e);                       // T_sw, T_ws, T_wp of the current state
int upload_cameras(Engine* e, bool eval_only);         // camera table(s) from prob.cam_* / tvs_eval
int launch_reset_rays(Engine* e);                      // x_s rays of the current state from the reference pixels
int launch_calib_border(Engine* e);                    // S_pk, S_kk, rhs_k into A / rhs (after k_write_diag)
int launch_calib_dogleg(Engine* e, int gn_available, ba_hip_dogleg_scalars* out);
int launch_begin_solve(Engine* e);                     // x_s from x_w
int launch_end_solve(Engine* e);                       // x_w from x_s
int launch_residual_vectors(Engine* e, double* d_r2);  // z - pi per observation at the current state
int launch_residuals(Engine* e, int mode);             // mode 0: errors for the median; 1: EvaluateResiduals
int launch_landmarks(Engine* e, double c_huber, int use_robust);  // Jacobians, V, W, rows
int launch_gather_S(Engine* e);
int launch_pack_lower(Engine* e, int unpack);
size_t packed_lower_count(uint32_t n_pad);                        // pair gather -> A, rhs row, masks
int launch_backsub(Engine* e);                         // delta_l
int launch_compose_step(Engine* e, double coef_rhs, double coef_gn, double* norms2_host);
int launch_apply_step(Engine* e);                      // state[cur] -> state[1-cur]
int launch_dogleg(Engine* e, int gn_available, double* h7, bool skip_jrhs);
int select_kth(Engine* e, const double* d_values, uint32_t n_local, uint64_t k, double* out);
int sum_partials(Engine* e, uint32_t nparts, uint32_t ncomp, double* host_out, bool cross_shard = true);
void defer_begin(Engine* e);   // k_reduce.hip
int defer_flush(Engine* e);
int launch_imu_early(Engine* e, double c_huber_proj);
int launch_posepose_build(Engine* e, double c_huber_proj, double* h3);
int launch_posepose_eval(Engine* e, double* h3);
int launch_posepose_jrhs(Engine* e, double* out);
int cholesky_solve(Engine* e, double* dA, uint32_t n, uint32_t ld, double* dx, int* status,
                   const uint8_t* nz_tiles);
int factor_tile_pattern(Engine* e);
uint32_t choose_kout(uint32_t nblk);
bool dist_solve_enabled(const Engine* e);
int dist_reduce_scatter_S(Engine* e);
int launch_imu_residual_vectors(Engine* e, double* d_r15);
int build_tile_order(Engine* e);
int build_tile_desc(Engine* e);   // k_reduce.hip
int dist_assembly_tiles(Engine* e, std::vector<uint8_t>* need);  // k_chol.hip
// the static lists built on the device (structure_dev.hip); same contents as structure.h's host builder
int build_lists_device(Engine* e, const std::function<void(const char*)>& stage);
// broadcast of `count` doubles from `root`, ordered into `s`: native RCCL enqueues without a host
// round trip; with a caller-supplied hook the stream is drained first (hook contract)
int dist_broadcast(Engine* e, double* buf, size_t count, int root, hipStream_t s);
// One group of point-to-point transfers, ordered into `s`: `side` selects the side communicator.  Native
// RCCL: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd; hook: the stream is drained, sends first (they
// do not block), then the receives.
struct DistXfer { double* buf; size_t count; int peer; bool send; };
int dist_exchange(Engine* e, const std::vector<DistXfer>& x, bool side, hipStream_t s);
// in-place SUM over the ranks of `count` doubles, ordered into `s` (chain communicator)
int dist_allreduce_stream(Engine* e, double* buf, size_t count, hipStream_t s);
void comm_release(Engine* e);
int check_solve_residual(Engine* e, const double* dS, const double* dx, const double* db, double* out2);
int cholesky_solve_dist(Engine* e, double* dA, uint32_t ld, double* dx, int* status, const uint8_t* nz);
int trailing_marginals(Engine* e, const double* dA, uint32_t ld, uint32_t first, uint32_t K, double* cov);

}  // namespace bae
