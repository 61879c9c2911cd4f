"""Python view of the C++ host class ba::BundleAdjuster<> (include/ba/BundleAdjuster.h)
through its flat C wrapper (include/ba_capi.h, ba_amd/lib/libba_capi.so).

Method names follow the reference API (/root/reference/include/ba/BundleAdjuster.h:
177-631) so parity tests can feed the same calls to this class and to the oracle.
Everything below runs on the MI355X engine; importing works without a GPU, Solve()
does not (the result is SolverError and an error line — there is no CPU fallback).
"""
import ctypes as C
import os

import numpy as np

from . import hipapi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libba_capi.so")

dp = C.POINTER(C.c_double)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)

RESULT_NAMES = ["Success", "ErrorIncreased", "ErrorChangeBelowThreshold",
                "ParamChangeBelowThreshold", "FactorizationError", "SolverError"]


class BaOptions(C.Structure):
    _fields_ = [("trust_region_size", C.c_double),
                ("gyro_sigma", C.c_double), ("accel_sigma", C.c_double),
                ("gyro_bias_sigma", C.c_double), ("accel_bias_sigma", C.c_double),
                ("projection_outlier_threshold", C.c_double),
                ("error_change_threshold", C.c_double), ("param_change_threshold", C.c_double),
                ("dogleg_max_inner_iterations", C.c_uint32),
                ("apply_results", C.c_int32), ("use_dogleg", C.c_int32),
                ("use_triangular_matrices", C.c_int32), ("use_sparse_solver", C.c_int32),
                ("regularize_biases_in_batch", C.c_int32),
                ("enable_auto_regularization", C.c_int32),
                ("use_robust_norm_for_proj_residuals", C.c_int32),
                ("use_robust_norm_for_inertial_residuals", C.c_int32),
                ("write_reduced_camera_matrix", C.c_int32),
                ("device", C.c_int32), ("factorization_pivot_tolerance", C.c_double),
                ("calculate_calibration_marginals", C.c_int32), ("reserved", C.c_int32)]


class BaSummary(C.Structure):
    _fields_ = [("num_proj_residuals", C.c_uint32), ("num_inertial_residuals", C.c_uint32),
                ("num_cond_proj_residuals", C.c_uint32), ("num_cond_inertial_residuals", C.c_uint32),
                ("proj_error", C.c_double), ("inertial_error", C.c_double),
                ("unary_error", C.c_double), ("binary_error", C.c_double),
                ("delta_norm", C.c_double), ("pre_solve_norm", C.c_double),
                ("post_solve_norm", C.c_double), ("result", C.c_int32),
                ("iterations_run", C.c_uint32), ("trust_region_size", C.c_double)]


SYMBOLS = [
    "ba_default_options", "ba_adjuster_create", "ba_adjuster_destroy", "ba_adjuster_init",
    "ba_adjuster_set_gravity", "ba_adjuster_add_camera", "ba_adjuster_add_pose",
    "ba_adjuster_add_landmark", "ba_adjuster_add_projection_residual",
    "ba_adjuster_add_unary_constraint", "ba_adjuster_add_binary_constraint",
    "ba_adjuster_add_imu_residual", "ba_adjuster_regularize_pose", "ba_adjuster_set_root_pose_id",
    "ba_adjuster_set_pose_cam_params", "ba_adjuster_set_calculate_inertial_covariance_once", "ba_adjuster_set_imu_noise",
    "ba_adjuster_add_poses", "ba_adjuster_add_landmarks", "ba_adjuster_add_projection_residuals",
    "ba_adjuster_solve", "ba_adjuster_num_poses", "ba_adjuster_num_landmarks",
    "ba_adjuster_num_proj_residuals", "ba_adjuster_get_poses", "ba_adjuster_get_landmarks",
    "ba_adjuster_is_landmark_reliable", "ba_adjuster_landmark_outlier_ratio",
    "ba_adjuster_get_projection_residual", "ba_adjuster_get_imu_residual",
    "ba_adjuster_get_summary", "ba_adjuster_get_cond_errors", "ba_adjuster_get_timers", "ba_adjuster_engine",
    "ba_adjuster_set_allreduce", "ba_adjuster_set_communicator", "ba_adjuster_solve_is_distributed", "ba_adjuster_set_collectives", "ba_adjuster_create_calib", "ba_adjuster_get_camera_pose",
    "ba_adjuster_get_last_calib_step", "ba_adjuster_get_calibration_marginals", "ba_adjuster_get_camera_params",
    "ba_adjuster_add_camera_fov", "ba_adjuster_get_camera_fov",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        hipapi.lib()  # libba_capi.so links libba_hip.so; fail with the clear message first
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("ba_amd: %s is missing — run __graft_entry__.build()" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        L = _lib
        L.ba_adjuster_create.restype = C.c_void_p
        L.ba_adjuster_create_calib.restype = C.c_void_p
        L.ba_adjuster_engine.restype = C.c_void_p
        L.ba_adjuster_landmark_outlier_ratio.restype = C.c_double
        L.ba_adjuster_get_camera_fov.restype = C.c_double
        for n in ("ba_adjuster_add_camera", "ba_adjuster_add_pose", "ba_adjuster_add_landmark",
                  "ba_adjuster_add_projection_residual", "ba_adjuster_add_unary_constraint",
                  "ba_adjuster_add_binary_constraint", "ba_adjuster_add_imu_residual",
                  "ba_adjuster_num_poses", "ba_adjuster_num_landmarks",
                  "ba_adjuster_num_proj_residuals"):
            getattr(L, n).restype = C.c_uint32
    return _lib


def default_options():
    o = BaOptions()
    lib().ba_default_options(C.byref(o))
    return o


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class _EngineView(hipapi.Engine):
    """Debug taps (include/ba_hip.h) on the engine owned by a C++ adjuster."""

    def __init__(self, handle, lm_dim, pose_dim):  # the base constructor would create an engine
        self.L = hipapi.lib()
        self.lm_dim, self.pose_dim = lm_dim, pose_dim
        self.h = C.c_void_p(handle)
        self._cb = None

    def close(self):
        self.h = C.c_void_p()  # not owned


class BundleAdjuster:
    """ba::BundleAdjuster<double, lm_dim, pose_dim, 0, do_tvs> on the MI355X engine."""

    def __init__(self, lm_dim=1, pose_dim=6, do_tvs=False, calib_size=0):
        self.L = lib()
        self.lm_dim, self.pose_dim, self.do_tvs, self.calib_size = lm_dim, pose_dim, bool(do_tvs), int(calib_size)
        self.h = C.c_void_p(self.L.ba_adjuster_create_calib(lm_dim, pose_dim, int(calib_size), int(do_tvs)))
        if not self.h:
            raise ValueError("unsupported (lm_dim, pose_dim, calib_size, do_tvs)")
        self._cb = None

    def __del__(self):
        try:
            if self.h:
                self.L.ba_adjuster_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- reference API -----------------------------------------------------------
    def Init(self, options=None):
        self.options = options if options is not None else default_options()
        self.L.ba_adjuster_init(self.h, C.byref(self.options))

    def SetGravity(self, g):
        g = _d(g)
        self.L.ba_adjuster_set_gravity(self.h, _p(g, dp))

    def AddCamera(self, params, t_vs=(0, 0, 0, 0, 0, 0, 1)):
        """params (fx, fy, u0, v0): calibu::LinearCamera; (fx, fy, u0, v0, w): calibu::FovCamera."""
        p, t = _d(params), _d(t_vs)
        self._cam_fov = getattr(self, "_cam_fov", []) + [p.size == 5]
        if p.size == 5:
            return self.L.ba_adjuster_add_camera_fov(self.h, _p(p, dp), _p(t, dp))
        return self.L.ba_adjuster_add_camera(self.h, _p(p, dp), _p(t, dp))

    def AddPose(self, t_wp, is_active=True, time=-1.0, v_w=(0, 0, 0), b=(0,) * 6):
        t, v, bb = _d(t_wp), _d(v_w), _d(b)
        return self.L.ba_adjuster_add_pose(self.h, _p(t, dp), _p(v, dp), _p(bb, dp), int(is_active),
                                           C.c_double(time))

    def SetPoseCamParams(self, pose_cam_params):
        """Per-pose pinhole intrinsics (P x 4) + Options::use_per_pose_cam_params (reference
        BundleAdjuster.h:96, 292-323); None switches back to the rig camera."""
        if pose_cam_params is None:
            rc = self.L.ba_adjuster_set_pose_cam_params(self.h, 0, None)
        else:
            a = _d(pose_cam_params).reshape(-1, 4)
            rc = self.L.ba_adjuster_set_pose_cam_params(self.h, a.shape[0], _p(a, dp))
        if rc != 0:
            raise ValueError("SetPoseCamParams: one [fx, fy, u0, v0] per pose expected")

    def SetImuNoise(self, r6, rb6):
        """SetImuCalibration with other noise diagonals r / r_b (reference BundleAdjuster.h:566-567)."""
        a, b = _d(r6), _d(rb6)
        self.L.ba_adjuster_set_imu_noise(self.h, _p(a, dp), _p(b, dp))

    def SetCalculateInertialCovarianceOnce(self, on=True):
        """Options::calculate_inertial_covariance_once (reference BundleAdjuster.h:106)."""
        self.L.ba_adjuster_set_calculate_inertial_covariance_once(self.h, int(on))

    def SetUsePerPoseCamParams(self, on=True):
        if not on:
            self.SetPoseCamParams(None)

    def AddLandmark(self, x_w, ref_pose_id, ref_cam_id=0, is_active=True):
        x = _d(x_w)
        return self.L.ba_adjuster_add_landmark(self.h, _p(x, dp), int(ref_pose_id), int(ref_cam_id),
                                               int(is_active))

    def AddProjectionResidual(self, z, meas_pose_id, landmark_id, cam_id=0, weight=1.0):
        zz = _d(z)
        return self.L.ba_adjuster_add_projection_residual(self.h, _p(zz, dp), int(meas_pose_id),
                                                          int(landmark_id), int(cam_id),
                                                          C.c_double(weight))

    def AddUnaryConstraint(self, pose_id, t_wv, covariance, use_rotation=True):
        t, c = _d(t_wv), _d(covariance).reshape(36)
        return self.L.ba_adjuster_add_unary_constraint(self.h, int(pose_id), _p(t, dp), _p(c, dp),
                                                       int(use_rotation))

    def AddBinaryConstraint(self, p1, p2, t_12, covariance=None, weight=1.0, use_rotation=True):
        t = _d(t_12)
        c = _d(np.eye(6) if covariance is None else covariance).reshape(36)
        return self.L.ba_adjuster_add_binary_constraint(self.h, int(p1), int(p2), _p(t, dp),
                                                        _p(c, dp), C.c_double(weight),
                                                        int(use_rotation))

    def AddImuResidual(self, p1, p2, meas, weight=1.0):
        m = _d(meas).reshape(-1, 7)
        return self.L.ba_adjuster_add_imu_residual(self.h, int(p1), int(p2), _p(m, dp), m.shape[0],
                                                   C.c_double(weight))

    def RegularizePose(self, pose_id, translation, gravity, bias, rotation):
        self.L.ba_adjuster_regularize_pose(self.h, int(pose_id), int(translation), int(gravity),
                                           int(bias), int(rotation))

    def SetRootPoseId(self, i):
        self.L.ba_adjuster_set_root_pose_id(self.h, int(i))

    def Solve(self, max_iter, gn_damping=1.0, error_increase_allowed=False):
        self.L.ba_adjuster_solve(self.h, int(max_iter), C.c_double(gn_damping),
                                 int(error_increase_allowed))

    def GetNumPoses(self):
        return self.L.ba_adjuster_num_poses(self.h)

    def GetNumLandmarks(self):
        return self.L.ba_adjuster_num_landmarks(self.h)

    def GetNumProjResiduals(self):
        return self.L.ba_adjuster_num_proj_residuals(self.h)

    def IsLandmarkReliable(self, i):
        return bool(self.L.ba_adjuster_is_landmark_reliable(self.h, int(i)))

    def LandmarkOutlierRatio(self, i):
        return self.L.ba_adjuster_landmark_outlier_ratio(self.h, int(i))

    def GetProjectionResidual(self, i):
        """dict view of ba::ProjectionResidualT (reference BundleAdjuster.h:568-571)."""
        o = np.empty(11)
        self.L.ba_adjuster_get_projection_residual(self.h, int(i), _p(o, dp))
        return {"z": o[0:2].copy(), "residual": o[2:4].copy(), "weight": o[4], "orig_weight": o[5],
                "mahalanobis_distance": o[6], "x_meas_id": int(o[7]), "x_ref_id": int(o[8]),
                "landmark_id": int(o[9]), "cam_id": int(o[10])}

    def cond_errors(self):
        """(cond_proj_error, cond_inertial_error) of the SolutionSummary (reference :680-704)."""
        o = np.empty(2)
        self.L.ba_adjuster_get_cond_errors(self.h, _p(o, dp))
        return float(o[0]), float(o[1])

    def GetImuResidual(self, i):
        """dict view of ba::ImuResidualT (reference BundleAdjuster.h:563-565)."""
        o = np.empty(19)
        self.L.ba_adjuster_get_imu_residual.restype = C.c_uint32
        n = self.L.ba_adjuster_get_imu_residual(self.h, int(i), _p(o, dp))
        return {"pose1_id": int(o[0]), "pose2_id": int(o[1]), "weight": o[2], "num_measurements": int(n),
                "residual": o[4:19].copy()}

    # -- bulk adders -----------------------------------------------------------------
    def add_poses(self, t_wp, v_w=None, b=None, is_active=None, time=None):
        t = _d(t_wp).reshape(-1, 7)
        v = _d(v_w) if v_w is not None else None
        bb = _d(b) if b is not None else None
        a = np.ascontiguousarray(is_active, dtype=np.uint8) if is_active is not None else None
        tm = _d(time) if time is not None else None
        self.L.ba_adjuster_add_poses(self.h, t.shape[0], _p(t, dp), _p(v, dp), _p(bb, dp),
                                     _p(a, u8p), _p(tm, dp))

    def add_landmarks(self, x_w, ref_pose_id, ref_cam_id=None, is_active=None):
        x = _d(x_w).reshape(-1, 4)
        rp = np.ascontiguousarray(ref_pose_id, dtype=np.uint32)
        rc = np.ascontiguousarray(ref_cam_id, dtype=np.uint32) if ref_cam_id is not None else None
        a = np.ascontiguousarray(is_active, dtype=np.uint8) if is_active is not None else None
        self.L.ba_adjuster_add_landmarks(self.h, x.shape[0], _p(x, dp), _p(rp, u32p), _p(rc, u32p),
                                         _p(a, u8p))

    def add_projection_residuals(self, z, meas_pose_id, landmark_id, cam_id=None, weight=None):
        zz = _d(z).reshape(-1, 2)
        mp = np.ascontiguousarray(meas_pose_id, dtype=np.uint32)
        li = np.ascontiguousarray(landmark_id, dtype=np.uint32)
        ci = np.ascontiguousarray(cam_id, dtype=np.uint32) if cam_id is not None else None
        w = _d(weight) if weight is not None else None
        ids = np.empty(zz.shape[0], dtype=np.uint32)
        self.L.ba_adjuster_add_projection_residuals(self.h, zz.shape[0], _p(zz, dp), _p(mp, u32p),
                                                    _p(li, u32p), _p(ci, u32p), _p(w, dp),
                                                    _p(ids, u32p))
        return ids

    # -- results -------------------------------------------------------------------------
    def poses(self):
        n = self.GetNumPoses()
        t, v, b = np.empty((n, 7)), np.empty((n, 3)), np.empty((n, 6))
        self.L.ba_adjuster_get_poses(self.h, _p(t, dp), _p(v, dp), _p(b, dp))
        return t, v, b

    def landmarks(self):
        x = np.empty((self.GetNumLandmarks(), 4))
        self.L.ba_adjuster_get_landmarks(self.h, _p(x, dp))
        return x

    def summary(self):
        s = BaSummary()
        self.L.ba_adjuster_get_summary(self.h, C.byref(s))
        return s

    def timers(self):
        t = hipapi.Timers()
        self.L.ba_adjuster_get_timers(self.h, C.byref(t))
        return {n: getattr(t, n) for n, _ in hipapi.Timers._fields_}

    # -- engine taps (valid after a Solve) -------------------------------------------------
    def engine(self):
        h = self.L.ba_adjuster_engine(self.h)
        if not h:
            raise hipapi.HipError("the adjuster has no engine yet (Solve() not run or no GPU)")
        return _EngineView(h, self.lm_dim, self.pose_dim)

    def S(self):
        return self.engine().get_S()

    def rhs(self):
        return self.engine().get_rhs()[0]

    def rhs_p(self):
        return self.engine().get_rhs()[1]

    def rhs_l(self):
        return self.engine().get_rhs()[2]

    def delta_p(self):
        e = self.engine()
        return e.get_step()[0][:e.num_pose_params()]

    def delta_k(self):
        e = self.engine()
        return e.get_step()[0][e.num_pose_params():]

    def rhs_k(self):
        e = self.engine()
        return e.get_rhs()[1][e.num_pose_params():]

    def num_pose_params(self):
        return self.engine().num_pose_params()

    def proj_tvs_jacobians(self):
        return self.engine().get_calib_jacobians(self.GetNumProjResiduals())

    def calibration_marginals(self):
        """SolutionSummary::calibration_marginals (empty without the option / without do_tvs)."""
        c = np.zeros(36)
        self.L.ba_adjuster_get_calibration_marginals.restype = C.c_uint32
        k = self.L.ba_adjuster_get_calibration_marginals(self.h, c.ctypes.data_as(C.POINTER(C.c_double)))
        return c[:k * k].reshape(k, k)

    def camera_params(self, cam_id=0):
        """rig()->cameras_[cam_id]->GetParams()"""
        p = np.empty(4)
        self.L.ba_adjuster_get_camera_params(self.h, int(cam_id), p.ctypes.data_as(C.POINTER(C.c_double)))
        if getattr(self, "_cam_fov", [])[cam_id:cam_id + 1] == [True]:
            p = np.append(p, self.L.ba_adjuster_get_camera_fov(self.h, int(cam_id)))
        return p

    def proj_calib_jacobians(self):
        """sqrt(w) dz_dk per residual id in the j_kpr_ layout, 2 x kCalibDim."""
        e = self.engine()
        return e.get_calib_jacobians(self.GetNumProjResiduals())[:, :, :e.num_calib_params()]

    def camera_pose(self, cam_id=0):
        """rig()->cameras_[cam_id]->Pose()"""
        t = np.empty(7)
        self.L.ba_adjuster_get_camera_pose(self.h, int(cam_id), t.ctypes.data_as(C.POINTER(C.c_double)))
        return t

    def delta_l(self):
        return self.engine().get_step()[1]

    def proj_weights(self):
        return self.engine().get_proj_weights(self.GetNumProjResiduals())

    def set_allreduce(self, fn, rank, nranks):
        """fn(dev_ptr:int, count:int, dtype:int) -> int; kept alive by this object."""
        if fn is None:
            self._cb = None
            self.L.ba_adjuster_set_allreduce(self.h, None, None, 0, 1)
            return
        self._cb = hipapi.ALLREDUCE_FN(lambda ctx, ptr, count, dtype: int(fn(ptr, count, dtype)))
        self.L.ba_adjuster_set_allreduce(self.h, self._cb, None, int(rank), int(nranks))

    def set_communicator(self, unique_id, rank, nranks, distributed_solve=True):
        """ba::BundleAdjuster::SetCommunicator: the engine's own RCCL communicator (unique_id = 128 bytes from
        hipapi.Engine.comm_unique_id() on rank 0); None clears it.  The next Solve() joins (collective)."""
        if unique_id is None:
            self.L.ba_adjuster_set_communicator(self.h, None, 0, 1, 0)
            return
        self.L.ba_adjuster_set_communicator(self.h, C.c_char_p(unique_id), int(rank), int(nranks), 1 if distributed_solve else 0)

    def solve_is_distributed(self):
        return bool(self.L.ba_adjuster_solve_is_distributed(self.h))

    def set_collectives(self, fn):
        """ba::BundleAdjuster::SetCollectives: fn(op, dev_ptr, count, root) -> int, on top of set_allreduce."""
        if fn is None:
            self._cb2 = None
            self.L.ba_adjuster_set_collectives(self.h, None, None)
            return
        self._cb2 = hipapi.COLLECTIVE_FN(lambda ctx, op, ptr, count, root: int(fn(op, ptr, count, root)))
        self.L.ba_adjuster_set_collectives(self.h, self._cb2, None)
