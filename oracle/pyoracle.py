"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libba_oracle.so (the CPU restatement of arpg/ba's
BundleAdjuster<> Gauss-Newton path, oracle/ba_oracle.cpp).  Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
(ba_amd/, include/) never imports it.

The class mirrors the reference API (/root/reference/include/ba/BundleAdjuster.h:
177-631): Init / AddCamera / AddPose / AddLandmark / AddProjectionResidual /
AddUnaryConstraint / AddBinaryConstraint / AddImuResidual / Solve / Get*.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libba_oracle.so")

c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)


class OrcOptions(C.Structure):
    _fields_ = [
        ("trust_region_size", C.c_double),
        ("gyro_sigma", C.c_double), ("accel_sigma", C.c_double),
        ("gyro_bias_sigma", C.c_double), ("accel_bias_sigma", C.c_double),
        ("projection_outlier_threshold", C.c_double),
        ("error_change_threshold", C.c_double), ("param_change_threshold", C.c_double),
        ("dogleg_max_inner_iterations", C.c_uint32),
        ("apply_results", C.c_int), ("use_dogleg", C.c_int),
        ("use_triangular_matrices", C.c_int), ("use_sparse_solver", C.c_int),
        ("regularize_biases_in_batch", C.c_int), ("enable_auto_regularization", C.c_int),
        ("use_robust_norm_for_proj_residuals", C.c_int),
        ("use_robust_norm_for_inertial_residuals", C.c_int),
    ]


class OrcSummary(C.Structure):
    _fields_ = [
        ("num_proj_residuals", C.c_uint32), ("num_inertial_residuals", C.c_uint32),
        ("num_cond_proj_residuals", C.c_uint32), ("num_cond_inertial_residuals", C.c_uint32),
        ("cond_proj_error", C.c_double), ("cond_inertial_error", C.c_double),
        ("proj_error", C.c_double), ("inertial_error", C.c_double),
        ("delta_norm", C.c_double), ("pre_solve_norm", C.c_double),
        ("post_solve_norm", C.c_double), ("result", C.c_int),
        ("unary_error", C.c_double), ("binary_error", C.c_double),
        ("iterations_run", C.c_uint32), ("trust_region_size", C.c_double),
    ]


class OrcTimers(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "build_problem", "j_evaluation_proj", "jtj", "schur_complement", "solve",
        "back_substitution", "evaluate_residuals", "apply_update", "total")]


def build(force=False):
    """Compile oracle/libba_oracle.so with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("ba_oracle.cpp", "ba_oracle.h", "omath.h", "outils.h", "oimu.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int]
        L.orc_create_calib.restype = C.c_void_p
        L.orc_create_calib.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        for name in ("orc_add_camera", "orc_add_pose", "orc_add_landmark",
                     "orc_add_projection_residual", "orc_add_unary_constraint",
                     "orc_add_binary_constraint", "orc_add_imu_residual", "orc_num_poses",
                     "orc_num_landmarks", "orc_num_proj_residuals", "orc_num_pose_params",
                     "orc_num_lm_params", "orc_num_calib_params"):
            getattr(L, name).restype = C.c_uint32
        L.orc_landmark_outlier_ratio.restype = C.c_double
        L.orc_get_camera_fov.restype = C.c_double
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def default_options():
    o = OrcOptions()
    lib().orc_default_options(C.byref(o))
    return o


class OracleBundleAdjuster:
    """CPU restatement of ba::BundleAdjuster<double, lm_dim, pose_dim, 0, do_tvs>."""

    def __init__(self, lm_dim=1, pose_dim=6, do_tvs=False, calib_size=0):
        self.L = lib()
        self.lm_dim, self.pose_dim, self.do_tvs, self.calib_size = lm_dim, pose_dim, bool(do_tvs), int(calib_size)
        self.h = C.c_void_p(self.L.orc_create_calib(lm_dim, pose_dim, int(calib_size), int(do_tvs)))
        if not self.h:
            raise ValueError("unsupported (lm_dim, pose_dim, calib_size, do_tvs)")

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- problem construction (reference API names) -----------------------
    def Init(self, options=None):
        self.options = options if options is not None else default_options()
        self.L.orc_init(self.h, C.byref(self.options))

    def SetGravity(self, g):
        g = _d(g)
        self.L.orc_set_gravity(self.h, _dp(g))

    def AddCamera(self, params, t_vs=(0, 0, 0, 0, 0, 0, 1)):
        """params (fx, fy, u0, v0): calibu::LinearCamera; (fx, fy, u0, v0, w): calibu::FovCamera."""
        p, t = _d(params), _d(t_vs)
        n = self.L.orc_add_camera(self.h, _dp(p[:4].copy()), _dp(t))
        if p.size == 5:
            self.L.orc_set_camera_fov(self.h, n - 1, C.c_double(p[4]))
        self._cam_fov = getattr(self, "_cam_fov", []) + [p.size == 5]
        return n

    def AddPose(self, t_wp, is_active=True, time=-1.0, v_w=(0, 0, 0), b=(0,) * 6):
        t, v, bb = _d(t_wp), _d(v_w), _d(b)
        return self.L.orc_add_pose(self.h, _dp(t), _dp(v), _dp(bb), int(is_active),
                                   C.c_double(time))

    def SetPoseCamParams(self, pose_cam_params):
        """Per-pose pinhole intrinsics (P x 4: fx, fy, u0, v0), the `cam_params` argument of the
        reference's AddPose overload (BundleAdjuster.h:292-323)."""
        a = _d(pose_cam_params).reshape(-1, 4)
        for i in range(a.shape[0]):
            row = np.ascontiguousarray(a[i])
            self.L.orc_set_pose_cam_params(self.h, i, _dp(row))

    def SetImuNoise(self, r6, rb6):
        """SetImuCalibration with other noise diagonals imu_.r / imu_.r_b (BundleAdjuster.h:567)."""
        a, b = _d(r6), _d(rb6)
        self.L.orc_set_imu_noise(self.h, _dp(a), _dp(b))

    def SetCalculateInertialCovarianceOnce(self, on=True):
        """Options::calculate_inertial_covariance_once (BundleAdjuster.h:106)."""
        self.L.orc_set_calculate_inertial_covariance_once(self.h, int(on))

    def SetUsePerPoseCamParams(self, on=True):
        """Options::use_per_pose_cam_params (BundleAdjuster.h:96)."""
        if self.L.orc_set_use_per_pose_cam_params(self.h, int(on)) != 0:
            raise ValueError("use_per_pose_cam_params: a pose has no camera parameters")

    def AddLandmark(self, x_w, ref_pose_id, ref_cam_id=0, is_active=True):
        x = _d(x_w)
        return self.L.orc_add_landmark(self.h, _dp(x), int(ref_pose_id), int(ref_cam_id),
                                       int(is_active))

    def AddProjectionResidual(self, z, meas_pose_id, landmark_id, cam_id=0, weight=1.0):
        zz = _d(z)
        return self.L.orc_add_projection_residual(self.h, _dp(zz), int(meas_pose_id),
                                                  int(landmark_id), int(cam_id),
                                                  C.c_double(weight))

    def AddUnaryConstraint(self, pose_id, t_wv, covariance, use_rotation=True):
        t, c = _d(t_wv), _d(covariance).reshape(36)
        return self.L.orc_add_unary_constraint(self.h, int(pose_id), _dp(t), _dp(c),
                                               int(use_rotation))

    def AddBinaryConstraint(self, p1, p2, t_12, covariance=None, weight=1.0,
                            use_rotation=True):
        t = _d(t_12)
        c = _d(np.eye(6) if covariance is None else covariance).reshape(36)
        return self.L.orc_add_binary_constraint(self.h, int(p1), int(p2), _dp(t), _dp(c),
                                                C.c_double(weight), int(use_rotation))

    def AddImuResidual(self, p1, p2, meas, weight=1.0):
        m = _d(meas).reshape(-1, 7)
        return self.L.orc_add_imu_residual(self.h, int(p1), int(p2), _dp(m), m.shape[0],
                                           C.c_double(weight))

    def RegularizePose(self, pose_id, translation, gravity, bias, rotation):
        self.L.orc_regularize_pose(self.h, int(pose_id), int(translation), int(gravity),
                                   int(bias), int(rotation))

    def SetRootPoseId(self, i):
        self.L.orc_set_root_pose_id(self.h, int(i))

    # -- bulk adders -------------------------------------------------------
    def add_poses(self, t_wp, v_w=None, b=None, is_active=None, time=None):
        t = _d(t_wp).reshape(-1, 7)
        n = t.shape[0]
        v = _d(v_w) if v_w is not None else None
        bb = _d(b) if b is not None else None
        act = np.ascontiguousarray(is_active, dtype=np.uint8) if is_active is not None else None
        tm = _d(time) if time is not None else None
        self.L.orc_add_poses(self.h, n, _dp(t), _dp(v) if v is not None else None,
                             _dp(bb) if bb is not None else None,
                             act.ctypes.data_as(c_u8_p) if act is not None else None,
                             _dp(tm) if tm is not None else None)

    def add_landmarks(self, x_w, ref_pose_id, ref_cam_id=None, is_active=None):
        x = _d(x_w).reshape(-1, 4)
        rp = np.ascontiguousarray(ref_pose_id, dtype=np.uint32)
        rc = np.ascontiguousarray(ref_cam_id, dtype=np.uint32) if ref_cam_id is not None else None
        act = np.ascontiguousarray(is_active, dtype=np.uint8) if is_active is not None else None
        self.L.orc_add_landmarks(self.h, x.shape[0], _dp(x), rp.ctypes.data_as(c_u32_p),
                                 rc.ctypes.data_as(c_u32_p) if rc is not None else None,
                                 act.ctypes.data_as(c_u8_p) if act is not None else None)

    def add_projection_residuals(self, z, meas_pose_id, landmark_id, cam_id=None, weight=None):
        zz = _d(z).reshape(-1, 2)
        mp = np.ascontiguousarray(meas_pose_id, dtype=np.uint32)
        li = np.ascontiguousarray(landmark_id, dtype=np.uint32)
        ci = np.ascontiguousarray(cam_id, dtype=np.uint32) if cam_id is not None else None
        w = _d(weight) if weight is not None else None
        ids = np.empty(zz.shape[0], dtype=np.uint32)
        self.L.orc_add_projection_residuals(
            self.h, zz.shape[0], _dp(zz), mp.ctypes.data_as(c_u32_p), li.ctypes.data_as(c_u32_p),
            ci.ctypes.data_as(c_u32_p) if ci is not None else None,
            _dp(w) if w is not None else None, ids.ctypes.data_as(c_u32_p))
        return ids

    # -- solve + results ---------------------------------------------------
    def _check_calib_camera(self):
        # parallel_algos.h:115-118 assigns dTransfer_dparams (2 x NumParams) to a 2 x CalibSize block
        if self.calib_size and (not getattr(self, "_cam_fov", [])
                                or (5 if self._cam_fov[0] else 4) != self.calib_size):
            raise ValueError("CalibSize %d needs camera 0 with that many parameters" % self.calib_size)

    def Solve(self, max_iter, gn_damping=1.0, error_increase_allowed=False):
        self._check_calib_camera()
        self.L.orc_solve(self.h, int(max_iter), C.c_double(gn_damping),
                         int(error_increase_allowed))

    def GetNumPoses(self):
        return self.L.orc_num_poses(self.h)

    def GetNumLandmarks(self):
        return self.L.orc_num_landmarks(self.h)

    def GetNumProjResiduals(self):
        return self.L.orc_num_proj_residuals(self.h)

    def poses(self):
        n = self.GetNumPoses()
        t, v, b = np.empty((n, 7)), np.empty((n, 3)), np.empty((n, 6))
        self.L.orc_get_poses(self.h, _dp(t), _dp(v), _dp(b))
        return t, v, b

    def landmarks(self):
        x = np.empty((self.GetNumLandmarks(), 4))
        self.L.orc_get_landmarks(self.h, _dp(x))
        return x

    def IsLandmarkReliable(self, i):
        return bool(self.L.orc_is_landmark_reliable(self.h, int(i)))

    def LandmarkOutlierRatio(self, i):
        return self.L.orc_landmark_outlier_ratio(self.h, int(i))

    def summary(self):
        s = OrcSummary()
        self.L.orc_get_summary(self.h, C.byref(s))
        return s

    def timers(self):
        t = OrcTimers()
        self.L.orc_get_timers(self.h, C.byref(t))
        return {n: getattr(t, n) for n, _ in OrcTimers._fields_}

    # -- parity taps -------------------------------------------------------
    def num_pose_params(self):
        return self.L.orc_num_pose_params(self.h)

    def num_lm_params(self):
        return self.L.orc_num_lm_params(self.h)

    def num_calib_params(self):
        return self.L.orc_num_calib_params(self.h)

    def delta_k(self):
        return self._vec(self.L.orc_get_delta_k, self.num_calib_params())

    def rhs_k(self):
        return self._vec(self.L.orc_get_rhs_k, self.num_calib_params())

    def camera_pose(self, cam_id=0):
        t = np.empty(7)
        self.L.orc_get_camera_pose(self.h, int(cam_id), _dp(t))
        return t

    def camera_params(self, cam_id=0):
        p = np.empty(4)
        self.L.orc_get_camera_params(self.h, int(cam_id), _dp(p))
        if getattr(self, "_cam_fov", [])[cam_id:cam_id + 1] == [True]:
            p = np.append(p, self.L.orc_get_camera_fov(self.h, int(cam_id)))
        return p

    def proj_calib_jacobians(self):
        """dz_dk per residual id in the j_kpr_ layout (intrinsics first, then T_vs), unweighted."""
        j = np.empty((self.GetNumProjResiduals(), 2, self.num_calib_params()))
        self.L.orc_get_proj_calib_jacobians(self.h, _dp(j))
        return j

    def calibration_marginals(self):
        k = self.num_calib_params()
        c = np.empty((k, k))
        self.L.orc_get_calibration_marginals(self.h, _dp(c))
        return c

    def proj_tvs_jacobians(self):
        j = np.empty((self.GetNumProjResiduals(), 2, 6))
        self.L.orc_get_proj_tvs_jacobians(self.h, _dp(j))
        return j

    def S(self):
        """(n + kCalibDim)^2: the pose block, then the calibration border."""
        n = self.num_pose_params() + self.num_calib_params()
        s = np.empty((n, n))
        self.L.orc_get_S(self.h, _dp(s))
        return s

    def _vec(self, fn, n):
        v = np.empty(n)
        fn(self.h, _dp(v))
        return v

    def rhs(self):
        return self._vec(self.L.orc_get_rhs, self.num_pose_params() + self.num_calib_params())

    def rhs_p(self):
        return self._vec(self.L.orc_get_rhs_p, self.num_pose_params())

    def rhs_l(self):
        return self._vec(self.L.orc_get_rhs_l, self.num_lm_params())

    def delta_p(self):
        return self._vec(self.L.orc_get_delta_p, self.num_pose_params())

    def delta_l(self):
        return self._vec(self.L.orc_get_delta_l, self.num_lm_params())

    def proj_weights(self):
        return self._vec(self.L.orc_get_proj_weights, self.GetNumProjResiduals())

    def proj_residuals(self):
        return self._vec(self.L.orc_get_proj_residuals, 2 * self.GetNumProjResiduals()).reshape(-1, 2)

    def imu_residuals(self):
        self.L.orc_num_imu_residuals.restype = C.c_uint32
        n = self.L.orc_num_imu_residuals(self.h)
        return self._vec(self.L.orc_get_imu_residuals, 15 * n).reshape(-1, 15)

    def proj_jacobians(self):
        n = self.GetNumProjResiduals()
        jm, jr = np.zeros((n, 2, 6)), np.zeros((n, 2, 6))
        jl = np.zeros((n, 2, max(self.lm_dim, 1)))
        self.L.orc_get_proj_jacobians(self.h, _dp(jm), _dp(jr), _dp(jl))
        return jm, jr, jl[:, :, :self.lm_dim]

    def imu_jacobians(self, i):
        a, b, c, r = np.zeros((15, 15)), np.zeros((15, 15)), np.zeros((15, 15)), np.zeros(15)
        self.L.orc_get_imu_jacobians(self.h, int(i), _dp(a), _dp(b), _dp(c), _dp(r))
        return a, b, c, r

    def binary_jacobians(self, i):
        a, b, r = np.zeros((6, 6)), np.zeros((6, 6)), np.zeros(6)
        self.L.orc_get_binary_jacobians(self.h, int(i), _dp(a), _dp(b), _dp(r))
        return a, b, r

    def unary_jacobian(self, i):
        a, r = np.zeros((6, 6)), np.zeros(6)
        self.L.orc_get_unary_jacobian(self.h, int(i), _dp(a), _dp(r))
        return a, r


# -- stand-alone math taps --------------------------------------------------
def dlog_dq(q):
    q, out = _d(q), np.empty((3, 4))
    lib().orc_math_dlog_dq(_dp(q), _dp(out))
    return out


def so3_log(q):
    q, out = _d(q), np.empty(3)
    lib().orc_math_so3_log(_dp(q), _dp(out))
    return out


def so3_exp(w):
    w, out = _d(w), np.empty(4)
    lib().orc_math_so3_exp(_dp(w), _dp(out))
    return out


def exp_decoupled(t, x):
    t, x, out = _d(t), _d(x), np.empty(7)
    lib().orc_math_exp_decoupled(_dp(t), _dp(x), _dp(out))
    return out


def log_decoupled(a, b):
    a, b, out = _d(a), _d(b), np.empty(6)
    lib().orc_math_log_decoupled(_dp(a), _dp(b), _dp(out))
    return out


def se3_mul(a, b):
    a, b, out = _d(a), _d(b), np.empty(7)
    lib().orc_math_se3_mul(_dp(a), _dp(b), _dp(out))
    return out


def se3_inv(a):
    a, out = _d(a), np.empty(7)
    lib().orc_math_se3_inv(_dp(a), _dp(out))
    return out


def set_num_threads(n):
    """Threads of the oracle's dense LDL^T: 1 = reference-faithful, > 1 = best-effort CPU mode."""
    lib().orc_set_num_threads(int(n))


def dense_solve_upper(s, rhs):
    s, rhs = _d(s), _d(rhs)
    x = np.empty(rhs.shape[0])
    lib().orc_math_dense_solve_upper(rhs.shape[0], _dp(s), _dp(rhs), _dp(x))
    return x


def integrate(pose_t, v, meas, bg, ba, g, r6=None, jac=False):
    pose_t, v, meas, bg, ba, g = map(_d, (pose_t, v, meas, bg, ba, g))
    meas = meas.reshape(-1, 7)
    out_t, out_v = np.empty(7), np.empty(3)
    db, c = np.zeros((10, 6)), np.zeros((10, 10))
    r = _d(r6) if r6 is not None else np.zeros(6)
    lib().orc_math_integrate(_dp(pose_t), _dp(v), _dp(meas), meas.shape[0], _dp(bg), _dp(ba),
                             _dp(g), _dp(r), _dp(out_t), _dp(out_v),
                             _dp(db) if jac else None, _dp(c) if jac else None)
    return (out_t, out_v, db, c) if jac else (out_t, out_v)


def lie(op, a, b=None):
    """Utils.h helper number `op` (numbering of ba_hip_lie, include/ba_hip.h) -> flat result."""
    a = _d(a)
    bb = _d(b) if b is not None else None
    out = np.empty(64)
    n = lib().orc_math_lie(int(op), _dp(a), _dp(bb) if bb is not None else None, _dp(out))
    return out[:n].copy()


def integrate_jacobians(pose_t, v, meas, bg, ba, g, r6):
    """dpose_db (10x6), dpose_dpose (10x10) and the covariance of IntegrateResidual (Types.h:662-738)."""
    pose_t, v, meas, bg, ba, g, r = map(_d, (pose_t, v, meas, bg, ba, g, r6))
    meas = meas.reshape(-1, 7)
    db, dd, c = np.zeros((10, 6)), np.zeros((10, 10)), np.zeros((10, 10))
    lib().orc_math_integrate_jacobians(_dp(pose_t), _dp(v), _dp(meas), meas.shape[0], _dp(bg), _dp(ba), _dp(g),
                                       _dp(r), _dp(db), _dp(dd), _dp(c))
    return db, dd, c


def transfer(params, t_ba, pix, rho, jac=False):
    """Transfer(T_ba, pix, rho) and (jac) its Jacobian w.r.t. the camera parameters: 2x4 for the
    pinhole (fx, fy, u0, v0), 2x5 for the FOV camera (fx, fy, u0, v0, w)."""
    p, t, x, out = _d(params), _d(t_ba), _d(pix), np.empty(2)
    J = np.empty((2, p.size))
    fn = lib().orc_math_transfer_fov if p.size == 5 else lib().orc_math_transfer
    fn(_dp(p), _dp(t), _dp(x), C.c_double(rho), _dp(out), _dp(J) if jac else None)
    return (out, J) if jac else out


def fov_project(params5, P):
    """FOV camera: Project(P), its 2x3 derivative over P, and Unproject(Project(P)) (the z = 1 ray)."""
    p, X, pix, d, ray = _d(params5), _d(P), np.empty(2), np.empty((2, 3)), np.empty(3)
    lib().orc_math_fov_project(_dp(p), _dp(X), _dp(pix), _dp(d), _dp(ray))
    return pix, d, ray
