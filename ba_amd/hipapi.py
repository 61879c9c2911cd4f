"""ctypes binding of the C-ABI in include/ba_hip.h (ba_amd/lib/libba_hip.so).

The library is the product: if it is missing or no MI355X is usable this module fails
loudly — there is no CPU fallback anywhere in ba_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libba_hip.so")
if os.environ.get("BA_AMD_LIB"):   # A/B measurements of library builds (scratch/): another libba_hip.so
    LIB_PATH = os.path.abspath(os.environ["BA_AMD_LIB"])

dp = C.POINTER(C.c_double)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)


class Options(C.Structure):
    _fields_ = [("projection_outlier_threshold", C.c_double),
                ("use_robust_norm_for_proj_residuals", C.c_int32),
                ("use_robust_norm_for_inertial_residuals", C.c_int32),
                ("use_triangular_matrices", C.c_int32), ("keep_reduced_system", C.c_int32),
                ("gyro_sigma", C.c_double), ("accel_sigma", C.c_double),
                ("gyro_bias_sigma", C.c_double), ("accel_bias_sigma", C.c_double),
                ("pivot_rel_tolerance", C.c_double)]


class Errors(C.Structure):
    _fields_ = [("proj_error", C.c_double), ("binary_error", C.c_double),
                ("unary_error", C.c_double), ("inertial_error", C.c_double)]

    def total(self):
        return self.proj_error + self.binary_error + self.unary_error + self.inertial_error


class DoglegScalars(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("rhs_p_sq", "rhs_l_sq", "j_rhs_sq", "gn_p_sq",
                                          "gn_l_sq", "rhs_gn_p", "rhs_gn_l",
                                          "rhs_k_sq", "gn_k_sq", "rhs_gn_k")]


class StepNorms(C.Structure):
    _fields_ = [("step_p_norm", C.c_double), ("step_l_norm", C.c_double)]


class Timers(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("j_evaluation", "robust_weights", "jtj_schur", "solve",
                                          "back_substitution", "evaluate_residuals",
                                          "apply_update")]


class KernelStats(C.Structure):
    _fields_ = [("syrk_launches", C.c_uint32), ("gather_launches", C.c_uint32),
                ("landmarks_launches", C.c_uint32), ("imu_launches", C.c_uint32),
                ("syrk_ms", C.c_double), ("gather_ms", C.c_double), ("landmarks_ms", C.c_double),
                ("syrk_flops", C.c_double), ("imu_ms", C.c_double), ("pose_blocks_ms", C.c_double),
                ("pose_blocks_launches", C.c_uint32), ("reserved", C.c_uint32)]


class StructureStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("poses_active", "landmarks_active", "observations", "incidences",
                                          "factor_rows", "pair_blocks", "pair_entries", "tiles_lower",
                                          "tiles_S", "tiles_L", "tile_refs", "pose_entries", "linearize_waves",
                                          "factor_tile_products")]


class CommStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("chain_bytes_sent", "chain_bytes_recv", "side_bytes_sent", "side_bytes_recv",
                                          "reduce_scatter_bytes", "allreduce_bytes")] + \
               [(n, C.c_uint64) for n in ("chain_messages", "side_messages", "factorisations")]


class DistPlanStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("factor_bytes", "chain_recv_max", "chain_recv_total", "side_recv_max",
                                          "side_recv_total", "chain_sent_total", "side_sent_total", "recv_max",
                                          "backward_allreduce_bytes")] + \
               [(n, C.c_uint32) for n in ("messages_chain", "messages_side", "panels", "ranks", "classes", "kout")]


def dist_plan_stats(nblk, nz_lower, nranks, layout="auto", kout=0):
    """Byte accounting of the distributed solve's message plan (pure host: no device needed).
    nz_lower: (nblk, nblk) uint8 tile pattern of the factor, or None for a dense one."""
    out = DistPlanStats()
    ptr = None
    if nz_lower is not None:
        nz = np.ascontiguousarray(nz_lower, dtype=np.uint8)
        assert nz.shape == (nblk, nblk)
        ptr = nz.ctypes.data_as(C.POINTER(C.c_uint8))
    rc = lib().ba_hip_dist_plan_stats(int(nblk), ptr, int(nranks), layout.encode(), int(kout), C.byref(out))
    if rc != 0:
        raise ValueError("layout %r does not exist for %d ranks" % (layout, nranks))
    return {k: getattr(out, k) for k, _ in DistPlanStats._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int)

# every entry point include/ba_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "ba_hip_create", "ba_hip_destroy", "ba_hip_last_error", "ba_hip_set_options",
    "ba_hip_set_cameras", "ba_hip_set_pose_cam_params", "ba_hip_set_poses", "ba_hip_set_landmarks",
    "ba_hip_set_projection_residuals", "ba_hip_set_unary_residuals",
    "ba_hip_set_binary_residuals", "ba_hip_set_imu_residuals", "ba_hip_set_inertial_covariance_once", "ba_hip_set_imu_noise", "ba_hip_integrate_imu", "ba_hip_set_gravity",
    "ba_hip_finalize", "ba_hip_begin_solve", "ba_hip_set_pose_masks", "ba_hip_linearize",
    "ba_hip_solve_gn", "ba_hip_dogleg_terms", "ba_hip_compose_step", "ba_hip_apply_step",
    "ba_hip_rollback", "ba_hip_eval_residuals", "ba_hip_end_solve", "ba_hip_get_poses",
    "ba_hip_get_landmarks", "ba_hip_get_landmark_flags", "ba_hip_num_pose_params",
    "ba_hip_num_lm_params", "ba_hip_get_S", "ba_hip_get_rhs", "ba_hip_get_delta_gn",
    "ba_hip_get_step", "ba_hip_get_proj_weights", "ba_hip_get_proj_residuals", "ba_hip_get_imu_residuals", "ba_hip_get_imu_errors", "ba_hip_get_timers", "ba_hip_get_unary_scales", "ba_hip_device_buffer",
    "ba_hip_set_allreduce", "ba_hip_set_collectives", "ba_hip_solve_is_distributed", "ba_hip_dense_solve", "ba_hip_select_kth", "ba_hip_set_profiling",
    "ba_hip_get_kernel_stats", "ba_hip_check_solve", "ba_hip_get_structure_stats", "ba_hip_debug_set", "ba_hip_set_conditioning_residuals",
    "ba_hip_get_conditioning_error", "ba_hip_comm_unique_id", "ba_hip_comm_init", "ba_hip_comm_destroy", "ba_hip_allreduce_host", "ba_hip_get_proj_jacobians",
    "ba_hip_set_calibration", "ba_hip_num_calib_params", "ba_hip_get_cameras", "ba_hip_get_calib_jacobians", "ba_hip_get_calibration_marginals", "ba_hip_set_landmark_ref_pixels", "ba_hip_get_camera_params",
    "ba_hip_set_camera_models", "ba_hip_get_camera_fov",
    "ba_hip_integrate_imu_jacobians", "ba_hip_imu_pose_derivative", "ba_hip_imu_integrate_pose", "ba_hip_lie",
    "ba_hip_get_comm_stats", "ba_hip_reset_comm_stats", "ba_hip_dist_plan_stats", "ba_hip_get_factor_tile_pattern",
]


def build(force=False):
    """Compile the gfx950 engine in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean", "-s"])
    subprocess.check_call(["make", "-C", src, "-j4", "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ba_amd: %s is missing — build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 /
        # libhsa-runtime64.  If the engine pulled in /opt/rocm's copies first, a later
        # `import torch` would bring a second HSA runtime into the process and torch would
        # see no GPU (measured on the MI355X box).  Importing torch first makes the engine
        # bind to torch's runtime (same SONAME), so torch.distributed/RCCL and the engine
        # share one device context.  C++ users without torch link /opt/rocm directly.
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: the engine uses the ROCm runtime it was linked with
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.ba_hip_last_error.restype = C.c_char_p
        _lib.ba_hip_num_pose_params.restype = C.c_uint32
        _lib.ba_hip_num_lm_params.restype = C.c_uint32
        _lib.ba_hip_num_calib_params.restype = C.c_uint32
    return _lib


class HipError(RuntimeError):
    pass


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class Engine:
    """Thin object wrapper of one ba_hip_engine (one device)."""

    def __init__(self, lm_dim=1, pose_dim=6, device=0, stream=None):
        self.L = lib()
        self.lm_dim, self.pose_dim = lm_dim, pose_dim
        self.h = C.c_void_p()
        rc = self.L.ba_hip_create(lm_dim, pose_dim, device, C.c_void_p(stream), C.byref(self.h))
        if rc != 0 or not self.h:
            raise HipError("ba_hip_create failed (rc=%d): no usable HIP device — "
                           "the engine has no CPU fallback" % rc)
        self._cb = None

    def close(self):
        if self.h:
            self.L.ba_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, positive_ok=False):
        if rc < 0 or (rc > 0 and not positive_ok):
            raise HipError("ba_hip call failed rc=%d: %s" %
                           (rc, self.L.ba_hip_last_error(self.h).decode()))
        return rc

    # -- upload ----------------------------------------------------------------
    def set_options(self, opt):
        self._chk(self.L.ba_hip_set_options(self.h, C.byref(opt)))

    def set_cameras(self, params, t_vs):
        """params: C x 4 (calibu::LinearCamera) or C x 5 (calibu::FovCamera: fx, fy, u0, v0, w)."""
        p = np.atleast_2d(_d(params))
        if p.shape[1] not in (4, 5):
            p = p.reshape(-1, 4)
        t = _d(t_vs).reshape(-1, 7)
        p4 = np.ascontiguousarray(p[:, :4])
        self._chk(self.L.ba_hip_set_cameras(self.h, p4.shape[0], _p(p4, dp), _p(t, dp)))
        if p.shape[1] == 5:
            self.set_camera_models(np.ones(p.shape[0], dtype=np.int32), p[:, 4])

    def set_camera_models(self, model, w):
        m, ww = np.ascontiguousarray(model, dtype=np.int32), _d(w)
        self._chk(self.L.ba_hip_set_camera_models(self.h, m.shape[0], m.ctypes.data_as(C.POINTER(C.c_int32)), _p(ww, dp)))

    def get_camera_fov(self, n):
        w = np.empty(n)
        self._chk(self.L.ba_hip_get_camera_fov(self.h, int(n), _p(w, dp)))
        return w

    def set_pose_cam_params(self, params):
        """P x 4 pinhole intrinsics per pose (use_per_pose_cam_params), None = rig camera."""
        if params is None:
            self._chk(self.L.ba_hip_set_pose_cam_params(self.h, 0, None))
        else:
            p = _d(params).reshape(-1, 4)
            self._chk(self.L.ba_hip_set_pose_cam_params(self.h, p.shape[0], _p(p, dp)))

    def set_poses(self, t_wp, v_w=None, b=None, is_active=None):
        t = _d(t_wp).reshape(-1, 7)
        v = _d(v_w) if v_w is not None else None
        bb = _d(b) if b is not None else None
        a = np.ascontiguousarray(is_active, dtype=np.uint8) if is_active is not None else None
        self._chk(self.L.ba_hip_set_poses(self.h, t.shape[0], _p(t, dp), _p(v, dp), _p(bb, dp),
                                          _p(a, u8p)))

    def set_landmarks(self, x_w, ref_pose, ref_cam=None, is_active=None):
        x = _d(x_w).reshape(-1, 4)
        rp = np.ascontiguousarray(ref_pose, dtype=np.uint32)
        rc = np.ascontiguousarray(ref_cam, dtype=np.uint32) if ref_cam is not None else None
        a = np.ascontiguousarray(is_active, dtype=np.uint8) if is_active is not None else None
        self._chk(self.L.ba_hip_set_landmarks(self.h, x.shape[0], _p(x, dp), _p(rp, u32p),
                                              _p(rc, u32p), _p(a, u8p)))

    def set_projection_residuals(self, z, pose, lm, cam=None, weight=None):
        zz = _d(z).reshape(-1, 2)
        pp = np.ascontiguousarray(pose, dtype=np.uint32)
        ll = np.ascontiguousarray(lm, dtype=np.uint32)
        cc = np.ascontiguousarray(cam, dtype=np.uint32) if cam is not None else None
        ww = _d(weight) if weight is not None else None
        self._chk(self.L.ba_hip_set_projection_residuals(self.h, zz.shape[0], _p(zz, dp),
                                                         _p(pp, u32p), _p(ll, u32p), _p(cc, u32p),
                                                         _p(ww, dp)))

    def finalize(self):
        self._chk(self.L.ba_hip_finalize(self.h))

    # -- phases ----------------------------------------------------------------
    def begin_solve(self):
        self._chk(self.L.ba_hip_begin_solve(self.h))

    def set_pose_masks(self, masks):
        m = np.ascontiguousarray(masks, dtype=np.uint16)
        self._chk(self.L.ba_hip_set_pose_masks(self.h, m.shape[0], _p(m, u16p)))

    def linearize(self):
        e = Errors()
        self._chk(self.L.ba_hip_linearize(self.h, C.byref(e)))
        return e

    def solve_gn(self):
        return self._chk(self.L.ba_hip_solve_gn(self.h), positive_ok=True)

    def dogleg_terms(self, gn_available):
        s = DoglegScalars()
        self._chk(self.L.ba_hip_dogleg_terms(self.h, int(gn_available), C.byref(s)))
        return s

    def compose_step(self, coef_rhs, coef_gn):
        n = StepNorms()
        self._chk(self.L.ba_hip_compose_step(self.h, C.c_double(coef_rhs), C.c_double(coef_gn),
                                             C.byref(n)))
        return n

    def apply_step(self):
        self._chk(self.L.ba_hip_apply_step(self.h))

    def rollback(self):
        self._chk(self.L.ba_hip_rollback(self.h))

    def eval_residuals(self):
        e = Errors()
        self._chk(self.L.ba_hip_eval_residuals(self.h, C.byref(e)))
        return e

    def end_solve(self):
        self._chk(self.L.ba_hip_end_solve(self.h))

    # -- results -----------------------------------------------------------------
    def num_pose_params(self):
        return self.L.ba_hip_num_pose_params(self.h)

    def num_lm_params(self):
        return self.L.ba_hip_num_lm_params(self.h)

    def get_poses(self, n):
        t, v, b = np.empty((n, 7)), np.empty((n, 3)), np.empty((n, 6))
        self._chk(self.L.ba_hip_get_poses(self.h, _p(t, dp), _p(v, dp), _p(b, dp)))
        return t, v, b

    def get_landmarks(self, n):
        x = np.empty((n, 4))
        self._chk(self.L.ba_hip_get_landmarks(self.h, _p(x, dp)))
        return x

    def get_landmark_flags(self, n):
        r, o = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint32)
        self._chk(self.L.ba_hip_get_landmark_flags(self.h, _p(r, u8p), _p(o, u32p)))
        return r, o

    def num_calib_params(self):
        return self.L.ba_hip_num_calib_params(self.h)

    def set_calibration(self, calib_size=0, do_tvs=True):
        """CalibSize / DoTvs of the reference's class template; before finalize()."""
        self._chk(self.L.ba_hip_set_calibration(self.h, int(calib_size), int(do_tvs)))

    def get_calibration_marginals(self):
        k = self.num_calib_params()
        c = np.empty((k, k))
        self._chk(self.L.ba_hip_get_calibration_marginals(self.h, _p(c, dp)))
        return c

    def set_landmark_ref_pixels(self, z_ref):
        z = _d(z_ref).reshape(-1, 2)
        self._chk(self.L.ba_hip_set_landmark_ref_pixels(self.h, z.shape[0], _p(z, dp)))

    def get_camera_params(self, n):
        p = np.empty((n, 4))
        self._chk(self.L.ba_hip_get_camera_params(self.h, int(n), _p(p, dp)))
        return p

    def get_cameras(self, n):
        t = np.empty((n, 7))
        self._chk(self.L.ba_hip_get_cameras(self.h, int(n), _p(t, dp)))
        return t

    def get_calib_jacobians(self, n):
        """sqrt(w) dz_dtvs (2x6) per residual id of the last linearisation."""
        j = np.empty((n, 2, 6))
        self._chk(self.L.ba_hip_get_calib_jacobians(self.h, _p(j, dp)))
        return j

    def get_S(self):
        """(n + K)^2 with K calibration unknowns behind the n pose unknowns."""
        n = self.num_pose_params() + self.num_calib_params()
        s = np.empty((n, n))
        self._chk(self.L.ba_hip_get_S(self.h, _p(s, dp)))
        return s

    def get_rhs(self):
        n, nl = self.num_pose_params() + self.num_calib_params(), self.num_lm_params()
        a, b, c = np.empty(n), np.empty(n), np.zeros(max(nl, 1))
        self._chk(self.L.ba_hip_get_rhs(self.h, _p(a, dp), _p(b, dp), _p(c, dp)))
        return a, b, c[:nl]

    def get_delta_gn(self):
        n, nl = self.num_pose_params() + self.num_calib_params(), self.num_lm_params()
        a, c = np.empty(n), np.zeros(max(nl, 1))
        self._chk(self.L.ba_hip_get_delta_gn(self.h, _p(a, dp), _p(c, dp)))
        return a, c[:nl]

    def get_step(self):
        n, nl = self.num_pose_params() + self.num_calib_params(), self.num_lm_params()
        a, c = np.empty(n), np.zeros(max(nl, 1))
        self._chk(self.L.ba_hip_get_step(self.h, _p(a, dp), _p(c, dp)))
        return a, c[:nl]

    def get_proj_weights(self, n):
        w = np.empty(n)
        self._chk(self.L.ba_hip_get_proj_weights(self.h, _p(w, dp)))
        return w

    def get_proj_jacobians(self, n):
        """sqrt(w)-weighted (dz_dx_meas, dz_dx_ref, dz_dlm, r) per residual id of the last linearisation."""
        lm = max(self.lm_dim, 1)
        jm, jr, jl, r = np.zeros((n, 2, 6)), np.zeros((n, 2, 6)), np.zeros((n, 2, lm)), np.zeros((n, 2))
        self._chk(self.L.ba_hip_get_proj_jacobians(self.h, _p(jm, dp), _p(jr, dp), _p(jl, dp), _p(r, dp)))
        return jm, jr, jl[:, :, :self.lm_dim], r

    def get_timers(self):
        t = Timers()
        self._chk(self.L.ba_hip_get_timers(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in Timers._fields_}

    def check_solve(self):
        """(|S delta_gn - rhs|, |rhs|) formed on the device from the kept copy of S."""
        a, b = C.c_double(), C.c_double()
        self._chk(self.L.ba_hip_check_solve(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def structure_stats(self):
        st = StructureStats()
        self._chk(self.L.ba_hip_get_structure_stats(self.h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in StructureStats._fields_}

    def debug_set(self, key, value):
        self._chk(self.L.ba_hip_debug_set(self.h, int(key), int(value)))

    def set_profiling(self, on):
        self._chk(self.L.ba_hip_set_profiling(self.h, int(on)))

    def kernel_stats(self):
        k = KernelStats()
        self._chk(self.L.ba_hip_get_kernel_stats(self.h, C.byref(k)))
        return k

    def set_allreduce(self, fn, rank, nranks):
        """fn(dev_ptr:int, count:int, dtype:int) -> int (0 = ok); kept alive by this object."""
        if fn is None:
            self._cb = None
            self._chk(self.L.ba_hip_set_allreduce(self.h, None, None, 0, 1))
            return
        self._cb = ALLREDUCE_FN(lambda ctx, ptr, count, dtype: int(fn(ptr, count, dtype)))
        self._chk(self.L.ba_hip_set_allreduce(self.h, self._cb, None, int(rank), int(nranks)))

    def set_collectives(self, fn):
        """fn(op:int, dev_ptr:int, count:int, root:int) -> int (0 = ok); op 1 = broadcast from
        root, op 2 = in-place reduce-scatter of nranks chunks of `count` doubles.  Switches the
        reduced solve from replicated to distributed (ba_hip.h: ba_hip_set_collectives)."""
        if fn is None:
            self._cb2 = None
            self._chk(self.L.ba_hip_set_collectives(self.h, None, None))
            return
        self._cb2 = COLLECTIVE_FN(lambda ctx, op, ptr, count, root: int(fn(op, ptr, count, root)))
        self._chk(self.L.ba_hip_set_collectives(self.h, self._cb2, None))

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (rank 0 creates it, the other ranks receive it out of band)."""
        buf = C.create_string_buffer(128)
        if lib().ba_hip_comm_unique_id(buf) != 0:
            raise HipError("ba_hip_comm_unique_id failed (librccl not loadable?)")
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        """Native RCCL communicator inside the engine (collective over all ranks)."""
        self._chk(self.L.ba_hip_comm_init(self.h, C.c_char_p(unique_id), int(rank), int(nranks)))

    def comm_destroy(self):
        self._chk(self.L.ba_hip_comm_destroy(self.h))

    def solve_is_distributed(self):
        return bool(self.L.ba_hip_solve_is_distributed(self.h))

    def comm_stats(self, reset=False):
        """Bytes this rank moved per stream of the communicator(s) since the last reset (ba_hip_comm_stats)."""
        out = CommStats()
        self._chk(self.L.ba_hip_get_comm_stats(self.h, C.byref(out)))
        if reset:
            self._chk(self.L.ba_hip_reset_comm_stats(self.h))
        return {k: getattr(out, k) for k, _ in CommStats._fields_}

    def factor_tile_pattern(self):
        """(nblk, nblk) uint8 lower tile pattern of the factor of the reduced system."""
        nblk = (self.num_pose_params() + self.num_calib_params() + 63) // 64
        nblk = max(nblk, 1)
        out = np.zeros((nblk, nblk), dtype=np.uint8)
        self._chk(self.L.ba_hip_get_factor_tile_pattern(self.h, nblk, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def dense_solve(self, a_lower, b):
        a, b = _d(a_lower), _d(b)
        x = np.empty(b.shape[0])
        rc = self._chk(self.L.ba_hip_dense_solve(self.h, b.shape[0], _p(a, dp), _p(b, dp), _p(x, dp)),
                       positive_ok=True)
        return x, rc

    def select_kth(self, values, k):
        v = _d(values)
        out = C.c_double()
        self._chk(self.L.ba_hip_select_kth(self.h, v.shape[0], _p(v, dp), int(k), C.byref(out)))
        return out.value
