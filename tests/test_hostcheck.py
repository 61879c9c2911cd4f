"""Device math compiled for the host vs the oracle, no GPU needed.

`ba_amd/lib/libba_hostcheck.so` is a host build of the very headers the gfx950 kernels include
(`ba_amd/csrc/dmath.h`: closed-form projection Jacobians; `dpose.h`: unary / binary / IMU
residual blocks).  The oracle evaluates the reference's literal matrix chains
(/root/reference/include/ba/parallel_algos.h:35-152, 178-358; src/BundleAdjuster.cpp:1392-1482).
Agreement here is the CPU-side gate on the kernels' arithmetic; the `-m gpu` tests repeat it
on the device through the C-ABI.

Tolerance: both sides are FP64 evaluations of the same derivative with different
association order; 1e-11 relative to the block's norm (observed 1e-14..1e-13).
"""
import ctypes
import os

import numpy as np
import pytest

from ba_amd import scene
from helpers import accepted_obs, fill, gn_options, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ba_amd", "lib", "libba_hostcheck.so")


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@pytest.fixture(scope="module")
def hc():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


@pytest.fixture
def fov_switch(hc):
    """camera model of the hostcheck calls: back to the pinhole after the test"""
    yield hc.ba_hostcheck_set_fov
    hc.ba_hostcheck_set_fov(ctypes.c_double(0.0))


@pytest.mark.parametrize("fov", [False, True], ids=["pinhole", "fov_camera"])
@pytest.mark.parametrize("variant", [1, 0], ids=["proj_linearize", "proj_jacobians"])
@pytest.mark.parametrize("lm_dim", [1, 3])
def test_projection_jacobians_of_the_kernels_match_the_oracle(oracle_lib, hc, fov_switch, lm_dim, variant, fov):
    """variant 1: dmath.h proj_linearize — the form k_linearize evaluates (fewer transforms held in
    registers); variant 0: proj_jacobians, the literal closed form.  Both against the oracle's
    2x4 . 4x7 . 7x7 . 7x6 chains."""
    po = oracle_lib
    hc.ba_hostcheck_set_variant(variant)
    sc = scene.make_scene(30, 60, 5, lm_dim=lm_dim, seed=17)
    if fov:
        scene.to_fov_camera(sc, 0.93)
        fov_switch(ctypes.c_double(0.93))
    t_vs = np.array([0.05, -0.02, 0.1, 0.0, 0.0, 0.0, 1.0])
    t_vs[3:] = scene.quat_exp(np.array([0.02, -0.03, 0.01]))
    o = po.OracleBundleAdjuster(lm_dim, 6)
    o.Init(gn_options(po, apply_results=0))
    o.AddCamera(sc.cam_params, t_vs)
    o.add_poses(sc.poses)
    o.add_landmarks(sc.landmarks, sc.lm_ref_pose)
    o.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
    # x_s (LmSize 1) as Solve() forms it (BundleAdjuster.cpp:288-296): T_sw(ref) x_w, normalised
    o.Solve(1)
    jm_o, jr_o, jl_o = o.proj_jacobians()
    r_o = o.proj_residuals()
    acc = accepted_obs(sc)
    cam = np.asarray(sc.cam_params, dtype=np.float64)
    nsel = sc.obs_per_landmark + (1 if lm_dim == 1 else 0)
    zs = [sc.obs_z[i] for i in range(len(sc.obs_pose)) if not (lm_dim == 1 and i % nsel == 0)]
    worst = 0.0
    for rid, (pm, pr, l) in enumerate(acc):
        xw = np.asarray(sc.landmarks[l], dtype=np.float64)
        if lm_dim == 1:
            t_ws = po.se3_mul(np.asarray(sc.poses[pr], dtype=np.float64), t_vs)
            t_sw = po.se3_inv(t_ws)
            R = scene.quat_to_rot(t_sw[3:])
            p = R @ xw[:3] + t_sw[:3] * xw[3]
            nrm = np.linalg.norm(p)
            x = np.concatenate([p / nrm, [xw[3] / nrm]])
        else:
            x = xw
        r2 = np.zeros(2); jm = np.zeros(12); jr = np.zeros(12); jl = np.zeros(2 * lm_dim)
        z = np.asarray(zs[rid], dtype=np.float64)
        hc.ba_hostcheck_proj_jacobians(
            lm_dim, _dp(cam), _dp(z), _dp(x), _dp(np.asarray(sc.poses[pm], dtype=np.float64)), _dp(t_vs),
            _dp(np.asarray(sc.poses[pr], dtype=np.float64)), _dp(t_vs), int(lm_dim == 1 and pm == pr),
            _dp(r2), _dp(jm), _dp(jr), _dp(jl))
        worst = max(worst, rel_err(r2, r_o[rid]), rel_err(jm, jm_o[rid].ravel()),
                    rel_err(jl, jl_o[rid].ravel()))
        if lm_dim == 1:
            worst = max(worst, rel_err(jr, jr_o[rid].ravel()))
    assert len(acc) > 100
    assert worst < 1e-11, worst


def test_extrinsics_jacobian_of_the_kernels_matches_the_oracle(oracle_lib, hc):
    """dz_dtvs (DoTvs instantiations): the closed form of dmath.h proj_linearize<1, true> against the
    oracle's chain of parallel_algos.h:120-131."""
    po = oracle_lib
    sc = scene.make_scene(30, 60, 5, lm_dim=1, seed=19)
    t_vs = np.concatenate([[0.05, -0.02, 0.1], scene.quat_exp(np.array([0.2, -0.3, 0.1]))])
    o = po.OracleBundleAdjuster(1, 6, do_tvs=True)
    o.Init(gn_options(po, apply_results=0))
    o.AddCamera(sc.cam_params, t_vs)
    o.add_poses(sc.poses)
    o.add_landmarks(sc.landmarks, sc.lm_ref_pose)
    o.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
    o.Solve(1)
    jk_o = o.proj_tvs_jacobians()
    acc = accepted_obs(sc)
    cam = np.asarray(sc.cam_params, dtype=np.float64)
    nsel = sc.obs_per_landmark + 1
    zs = [sc.obs_z[i] for i in range(len(sc.obs_pose)) if i % nsel != 0]
    worst = 0.0
    for rid, (pm, pr, l) in enumerate(acc):
        xw = np.asarray(sc.landmarks[l], dtype=np.float64)
        t_sw = po.se3_inv(po.se3_mul(np.asarray(sc.poses[pr], dtype=np.float64), t_vs))
        p = scene.quat_to_rot(t_sw[3:]) @ xw[:3] + t_sw[:3] * xw[3]
        nrm = np.linalg.norm(p)
        x = np.concatenate([p / nrm, [xw[3] / nrm]])
        jk = np.zeros(12)
        hc.ba_hostcheck_proj_tvs_jacobian(
            _dp(cam), _dp(np.asarray(zs[rid], dtype=np.float64)), _dp(x),
            _dp(np.asarray(sc.poses[pm], dtype=np.float64)), _dp(t_vs),
            _dp(np.asarray(sc.poses[pr], dtype=np.float64)), _dp(t_vs), 0, _dp(jk))
        worst = max(worst, rel_err(jk, jk_o[rid].ravel()))
    assert len(acc) > 100 and np.abs(jk_o).max() > 10
    assert worst < 1e-11, worst


@pytest.mark.parametrize("fov", [False, True], ids=["pinhole", "fov_camera"])
def test_intrinsics_jacobian_of_the_kernels_matches_the_oracle(oracle_lib, hc, fov_switch, fov):
    """dz_dcam_params (CalibSize 4 / 5): dmath.h proj_intrinsics_rows against the oracle's
    -dTransfer_dparams(T_sw_m T_ws_r, z_ref, x_s(3)) (parallel_algos.h:115-118)."""
    po = oracle_lib
    sc = scene.make_scene(30, 60, 5, lm_dim=1, seed=19)
    if fov:
        scene.to_fov_camera(sc, 0.93)
        fov_switch(ctypes.c_double(0.93))
    K = 5 if fov else 4
    t_vs = np.concatenate([[0.05, -0.02, 0.1], scene.quat_exp(np.array([0.2, -0.3, 0.1]))])
    o = po.OracleBundleAdjuster(1, 6, calib_size=K)
    o.Init(gn_options(po, apply_results=0))
    o.AddCamera(sc.cam_params, t_vs)
    o.add_poses(sc.poses)
    o.add_landmarks(sc.landmarks, sc.lm_ref_pose)
    o.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
    o.Solve(1)
    jk_o = o.proj_calib_jacobians()
    acc = accepted_obs(sc)
    cam = np.asarray(sc.cam_params, dtype=np.float64)
    nsel = sc.obs_per_landmark + 1
    worst = 0.0
    for rid, (pm, pr, l) in enumerate(acc):
        xw = np.asarray(sc.landmarks[l], dtype=np.float64)
        t_sw = po.se3_inv(po.se3_mul(np.asarray(sc.poses[pr], dtype=np.float64), t_vs))
        p = scene.quat_to_rot(t_sw[3:]) @ xw[:3] + t_sw[:3] * xw[3]
        rho = xw[3] / np.linalg.norm(p)
        jk = np.zeros(12)
        hc.ba_hostcheck_proj_intrinsics_jacobian(
            _dp(cam), _dp(np.asarray(sc.obs_z[l * nsel], dtype=np.float64)), ctypes.c_double(rho),
            _dp(np.asarray(sc.poses[pm], dtype=np.float64)), _dp(t_vs),
            _dp(np.asarray(sc.poses[pr], dtype=np.float64)), _dp(t_vs), _dp(jk))
        jk = jk.reshape(2, 6)
        assert np.all(jk[:, K:] == 0)
        worst = max(worst, rel_err(jk[:, :K], jk_o[rid]))
    assert len(acc) > 100 and np.abs(jk_o).max() > 0.1
    assert worst < 1e-11, worst


@pytest.mark.parametrize("stale", [False, True], ids=["consistent", "after_rejected_step"])
def test_chain_with_cached_and_rig_extrinsics_matches_the_oracle(oracle_lib, hc, stale):
    """dmath.h proj_chain_two_tvs — the reference's Jacobian chains factor by factor with the cached
    T_sw and the rig's T_vs kept apart.  consistent: both equal, must reproduce the ordinary Jacobians.
    after_rejected_step: the oracle is driven into the state the reference is in after a rejected
    DoTvs step (poses restored with their cached T_sw, rig moved; BundleAdjuster.cpp:72-83, 1139-1149)
    and linearised there."""
    po = oracle_lib
    sc = scene.mount_camera(scene.make_scene(30, 90, 6, lm_dim=1, seed=11, roll_amp=0.6),
                            np.concatenate([[0.05, -0.02, 0.1], scene.quat_exp(np.array([0.02, -0.03, 0.01]))]))
    pa = np.ones(sc.num_poses, dtype=np.uint8)
    pa[::3] = 0
    t0 = po.exp_decoupled(sc.gt_t_vs, np.array([0.06, -0.05, 0.05, 0.02, -0.03, 0.02]))
    sc.landmarks = scene.remount_landmarks(sc, sc.gt_t_vs, t0)
    o = po.OracleBundleAdjuster(1, 6, do_tvs=True)
    o.Init(gn_options(po))
    o.AddCamera(sc.cam_params, t0)
    o.add_poses(sc.poses, is_active=pa)
    o.add_landmarks(sc.landmarks, sc.lm_ref_pose)
    o.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
    o.Solve(2)
    t_cache = o.camera_pose(0).copy()
    if stale:
        o.Solve(1, 40.0)  # overshoots: rejected, the rig keeps the step
        assert o.summary().result == 1
        assert np.linalg.norm(o.camera_pose(0) - t_cache) > 1e-4
    t_rig = o.camera_pose(0).copy()
    poses, lms = o.poses()[0].copy(), o.landmarks().copy()
    o.Solve(1)  # its linearisation happens in that state
    jm_o, jr_o, _ = o.proj_jacobians()
    jk_o = o.proj_tvs_jacobians()
    acc = accepted_obs(sc)
    cam = np.asarray(sc.cam_params, dtype=np.float64)
    worst = 0.0
    for rid, (pm, pr, l) in enumerate(acc):
        if not (pa[pm] or pa[pr]):
            continue  # the reference forms no pose / calibration Jacobians there (parallel_algos.h:88)
        t_sw = po.se3_inv(po.se3_mul(poses[pr], t_cache))
        p = scene.quat_to_rot(t_sw[3:]) @ lms[l, :3] + t_sw[:3] * lms[l, 3]
        nrm = np.linalg.norm(p)
        x = np.concatenate([p / nrm, [lms[l, 3] / nrm]])
        jm, jr, jk = np.zeros(12), np.zeros(12), np.zeros(12)
        hc.ba_hostcheck_proj_chain_two_tvs(_dp(cam), _dp(x), _dp(np.ascontiguousarray(poses[pm])),
                                           _dp(np.ascontiguousarray(poses[pr])), _dp(t_rig), _dp(t_cache), 0,
                                           _dp(jm), _dp(jr), _dp(jk))
        worst = max(worst, rel_err(jm, jm_o[rid].ravel()), rel_err(jr, jr_o[rid].ravel()),
                    rel_err(jk, jk_o[rid].ravel()))
    assert len(acc) > 100
    assert worst < 1e-10, worst


def test_unary_and_binary_blocks_of_the_kernels_match_the_oracle(oracle_lib, hc):
    po = oracle_lib
    rng = np.random.default_rng(5)
    gt, _ = scene.trajectory(8)
    init = gt.copy()
    for i in range(8):
        init[i] = po.exp_decoupled(init[i], rng.normal(0, 0.03, 6))
    o = po.OracleBundleAdjuster(0, 6)
    o.Init(gn_options(po, apply_results=0))
    o.add_poses(init)
    priors, bins = [], []
    for i in range(0, 8, 2):
        m = rng.normal(size=(6, 6))
        cov = 1e-3 * (m @ m.T + 6 * np.eye(6))
        prior = po.exp_decoupled(gt[i], rng.normal(0, 0.01, 6))
        rot = bool(i % 4 == 0)
        o.AddUnaryConstraint(i, prior, cov, rot)
        priors.append((i, prior, rot))
    for i in range(7):
        m = rng.normal(size=(6, 6))
        cov = 1e-4 * (m @ m.T + 6 * np.eye(6))
        t12 = po.exp_decoupled(po.se3_mul(po.se3_inv(gt[i]), gt[i + 1]), rng.normal(0, 0.003, 6))
        w = float(rng.uniform(0.5, 2.0))
        rot = bool(i % 3 != 0)
        o.AddBinaryConstraint(i, i + 1, t12, cov, w, rot)
        bins.append((i, i + 1, t12, cov, w, rot))
    o.Solve(1)
    for k, (i, prior, rot) in enumerate(priors):
        J_o, r_o = o.unary_jacobian(k)
        r6 = np.zeros(6); J = np.zeros(36)
        hc.ba_hostcheck_unary(_dp(np.ascontiguousarray(init[i])), _dp(np.ascontiguousarray(prior)), int(rot),
                              _dp(r6), _dp(J))
        assert rel_err(r6, r_o) < 1e-11
        assert rel_err(J, np.asarray(J_o).ravel()) < 1e-11
    for k, (p1, p2, t12, cov, w, rot) in enumerate(bins):
        dz1, dz2, r_o = o.binary_jacobians(k)
        ci = np.linalg.inv(cov)
        ev, evec = np.linalg.eigh(ci)
        cis = (evec * np.sqrt(ev)) @ evec.T
        h11 = np.zeros(225); h12 = np.zeros(225); h22 = np.zeros(225)
        g1 = np.zeros(15); g2 = np.zeros(15)
        eb = ctypes.c_double(); ee = ctypes.c_double()
        hc.ba_hostcheck_binary(_dp(np.ascontiguousarray(init[p1])), _dp(np.ascontiguousarray(init[p2])),
                               _dp(np.ascontiguousarray(t12)), _dp(np.ascontiguousarray(ci)),
                               _dp(np.ascontiguousarray(cis)), ctypes.c_double(w), int(rot), _dp(h11), _dp(h12),
                               _dp(h22), _dp(g1), _dp(g2), ctypes.byref(eb), ctypes.byref(ee))
        # the kernels accumulate J^T (w Sigma^-1) J directly (jt_pp_ * j_pp_, BundleAdjuster.cpp:357-372
        # with :1662-1668): compare with the oracle's un-whitened Jacobians
        dz1 = np.asarray(dz1).reshape(6, 6); dz2 = np.asarray(dz2).reshape(6, 6)
        H11 = dz1.T @ (w * ci) @ dz1
        H12 = dz1.T @ (w * ci) @ dz2
        H22 = dz2.T @ (w * ci) @ dz2
        assert rel_err(h11.reshape(15, 15)[:6, :6], H11) < 1e-10
        assert rel_err(h12.reshape(15, 15)[:6, :6], H12) < 1e-10
        assert rel_err(h22.reshape(15, 15)[:6, :6], H22) < 1e-10


@pytest.mark.parametrize("pose_dim", [9, 15])
def test_imu_residual_blocks_of_the_kernels_match_the_oracle(oracle_lib, hc, pose_dim):
    po = oracle_lib
    P = 30
    sc = scene.make_scene(P, 60, 5, lm_dim=1, seed=3)
    scene.add_inertial(sc, period=60.0 * P / 100.0)
    o = po.OracleBundleAdjuster(1, pose_dim)
    opt = gn_options(po, apply_results=0)
    o.Init(opt)
    o.SetGravity(sc.gravity)
    fill(o, sc)
    for i in range(P - 1):
        o.AddImuResidual(i, i + 1, sc.imu_meas[i])
    o.Solve(1)
    g = np.asarray(sc.gravity, dtype=np.float64)
    r6 = np.array([opt.gyro_sigma ** 2] * 3 + [opt.accel_sigma ** 2] * 3)
    rb6 = np.array([opt.gyro_bias_sigma ** 2] * 3 + [opt.accel_bias_sigma ** 2] * 3)
    for i in range(0, P - 1, 4):
        dz1_o, dz2_o, ci_o, r_o = o.imu_jacobians(i)
        p1 = np.concatenate([sc.poses[i], sc.init_vel[i], sc.init_bias[i]]).astype(np.float64)
        p2 = np.concatenate([sc.poses[i + 1], sc.init_vel[i + 1], sc.init_bias[i + 1]]).astype(np.float64)
        meas = np.ascontiguousarray(np.asarray(sc.imu_meas[i], dtype=np.float64))
        r15 = np.zeros(15); dz1 = np.zeros(225); dz2 = np.zeros(225); ci = np.zeros(225)
        hc.ba_hostcheck_imu(_dp(p1), _dp(p2), _dp(meas), int(meas.shape[0]), _dp(g), _dp(r6), _dp(rb6),
                            int(pose_dim), _dp(r15), _dp(dz1), _dp(dz2), _dp(ci))
        # the two-launch form of the device (step Jacobians per sample, then the sequential part)
        r15s = np.zeros(15); dz1s = np.zeros(225); dz2s = np.zeros(225); cis = np.zeros(225)
        hc.ba_hostcheck_imu_split(_dp(p1), _dp(p2), _dp(meas), int(meas.shape[0]), _dp(g), _dp(r6), _dp(rb6),
                                  int(pose_dim), _dp(r15s), _dp(dz1s), _dp(dz2s), _dp(cis))
        assert np.array_equal(r15s, r15) and np.array_equal(dz1s, dz1) and np.array_equal(dz2s, dz2)
        assert np.array_equal(cis, ci)
        R = pose_dim
        sl = np.s_[:R, :R]
        assert rel_err(r15[:R], np.asarray(r_o)[:R]) < 1e-10
        assert rel_err(dz1.reshape(15, 15)[sl], np.asarray(dz1_o).reshape(15, 15)[sl]) < 1e-9
        assert rel_err(dz2.reshape(15, 15)[sl], np.asarray(dz2_o).reshape(15, 15)[sl]) < 1e-9
        assert rel_err(ci.reshape(15, 15)[sl], np.asarray(ci_o).reshape(15, 15)[sl]) < 1e-8
