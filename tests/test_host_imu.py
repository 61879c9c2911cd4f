"""ImuResidualT::IntegrateResidual on the host (reference Types.h:662-738): the C-ABI entry
ba_hip_integrate_imu runs the RK4 code of the device kernels (ba_amd/csrc/dpose.h) on the CPU —
no GPU involved — and is checked against the oracle's restatement of the same function."""
import ctypes as C
import os
import subprocess

import numpy as np

from ba_amd import hipapi, scene


def _integrate(t7, v, bg, ba, g, meas):
    L = hipapi.lib()
    dp = C.POINTER(C.c_double)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (t7, v, bg, ba, g, meas)]
    n = a[5].reshape(-1, 7).shape[0]
    out = np.empty((max(n, 1), 10))
    rc = L.ba_hip_integrate_imu(*[x.ctypes.data_as(dp) for x in a], C.c_uint32(n), out.ctypes.data_as(dp))
    assert rc == 0
    return out


def test_host_imu_integration_matches_oracle(oracle_lib):
    po = oracle_lib
    P = 14
    sc = scene.make_scene(P, 60, 5, lm_dim=1, seed=53)
    scene.add_inertial(sc, period=60.0 * P / 100.0)
    rng = np.random.default_rng(2)
    for i in range(P - 1):
        bg, ba = 1e-3 * rng.normal(size=3), 1e-2 * rng.normal(size=3)
        st = _integrate(sc.poses[i], sc.init_vel[i], bg, ba, sc.gravity, sc.imu_meas[i])
        t_ref, v_ref = po.integrate(sc.poses[i], sc.init_vel[i], sc.imu_meas[i], bg, ba, sc.gravity)
        assert st.shape[0] == len(sc.imu_meas[i])
        assert np.array_equal(st[0, :7], np.asarray(sc.poses[i], dtype=np.float64))   # the start state
        assert np.abs(st[-1, :7] - t_ref).max() < 1e-12 and np.abs(st[-1, 7:] - v_ref).max() < 1e-12
        # the intermediate states are the prefixes of the same integration
        k = len(sc.imu_meas[i]) // 2
        t_mid, v_mid = po.integrate(sc.poses[i], sc.init_vel[i], sc.imu_meas[i][:k + 1], bg, ba, sc.gravity)
        assert np.abs(st[k, :7] - t_mid).max() < 1e-12 and np.abs(st[k, 7:] - v_mid).max() < 1e-12


def test_host_imu_integration_jacobians_match_oracle(oracle_lib):
    """The Jacobian outputs of ImuResidualT::IntegrateResidual (reference Types.h:662-738) on the host:
    dpose_db, dpose_dpose and the covariance from ba_hip_integrate_imu_jacobians against the oracle, and
    GetPoseDerivative / IntegratePose (Types.h:324-416) against central differences."""
    po = oracle_lib
    L = hipapi.lib()
    dp = C.POINTER(C.c_double)
    P = 14
    sc = scene.make_scene(P, 60, 5, lm_dim=1, seed=54)
    scene.add_inertial(sc, period=60.0 * P / 100.0)
    rng = np.random.default_rng(3)
    r6 = np.array([1e-4] * 3 + [1e-2] * 3)

    def p(a):
        return a.ctypes.data_as(dp)
    for i in range(P - 1):
        bg, ba = 1e-3 * rng.normal(size=3), 1e-2 * rng.normal(size=3)
        t7, v = np.ascontiguousarray(sc.poses[i], dtype=np.float64), np.ascontiguousarray(sc.init_vel[i], dtype=np.float64)
        meas = np.ascontiguousarray(sc.imu_meas[i], dtype=np.float64).reshape(-1, 7)
        g = np.ascontiguousarray(sc.gravity, dtype=np.float64)
        st, db, dd, cov = np.empty((len(meas), 10)), np.empty((10, 6)), np.empty((10, 10)), np.zeros((10, 10))
        assert L.ba_hip_integrate_imu_jacobians(p(t7), p(v), p(bg), p(ba), p(g), p(meas), C.c_uint32(len(meas)), p(r6),
                                                p(st), p(db), p(dd), p(cov)) == 0
        assert np.array_equal(st, _integrate(t7, v, bg, ba, g, meas))   # the states of the Jacobian-free call
        db_o, dd_o, c_o = po.integrate_jacobians(t7, v, meas, bg, ba, g, r6)
        for a, b in ((db, db_o), (dd, dd_o), (cov, c_o)):
            assert np.abs(a - b).max() < 1e-11 * max(1.0, np.abs(b).max())
        # without a noise diagonal nothing is formed (the reference's `r != 0` condition)
        db2 = np.full((10, 6), 7.0)
        assert L.ba_hip_integrate_imu_jacobians(p(t7), p(v), p(bg), p(ba), p(g), p(meas), C.c_uint32(len(meas)), None,
                                                p(st), p(db2), None, None) == 0
        assert np.all(db2 == 0)
    # the reference's own checks of these matrices (_Test_IntegrateResidual_StateJacobian / _BiasJacobian,
    # Types.h:859-996): central differences of the integration over the start state — the four quaternion
    # entries perturbed freely, no renormalisation — and over the six bias entries
    i = 3
    t7, v = np.ascontiguousarray(sc.poses[i], dtype=np.float64), np.ascontiguousarray(sc.init_vel[i], dtype=np.float64)
    meas = np.ascontiguousarray(sc.imu_meas[i], dtype=np.float64).reshape(-1, 7)
    g = np.ascontiguousarray(sc.gravity, dtype=np.float64)
    bg, ba = np.array([1e-3, -2e-3, 5e-4]), np.array([1e-2, 2e-2, -1e-2])
    st, db, dd = np.empty((len(meas), 10)), np.empty((10, 6)), np.empty((10, 10))
    assert L.ba_hip_integrate_imu_jacobians(p(t7), p(v), p(bg), p(ba), p(g), p(meas), C.c_uint32(len(meas)), p(r6),
                                            p(st), p(db), p(dd), None) == 0
    h = 1e-6

    def final(s10, b6):
        return _integrate(s10[:7], s10[7:], b6[:3], b6[3:], g, meas)[-1]
    s10, b6 = np.concatenate([t7, v]), np.concatenate([bg, ba])
    for j in range(10):
        e = np.zeros(10)
        e[j] = h
        fd = (final(s10 + e, b6) - final(s10 - e, b6)) / (2 * h)
        assert np.abs(fd - dd[:, j]).max() < 1e-5 * max(1.0, np.abs(dd).max()), j
    for j in range(6):
        e = np.zeros(6)
        e[j] = h
        fd = (final(s10, b6 + e) - final(s10, b6 - e)) / (2 * h)
        assert np.abs(fd - db[:, j]).max() < 1e-5 * max(1.0, np.abs(db).max()), j
    # GetPoseDerivative / IntegratePose: Jacobians against central differences of the functions themselves
    s10 = np.concatenate([sc.poses[2], sc.init_vel[2]]).astype(np.float64)
    z0, z1 = np.ascontiguousarray(sc.imu_meas[2][0], dtype=np.float64), np.ascontiguousarray(sc.imu_meas[2][1], dtype=np.float64)
    bg, ba = np.array([1e-3, -2e-3, 5e-4]), np.array([1e-2, 2e-2, -1e-2])
    dt = 0.5 * (z1[6] - z0[6])

    def kder(s, b_g, b_a):
        k = np.empty(9)
        assert L.ba_hip_imu_pose_derivative(p(np.ascontiguousarray(s)), p(g), p(z0), p(z1), p(np.ascontiguousarray(b_g)),
                                            p(np.ascontiguousarray(b_a)), C.c_double(dt), p(k), None, None) == 0
        return k
    k, dk_db, dk_dx = np.empty(9), np.empty((9, 6)), np.empty((9, 10))
    assert L.ba_hip_imu_pose_derivative(p(s10), p(g), p(z0), p(z1), p(bg), p(ba), C.c_double(dt), p(k), p(dk_db), p(dk_dx)) == 0
    assert np.array_equal(k, kder(s10, bg, ba)) and np.array_equal(k[:3], s10[7:])
    h = 1e-6
    for j in range(6):
        e = np.zeros(6)
        e[j] = h
        fd = (kder(s10, bg + e[:3], ba + e[3:]) - kder(s10, bg - e[:3], ba - e[3:])) / (2 * h)
        assert np.abs(fd - dk_db[:, j]).max() < 1e-7
    for j in range(10):
        e = np.zeros(10)
        e[j] = h
        fd = (kder(s10 + e, bg, ba) - kder(s10 - e, bg, ba)) / (2 * h)
        assert np.abs(fd - dk_dx[:, j]).max() < 1e-6 * max(1.0, np.abs(dk_dx).max())

    def step(s, kk):
        o = np.empty(10)
        assert L.ba_hip_imu_integrate_pose(p(np.ascontiguousarray(s)), p(np.ascontiguousarray(kk)), C.c_double(dt), p(o), None, None) == 0
        return o
    y, dy_dk, dy_dy = np.empty(10), np.empty((10, 9)), np.empty((4, 4))
    assert L.ba_hip_imu_integrate_pose(p(s10), p(k), C.c_double(dt), p(y), p(dy_dk), p(dy_dy)) == 0
    assert np.array_equal(y, step(s10, k))
    assert np.abs(y[:3] - (s10[:3] + k[:3] * dt)).max() < 1e-15 and np.abs(y[7:] - (s10[7:] + k[6:] * dt)).max() < 1e-15
    for j in range(9):
        e = np.zeros(9)
        e[j] = h
        fd = (step(s10, k + e) - step(s10, k - e)) / (2 * h)
        assert np.abs(fd - dy_dk[:, j]).max() < 1e-7
    for j in range(4):
        e = np.zeros(10)
        e[3 + j] = h
        fd = (step(s10 + e, k) - step(s10 - e, k)) / (2 * h)
        assert np.abs(fd[3:7] - dy_dy[:, j]).max() < 1e-8


def test_host_imu_integration_degenerate_inputs():
    t7 = np.array([1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 1.0])
    z = np.zeros(3)
    g = np.array([0.0, 0.0, -9.8007])
    one = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 9.8007, 0.5]])
    st = _integrate(t7, z, z, z, g, one)          # a single sample: nothing to integrate
    assert st.shape == (1, 10) and np.array_equal(st[0, :7], t7)
    two = np.vstack([one, one])                     # zero time step (Types.h:430-436)
    st = _integrate(t7, z, z, z, g, two)
    assert np.array_equal(st[1], st[0])
    L = hipapi.lib()
    assert L.ba_hip_integrate_imu(None, None, None, None, None, None, C.c_uint32(0), None) == -1


def test_cpp_integrate_residual_wrappers(tmp_path):
    """include/ba/Types.h: ImuPoseT, ImuResidualT::IntegrateResidual / IntegrateImu compile against
    the header and run without a GPU (host code inside libba_hip.so)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipapi.lib()  # builds the library if needed
    exe = str(tmp_path / "imu_integrate_test")
    libdir = os.path.join(root, "ba_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "imu_integrate_test.cpp"), "-o", exe,
                           "-L", libdir, "-lba_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "imu integrate: ok" in out.stdout
