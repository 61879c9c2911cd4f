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
