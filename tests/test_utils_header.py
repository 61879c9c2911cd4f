"""include/ba/Utils.h — the reference's Lie-group / quaternion helpers (Utils.h:72-694) for host callers.
ba_hip_lie runs the code the kernels use (ba_amd/csrc/dmath.h, dpose.h) on the CPU; every op is compared
with the oracle's restatement of the same Utils.h function, the derivative ops also with central
differences of the function they differentiate.  No GPU involved."""
import ctypes as C
import os
import subprocess

import numpy as np

from ba_amd import hipapi, scene

DP = C.POINTER(C.c_double)


def _lie(op, a, b=None):
    L = hipapi.lib()
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    out = np.empty(64)
    n = L.ba_hip_lie(int(op), a.ctypes.data_as(DP), bb.ctypes.data_as(DP) if bb is not None else None,
                     out.ctypes.data_as(DP))
    assert n > 0, op
    return out[:n].copy()


def _rand_se3(rng, po):
    return np.concatenate([rng.normal(0, 1.0, 3), po.so3_exp(rng.normal(0, 0.6, 3))])


def test_every_helper_matches_the_oracle(oracle_lib):
    po = oracle_lib
    rng = np.random.default_rng(5)
    for _ in range(25):
        t1, t2 = _rand_se3(rng, po), _rand_se3(rng, po)
        q = po.so3_exp(rng.normal(0, 0.8, 3))
        w3, x3, x4, x6 = rng.normal(0, 0.5, 3), rng.normal(0, 1, 3), rng.normal(0, 1, 4), rng.normal(0, 0.3, 6)
        cases = [(1, q, None), (2, w3, None), (3, q, None), (4, q, None), (5, q, x3), (6, q, None), (7, t1, t2),
                 (8, t1, x6), (9, t1, t2), (10, t1, t2), (11, t1, t2), (12, t1, None), (13, t1, None), (14, t1, x4),
                 (15, t1, t2), (16, t1, None), (17, t1, x4)]
        for op, a, b in cases:
            mine, ref = _lie(op, a, b), po.lie(op, a, b)
            assert mine.shape == ref.shape, op
            assert np.abs(mine - ref).max() < 1e-13 * max(1.0, np.abs(ref).max()), (op, mine, ref)
    # small rotation vector: the series branch of dq_exp_dw / so3 exp
    assert np.abs(_lie(2, [1e-9, -2e-9, 0.0]) - po.lie(2, [1e-9, -2e-9, 0.0])).max() < 1e-15
    L = hipapi.lib()
    out = np.empty(64)
    assert L.ba_hip_lie(99, out.ctypes.data_as(DP), None, out.ctypes.data_as(DP)) == -1
    assert L.ba_hip_lie(7, out.ctypes.data_as(DP), None, out.ctypes.data_as(DP)) == -1    # second argument missing


def test_derivative_helpers_against_central_differences(oracle_lib):
    """the reference checks these the same way (_Test_dlog_dq, Utils.h:188-219; the commented blocks at :556-580)"""
    po = oracle_lib
    rng = np.random.default_rng(6)
    h = 1e-6
    t1, t2 = _rand_se3(rng, po), _rand_se3(rng, po)
    # dlog_decoupled_dx: d log_decoupled(exp_decoupled(a, x), b) / dx at 0
    J = _lie(9, t1, t2).reshape(6, 6)
    for j in range(6):
        e = np.zeros(6)
        e[j] = h
        fd = (_lie(7, _lie(8, t1, e), t2) - _lie(7, _lie(8, t1, -e), t2)) / (2 * h)
        assert np.abs(fd - J[:, j]).max() < 1e-6
    # dexp_decoupled_dx: d exp_decoupled(t, x) / dx at 0 as [t q]
    J = _lie(12, t1).reshape(7, 6)
    for j in range(6):
        e = np.zeros(6)
        e[j] = h
        fd = (_lie(8, t1, e) - _lie(8, t1, -e)) / (2 * h)
        assert np.abs(fd - J[:, j]).max() < 1e-6
    # dt_x_dt: d (T x) / d (t, q) with the quaternion entries perturbed freely (Utils.h:556-580)
    x4 = rng.normal(0, 1, 4)
    J = _lie(14, t1, x4).reshape(4, 7)

    def tx(t):
        # R(q) from the polynomial form in q (no renormalisation): the form the Jacobian differentiates
        R = _lie(6, t[3:]).reshape(3, 3)
        return np.append(R @ x4[:3] + t[:3] * x4[3], x4[3])
    for j in range(3):
        e = np.zeros(7)
        e[j] = h
        fd = (tx(t1 + e) - tx(t1 - e)) / (2 * h)
        assert np.abs(fd - J[:, j]).max() < 1e-6
    assert np.abs(_lie(17, t1, x4) - tx(t1)).max() < 1e-12
    # dq1q2_dq1 / dq1q2_dq2: the quaternion product is bilinear
    qa, qb = po.so3_exp(rng.normal(0, 0.8, 3)), po.so3_exp(rng.normal(0, 0.8, 3))
    prod = scene.quat_mul(qa[None], qb[None])[0]
    assert np.abs(_lie(4, qa).reshape(4, 4) @ qb - prod).max() < 1e-13
    assert np.abs(_lie(3, qb).reshape(4, 4) @ qa - prod).max() < 1e-13


def test_cpp_utils_header(tmp_path):
    """include/ba/Utils.h compiles stand-alone against the value types of Types.h and gives the documented
    values: log_decoupled(exp_decoupled(a, x), a) = x, MultHomogeneous, dq1q2_*, powi, Tic / Toc."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipapi.lib()
    src = tmp_path / "u.cpp"
    src.write_text(r'''
#include <ba/Utils.h>
#include <cmath>
#include <cstdio>
int main() {
  int bad = 0;
  const double t[3] = {1, 2, 3}, q[4] = {0.1825742, 0.3651484, 0.5477226, 0.7302967};
  const ba::SE3 a(t, q);
  const ba::Vector6t x({0.1, -0.2, 0.3, 0.05, -0.04, 0.03});
  const ba::SE3 b = ba::exp_decoupled(a, x);
  const ba::Vector6t back = ba::log_decoupled(b, a);
  for (int i = 0; i < 3; ++i) bad += std::fabs(back[i] - x[i]) > 1e-12;
  const ba::Vector4t p({1, 0, 0, 1});
  const ba::Vector4t y = ba::MultHomogeneous(a, p);
  const ba::Matrix3t R = a.rotationMatrix();
  bad += std::fabs(y[0] - (R(0, 0) + 1)) > 1e-6 || y[3] != 1;
  const ba::Mat<4, 4> L = ba::dq1q2_dq2(ba::Vector4t({q[0], q[1], q[2], q[3]}));
  bad += L(3, 3) != q[3] || L(0, 3) != q[0];
  bad += ba::dqinv_dq()(0, 0) != -1 || ba::powi(2.0, 10) != 1024.0 || ba::powi(2.0, -1) != 0.5;
  bad += ba::dt1_t2_dt2(a)(0, 0) != R(0, 0);
  bad += ba::dexp_decoupled_dx(a)(0, 0) != 1.0 || ba::dlog_decoupled_dx(a, a)(0, 0) != 1.0;
  const double t0 = ba::Tic();
  bad += ba::Toc(t0) < 0;
  std::printf(bad ? "utils header: FAIL %d\n" : "utils header: ok\n", bad);
  return bad;
}
''')
    exe = str(tmp_path / "u")
    libdir = os.path.join(root, "ba_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(root, "include"), str(src), "-o", exe,
                           "-L", libdir, "-lba_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "utils header: ok" in out.stdout, out.stdout + out.stderr


def test_math_test_application():
    """applications/math_test — the analytic-vs-numeric Jacobian report of the reference's program of that name
    (main.cpp:26-148) on include/ba/Utils.h: every norm small, exit code 0.  Host only."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipapi.lib()  # builds the libraries and applications if needed
    exe = os.path.join(root, "ba_amd", "lib", "math_test")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "math_test: ok" in out.stdout, out.stdout + out.stderr
    for name in ("dlog_dq", "dExp_dq", "dTerror", "dlog_Terror"):
        assert name in out.stdout
