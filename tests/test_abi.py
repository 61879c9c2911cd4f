"""The C-ABI libraries load without a GPU and export every symbol the headers declare
(include/ba_hip.h, include/ba_capi.h); the product fails loudly when no device is usable."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_ba_hip_exports_every_declared_symbol():
    from ba_amd import hipapi
    lib = hipapi.lib()
    names = [n for n in _declared("ba_hip.h", "ba_hip_") if n != "ba_hip_allreduce_fn"]
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hipapi.SYMBOLS) == names


def test_ba_capi_exports_every_declared_symbol():
    from ba_amd import adjuster
    lib = adjuster.lib()
    names = [n for n in _declared("ba_capi.h", "ba_") if not n.startswith("ba_hip_")]
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(adjuster.SYMBOLS) == names


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ba_amd import hipapi
    with pytest.raises(hipapi.HipError):
        hipapi.Engine(1, 6)


def test_host_class_add_semantics_match_oracle(oracle_lib):
    """Add* return values and the rejection of privileged-frame observations
    (BundleAdjuster.h:489-501) — host logic only, no Solve()."""
    from ba_amd import adjuster, scene
    from helpers import fill
    for lm_dim in (1, 3):
        sc = scene.make_scene(24, 30, 4, lm_dim=lm_dim, seed=5)
        a = adjuster.BundleAdjuster(lm_dim, 6)
        a.Init()
        o = oracle_lib.OracleBundleAdjuster(lm_dim, 6)
        o.Init()
        ia, io = fill(a, sc), fill(o, sc)
        assert np.array_equal(ia, io)
        assert a.GetNumProjResiduals() == o.GetNumProjResiduals() == 30 * 4
        assert a.GetNumPoses() == 24 and a.GetNumLandmarks() == 30
        assert a.AddCamera(sc.cam_params) == 2  # returns NumCams() after insertion


def test_solve_without_gpu_reports_solver_error(capfd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ba_amd import adjuster, scene
    from helpers import fill
    sc = scene.make_scene(24, 30, 4, lm_dim=1, seed=5)
    a = adjuster.BundleAdjuster(1, 6)
    a.Init()
    fill(a, sc)
    a.Solve(1)
    assert adjuster.RESULT_NAMES[a.summary().result] == "SolverError"
    assert "no usable HIP device" in capfd.readouterr().err
