"""Oracle vs the committed golden fixtures (tests/golden/*.npz, made by
tests/golden/make_golden.py).  Guards the oracle against silent drift; the GPU side is
checked against the same fixtures in tests/test_gpu_parity.py."""
import glob
import os

import numpy as np
import pytest

from helpers import gn_options, rel_err

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1_*.npz")))


def load_into(ba, g):
    ba.AddCamera(g["cam_params"])
    ba.add_poses(g["poses"], is_active=g["pose_active"])
    ba.add_landmarks(g["landmarks"], g["lm_ref_pose"])
    return ba.add_projection_residuals(g["obs_z"], g["obs_pose"], g["obs_lm"])


def test_fixtures_exist():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(oracle_lib, path):
    po = oracle_lib
    g = np.load(path)
    ba = po.OracleBundleAdjuster(int(g["lm_dim"]), 6)
    ba.Init(gn_options(po, use_dogleg=int(g["use_dogleg"])))
    load_into(ba, g)
    ba.Solve(1)
    assert rel_err(ba.S(), g["S_it0"]) < 1e-12
    assert rel_err(ba.rhs(), g["rhs_it0"]) < 1e-12
    assert rel_err(ba.delta_p(), g["delta_p_it0"]) < 1e-9
    assert rel_err(ba.delta_l(), g["delta_l_it0"]) < 1e-9
    for _ in range(int(g["iters"]) - 1):
        ba.Solve(1)
    t, _, _ = ba.poses()
    assert rel_err(t, g["poses_final"]) < 1e-9
    assert rel_err(ba.landmarks(), g["landmarks_final"]) < 1e-9
