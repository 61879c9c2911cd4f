"""Generates the committed golden fixtures from the CPU oracle (oracle/ba_oracle.cpp).

The reference holds no golden numeric vectors and cannot be built in this image
(SURVEY.md §8c), so these vectors are outputs of the restatement, pinned by
tests/test_oracle_fd.py and tests/test_oracle_dense.py.  Each fixture stores the
complete inputs (poses, landmarks, observations) so it does not depend on the scene
generator staying unchanged.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ba_amd import scene  # noqa: E402
from helpers import fill, gn_options  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, lm_dim, P, L, K, seed, iters, dogleg=0, fov_w=None):
    sc = scene.make_scene(P, L, K, lm_dim=lm_dim, seed=seed)
    if fov_w:
        scene.to_fov_camera(sc, fov_w)  # cam_params gets its fifth entry: a calibu::FovCamera
    act = np.ones(P, dtype=np.uint8)
    act[sc.anchor_poses] = 0
    ba = po.OracleBundleAdjuster(lm_dim, 6)
    ba.Init(gn_options(po, use_dogleg=dogleg))
    fill(ba, sc, active=act)
    ba.Solve(1)
    out = dict(lm_dim=lm_dim, cam_params=sc.cam_params, poses=sc.poses, pose_active=act,
               landmarks=sc.landmarks, lm_ref_pose=sc.lm_ref_pose, obs_z=sc.obs_z,
               obs_pose=sc.obs_pose, obs_lm=sc.obs_lm, iters=iters, use_dogleg=dogleg,
               S_it0=ba.S(), rhs_it0=ba.rhs(), delta_p_it0=ba.delta_p(), delta_l_it0=ba.delta_l(),
               weights_it0=ba.proj_weights(), proj_error_it0=ba.summary().proj_error)
    for _ in range(iters - 1):
        ba.Solve(1)
    t, _, _ = ba.poses()
    out.update(poses_final=t, landmarks_final=ba.landmarks(),
               proj_error_final=ba.summary().proj_error, delta_norm_final=ba.summary().delta_norm)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "n =", out["S_it0"].shape[0], "proj_error", out["proj_error_it0"], "->",
          out["proj_error_final"])


if __name__ == "__main__":
    # BASELINE.json configs[0]: 50 poses / 200 landmarks / 2k reprojection residuals
    make("config1_lm1", 1, 50, 200, 10, seed=101, iters=4)
    make("config1_lm3", 3, 50, 200, 10, seed=103, iters=4)
    make("config1_lm1_dogleg", 1, 50, 200, 10, seed=105, iters=4, dogleg=1)
    make("config1_lm1_fov", 1, 50, 200, 10, seed=107, iters=4, fov_w=0.93)
