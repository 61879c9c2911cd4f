"""Writes tests/golden/config3_factor_tile_pattern.npz: the 64x64-tile pattern of the FACTOR of the reduced
pose system of BASELINE.json configs[3] (10k poses / 1M landmarks / 10M residuals, the bench.py scene,
seed 2, two anchor poses inactive), as the engine's symbolic elimination produced it
(ba_hip_get_factor_tile_pattern) — packed bits, 110 KB.  Needs a GPU (the structure is built on the
device); run once through gpurun:  python tests/golden/make_config3_pattern.py
Used by tests/test_dist_plan.py for the byte accounting of the distributed solve's message plan."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ba_amd import hipapi, scene  # noqa: E402

P, L, K = 10000, 1000000, 10
sc = scene.make_scene(P, L, K, lm_dim=1, seed=2)
keep = np.ones(len(sc.obs_pose), dtype=bool)
keep[::K + 1] = False
pa = np.ones(sc.num_poses, dtype=np.uint8)
pa[sc.anchor_poses] = 0
eng = hipapi.Engine(1, 6)
eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1])
eng.set_poses(sc.poses, is_active=pa)
eng.set_landmarks(sc.landmarks, sc.lm_ref_pose)
eng.set_projection_residuals(sc.obs_z[keep], sc.obs_pose[keep], sc.obs_lm[keep])
eng.finalize()
nz = eng.factor_tile_pattern()
out = os.path.join(ROOT, "gpurun_out", "config3_factor_tile_pattern.npz")
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, nblk=np.int64(nz.shape[0]), bits=np.packbits(nz, axis=None),
                    note=np.array("factor tile pattern of BASELINE configs[3], seed 2; tests/golden/make_config3_pattern.py"))
print("nblk", nz.shape[0], "lower tiles", int(nz.sum()), "of", nz.shape[0] * (nz.shape[0] + 1) // 2)
