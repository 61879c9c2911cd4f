"""Race screen (scratch): the look-ahead / tile-sparse factorisation must be bitwise reproducible.
Solves the configs[1] scene and a 3000-pose scene repeatedly in one process and compares the
step bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ba_amd import hipapi, scene

def run(P, L, reps):
    sc = scene.make_scene(P, L, 10, lm_dim=1, seed=2)
    keep = np.ones(len(sc.obs_pose), dtype=bool); keep[::11] = False
    pa = np.ones(sc.num_poses, dtype=np.uint8); pa[sc.anchor_poses] = 0
    eng = hipapi.Engine(1, 6)
    eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1]); eng.set_poses(sc.poses, is_active=pa)
    eng.set_landmarks(sc.landmarks, sc.lm_ref_pose)
    eng.set_projection_residuals(sc.obs_z[keep], sc.obs_pose[keep], sc.obs_lm[keep])
    eng.finalize(); eng.begin_solve(); eng.set_pose_masks(np.zeros(sc.num_poses, dtype=np.uint16))
    ref = None
    for r in range(reps):
        eng.linearize()
        assert eng.solve_gn() == 0
        dp, dl = eng.get_delta_gn()
        if ref is None:
            ref = (dp.copy(), dl.copy())
        else:
            assert np.array_equal(dp, ref[0]) and np.array_equal(dl, ref[1]), "run %d differs" % r
    print("P=%d: %d identical solves" % (P, reps), flush=True)

run(1000, 100000, 40)
run(3000, 30000, 20)
