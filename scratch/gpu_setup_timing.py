"""Where the one-off setup time of a Solve() goes at configs[3] (scratch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ba_amd import hipapi, scene
P, L, K = (int(sys.argv[1]), int(sys.argv[2]), 10) if len(sys.argv) > 2 else (10000, 1000000, 10)
t = time.time(); sc = scene.make_scene(P, L, K, lm_dim=1, seed=2); print("scene %.2f s" % (time.time() - t))
t = time.time()
keep = np.ones(len(sc.obs_pose), dtype=bool); keep[::K + 1] = False
z, op, ol = sc.obs_z[keep], sc.obs_pose[keep], sc.obs_lm[keep]
pa = np.ones(sc.num_poses, dtype=np.uint8); pa[sc.anchor_poses] = 0
print("numpy select %.2f s" % (time.time() - t))
eng = hipapi.Engine(1, 6)
t = time.time()
eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1]); eng.set_poses(sc.poses, is_active=pa)
eng.set_landmarks(sc.landmarks, sc.lm_ref_pose); eng.set_projection_residuals(z, op, ol)
print("set_* %.2f s" % (time.time() - t))
t = time.time(); eng.finalize(); print("finalize %.2f s" % (time.time() - t))
t = time.time(); eng.begin_solve(); eng.set_pose_masks(np.zeros(sc.num_poses, dtype=np.uint16)); print("begin_solve %.2f s" % (time.time() - t))
t = time.time(); eng.linearize(); print("first linearize %.2f s" % (time.time() - t))
t = time.time(); eng.solve_gn(); print("first solve (incl. symbolic pattern) %.2f s" % (time.time() - t))
t = time.time(); eng.solve_gn(); print("second solve %.2f s" % (time.time() - t))
