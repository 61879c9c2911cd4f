import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ba_amd import hipapi, scene
P, L, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lm_dim = int(sys.argv[4]) if len(sys.argv) > 4 else 1
t = time.time(); sc = scene.make_scene(P, L, K, lm_dim=lm_dim, seed=2); print('scene %.1fs' % (time.time() - t), flush=True)
nsel = K + (1 if lm_dim == 1 else 0)
keep = np.ones(len(sc.obs_pose), dtype=bool)
if lm_dim == 1: keep[::nsel] = False
pa = np.ones(P, dtype=np.uint8); pa[sc.anchor_poses] = 0
eng = hipapi.Engine(lm_dim, 6)
eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1])
eng.set_poses(sc.poses, is_active=pa)
eng.set_landmarks(sc.landmarks, sc.lm_ref_pose)
eng.set_projection_residuals(sc.obs_z[keep], sc.obs_pose[keep], sc.obs_lm[keep])
t = time.time(); eng.finalize(); print('finalize %.2fs' % (time.time() - t), flush=True)
eng.begin_solve(); eng.set_pose_masks(np.zeros(P, dtype=np.uint16))
for it in range(6):
    t0 = time.time()
    e0 = eng.linearize(); t1 = time.time()
    rc = eng.solve_gn(); t2 = time.time()
    nrm = eng.compose_step(0.0, 1.0)
    pre = eng.eval_residuals(); eng.apply_step(); post = eng.eval_residuals(); t3 = time.time()
    if post.proj_error > pre.proj_error: eng.rollback()
    print('it %d rc %d err %.1f -> %.1f step %.4f | wall lin %.2fms solve %.2fms rest %.2fms total %.2fms' % (it, rc, pre.proj_error, post.proj_error, nrm.step_p_norm + nrm.step_l_norm, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3), flush=True)
    print('   ', {k: round(v, 3) for k, v in eng.get_timers().items()}, flush=True)
eng.end_solve()
t, _, _ = eng.get_poses(P)
print('pose err', np.abs(t[:, :3] - sc.gt_poses[:, :3]).max(), 'init', np.abs(sc.poses[:, :3] - sc.gt_poses[:, :3]).max())
