import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from ba_amd import hipapi, scene
from oracle import pyoracle as po
from helpers import *
def rel(a, b): return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)
P, L, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lm_dim = int(sys.argv[4]) if len(sys.argv) > 4 else 1
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 4
sc = scene.make_scene(P, L, K, lm_dim=lm_dim, seed=2)
pa = np.ones(P, dtype=np.uint8); pa[sc.anchor_poses] = 0
ba = po.OracleBundleAdjuster(lm_dim, 6); ba.Init(gn_options(po))
ids = fill(ba, sc, active=pa); keep = ids != 0xffffffff
eng = hipapi.Engine(lm_dim, 6)
eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1]); eng.set_poses(sc.poses, is_active=pa)
eng.set_landmarks(sc.landmarks, sc.lm_ref_pose); eng.set_projection_residuals(sc.obs_z[keep], sc.obs_pose[keep], sc.obs_lm[keep])
eng.finalize(); eng.begin_solve(); eng.set_pose_masks(np.zeros(P, dtype=np.uint16))
for it in range(iters):
    t = time.time(); ba.Solve(1); to = time.time() - t
    e0 = eng.linearize()
    S = eng.get_S(); rhs, rhs_p, rhs_l = eng.get_rhs()
    So = ba.S()
    rc = eng.solve_gn(); dp_, dl_ = eng.get_delta_gn()
    nrm = eng.compose_step(0.0, 1.0); pre = eng.eval_residuals(); eng.apply_step(); post = eng.eval_residuals()
    rolled = post.proj_error > pre.proj_error
    if rolled: eng.rollback()
    s = ba.summary()
    ev = np.linalg.eigvalsh(np.triu(So) + np.triu(So, 1).T)
    print('it', it, 'S', rel(S, So), 'rhs', rel(rhs, ba.rhs()), 'dp', rel(dp_, ba.delta_p()), 'dl', rel(dl_, ba.delta_l()), 'rc', rc, 'oracle res', s.result,
          'err hip %.6e->%.6e rolled=%s oracle proj_err %.6e dn %.4f/%.4f' % (pre.proj_error, post.proj_error, rolled, s.proj_error, nrm.step_p_norm + nrm.step_l_norm, s.delta_norm),
          'cond %.2e mineig %.3e' % (ev[-1] / ev[0], ev[0]), 'oracle %.2fs' % to, flush=True)
t, _, _ = eng.get_poses(P); ot, _, _ = ba.poses()
print('poses hip vs oracle', rel(t, ot), 'gt err', np.abs(t[:, :3] - sc.gt_poses[:, :3]).max())
