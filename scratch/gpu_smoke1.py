import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from ba_amd import hipapi, scene
from oracle import pyoracle as po
from helpers import *

def rel(a, b): return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)

eng = hipapi.Engine(1, 6)
# 1. select
rng = np.random.default_rng(0)
for n in (1, 7, 1000, 100001):
    v = rng.random(n) ** 3 * 50
    k = n // 2
    got = eng.select_kth(v, k)
    print('select n=%d ok=%s' % (n, got == np.sort(v)[k]))
# 2. dense solve
for n in (5, 64, 100, 300, 1000):
    M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n); b = rng.normal(size=n)
    t = time.time(); x, rc = eng.dense_solve(np.tril(A), b); dt = time.time() - t
    print('dense n=%d rc=%d relerr=%.2e  (%.3fs)' % (n, rc, rel(x, np.linalg.solve(A, b)), dt))
eng.close()

# 3. linearize parity
for lm_dim in (1, 3):
    for variant in ('two_fixed', 'all_active', 'inactive_mix'):
        sc = scene.make_scene(30, 60, 5, lm_dim=lm_dim, seed=7)
        P = sc.num_poses
        pa = np.ones(P, dtype=np.uint8); la = np.ones(sc.num_landmarks, dtype=np.uint8)
        if variant == 'two_fixed': pa[sc.anchor_poses] = 0
        if variant == 'inactive_mix': pa[[0, 3, 4, 17]] = 0; la[[5, 6, 40]] = 0
        ba = po.OracleBundleAdjuster(lm_dim, 6)
        ba.Init(gn_options(po))
        ids = fill(ba, sc, active=pa, lm_active=la)
        ba.Solve(1)
        keep = ids != 0xffffffff
        eng = hipapi.Engine(lm_dim, 6)
        eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1])
        eng.set_poses(sc.poses, is_active=pa)
        eng.set_landmarks(sc.landmarks, sc.lm_ref_pose, is_active=la)
        eng.set_projection_residuals(sc.obs_z[keep], sc.obs_pose[keep], sc.obs_lm[keep])
        eng.finalize()
        masks = np.zeros(P, dtype=np.uint16)
        if variant == 'all_active': masks[0] = 0x3f
        eng.begin_solve()
        eng.set_pose_masks(masks)
        e0 = eng.linearize()
        S = eng.get_S(); rhs, rhs_p, rhs_l = eng.get_rhs()
        print('lm_dim', lm_dim, variant, 'S', rel(S, ba.S()), 'rhs', rel(rhs, ba.rhs()), 'rhs_p', rel(rhs_p, ba.rhs_p()), 'rhs_l', rel(rhs_l, ba.rhs_l()),
              'w', rel(eng.get_proj_weights(keep.sum()), ba.proj_weights()))
        rc = eng.solve_gn()
        dp_, dl_ = eng.get_delta_gn()
        print('   solve rc', rc, 'delta_p', rel(dp_, ba.delta_p()), 'delta_l', rel(dl_, ba.delta_l()))
        nrm = eng.compose_step(0.0, 1.0)
        pre = eng.eval_residuals()
        eng.apply_step()
        post = eng.eval_residuals()
        eng.end_solve()
        t, _, _ = eng.get_poses(P); x = eng.get_landmarks(sc.num_landmarks)
        ot, _, _ = ba.poses(); ox = ba.landmarks()
        s = ba.summary()
        print('   norms', nrm.step_p_norm + nrm.step_l_norm, s.delta_norm, 'post err', post.proj_error, s.proj_error, 'pre', pre.proj_error, e0.proj_error,
              'poses', rel(t, ot), 'lms', rel(x, ox))
        print('   timers', eng.get_timers())
        eng.close()
