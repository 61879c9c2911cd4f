"""scratch: why does the oracle report FactorizationError where the engine does not (seed 4, trial 52)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ba_amd import adjuster, scene
from oracle import pyoracle as po
from helpers import fill, gn_options
po.build()
rng = np.random.default_rng(4)
for trial in range(53):
    P = int(rng.integers(6, 60)); L = int(rng.integers(10, 200)); K = int(rng.integers(3, min(P - 1, 8)))
    lm_dim = int(rng.choice([1, 3])); pose_dim = int(rng.choice([6, 6, 15])); dog = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 10000))
    try:
        sc = scene.make_scene(P, L, K, lm_dim=lm_dim, seed=seed)
    except RuntimeError:
        continue
    pa = np.ones(P, dtype=np.uint8); la = np.ones(L, dtype=np.uint8)
    if pose_dim == 6: pa[sc.anchor_poses] = 0
    if rng.random() < 0.5: pa[rng.choice(P, max(1, P // 10), replace=False)] = 0
    if rng.random() < 0.5: la[rng.choice(L, max(1, L // 10), replace=False)] = 0
    if trial != 52:
        rng.random(); rng.random(); continue
    print(P, L, K, lm_dim, pose_dim, dog)
    objs = []
    for cls in (po.OracleBundleAdjuster, adjuster.BundleAdjuster):
        if cls is po.OracleBundleAdjuster: opts = gn_options(po, use_dogleg=dog)
        else:
            opts = adjuster.default_options(); opts.use_dogleg = dog; opts.error_change_threshold = 0; opts.param_change_threshold = 0
            opts.write_reduced_camera_matrix = 1
        b = cls(lm_dim, pose_dim); b.Init(opts); fill(b, sc, active=pa, lm_active=la); objs.append(b)
    r2 = np.random.default_rng(seed)
    for i in range(0, P - 1, 3):
        if r2.random() < 0.3:
            for b in objs: b.AddUnaryConstraint(i, sc.gt_poses[i], 1e-2 * np.eye(6), bool(i % 2))
    o, h = objs
    o.Solve(1); h.Solve(1)
    So, Sh = o.S(), h.S()
    do, dh = np.diag(So), np.diag(Sh)
    print("oracle result", o.summary().result, "engine", h.summary().result)
    print("oracle zero diag idx", np.where(do == 0)[0][:20], "engine zero diag", np.where(dh == 0)[0][:20])
    print("max |S diff|", np.abs(np.tril(Sh) - np.tril(So.T)).max() if So.shape == Sh.shape else (So.shape, Sh.shape))
    w = np.linalg.eigvalsh(np.triu(So) + np.triu(So, 1).T)
    print("eig min/max", w.min(), w.max(), "num tiny", (np.abs(w) < 1e-9 * w.max()).sum())
    break
