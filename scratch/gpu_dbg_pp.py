import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from ba_amd import adjuster, scene
from oracle import pyoracle as po
from helpers import *
from test_gpu_parity import hip_options
rng = np.random.default_rng(41)
gt, _ = scene.trajectory(30)
init = gt.copy(); init[:, :3] += rng.normal(0, 0.05, (30, 3))
for i in range(30): init[i] = po.exp_decoupled(init[i], np.concatenate([np.zeros(3), rng.normal(0, 0.02, 3)]))
objs = []
for cls, opts in ((po.OracleBundleAdjuster, gn_options(po, use_dogleg=1)), (adjuster.BundleAdjuster, hip_options(use_dogleg=1))):
    b = cls(0, 6); b.Init(opts); b.add_poses(init)
    r2 = np.random.default_rng(42)
    for i in range(0, 30, 3):
        m = r2.normal(size=(6, 6)); cov = 1e-3 * (m @ m.T + 6 * np.eye(6))
        prior = po.exp_decoupled(gt[i], r2.normal(0, 0.01, 6)); b.AddUnaryConstraint(i, prior, cov, bool(i % 2 == 0))
    for i in range(29):
        m = r2.normal(size=(6, 6)); cov = 1e-4 * (m @ m.T + 6 * np.eye(6))
        t12 = po.exp_decoupled(po.se3_mul(po.se3_inv(gt[i]), gt[i + 1]), r2.normal(0, 0.003, 6))
        b.AddBinaryConstraint(i, i + 1, t12, cov, float(r2.uniform(0.5, 2.0)), bool(i % 5 != 0))
    b.AddBinaryConstraint(0, 29, po.se3_mul(po.se3_inv(gt[0]), gt[29]))
    objs.append(b)
o, h = objs
for it in range(3):
    o.Solve(1); h.Solve(1)
    for n_, b in (('orc', o), ('hip', h)):
        s = b.summary()
        print(it, n_, 'res', s.result, 'un', s.unary_error, 'bin', s.binary_error, 'pre', s.pre_solve_norm, 'post', s.post_solve_norm, 'dn', s.delta_norm, 'tr', s.trust_region_size)
    print('   S', rel_err(h.S(), o.S()), 'rhs', rel_err(h.rhs(), o.rhs()), 'dp', rel_err(h.delta_p(), o.delta_p()))
