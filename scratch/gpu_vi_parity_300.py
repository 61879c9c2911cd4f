"""VI parity at moderate size (scratch): 300 poses / 3000 landmarks + IMU, PoseSize 15,
oracle vs engine over 3 Gauss-Newton iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ba_amd import adjuster, scene
from oracle import pyoracle as po
from helpers import fill, gn_options, rel_err
po.build()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sc = scene.make_scene(P, 10 * P, 6, lm_dim=1, seed=51)
scene.add_inertial(sc, period=60.0 * P / 100.0)
objs = []
for cls, opts in ((po.OracleBundleAdjuster, gn_options(po, use_dogleg=int(os.environ.get("DOGLEG", "0")))), (adjuster.BundleAdjuster, None)):
    b = cls(1, 15)
    if opts is None:
        opts = adjuster.default_options(); opts.use_dogleg = int(os.environ.get("DOGLEG", "0")); opts.error_change_threshold = 0; opts.param_change_threshold = 0
    b.Init(opts)
    b.SetGravity(sc.gravity)
    fill(b, sc)
    for i in range(P - 1):
        b.AddImuResidual(i, i + 1, sc.imu_meas[i])
    objs.append(b)
o, h = objs
for it in range(4):
    t = time.time(); o.Solve(1); to = time.time() - t
    h.Solve(1)
    so, sh = o.summary(), h.summary()
    po_, ph = o.poses()[0], h.poses()[0]
    print("it %d oracle: res %d proj %.6e inert %.6e dn %.3e | engine: res %d proj %.6e inert %.6e dn %.3e | pose rel diff %.2e (oracle %.1fs)" %
          (it, so.result, so.proj_error, so.inertial_error, so.delta_norm, sh.result, sh.proj_error, sh.inertial_error, sh.delta_norm, rel_err(ph, po_), to), flush=True)
