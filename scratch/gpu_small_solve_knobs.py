"""solve phase of small windows under the schedule knobs (scratch): run with BA_HIP_NO_LOOKAHEAD / BA_HIP_KOUT set."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ba_amd import adjuster, scene
for P, L, D in ((100, 5000, 15), (200, 20000, 6), (50, 2000, 6)):
    sc = scene.make_scene(P, L, 8, lm_dim=1, seed=5)
    if D == 15:
        scene.add_inertial(sc, period=60.0 * P / 100.0)
    h = adjuster.BundleAdjuster(1, D)
    o = adjuster.default_options(); o.error_change_threshold = 0; o.param_change_threshold = 0
    h.Init(o)
    pa = None
    if D == 6:
        pa = np.ones(P, dtype=np.uint8); pa[sc.anchor_poses] = 0
    scene.populate(h, sc, imu=D == 15, active=pa)
    h.Solve(2)
    ts = []
    for _ in range(5):
        h.Solve(1); ts.append(h.timers()["solve"])
    print("%s KOUT=%s nolook=%s  P %3d D %2d: solve %.3f ms" % ("knobs", os.environ.get("BA_HIP_KOUT"), os.environ.get("BA_HIP_NO_LOOKAHEAD"), P, D, np.median(ts)))
