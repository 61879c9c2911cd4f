"""Steady-state Solve(1) of a small visual-inertial window with ONE variant of k_imu's step pass per process
(scratch): argv[1] = value of ba_hip_debug_set key 6."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ba_amd import adjuster, scene
v = int(sys.argv[1])
for P, L in ((30, 1500), (100, 5000)):
    sc = scene.make_scene(P, L, 8, lm_dim=1, seed=5)
    scene.add_inertial(sc, period=60.0 * P / 100.0)
    h = adjuster.BundleAdjuster(1, 15)
    o = adjuster.default_options(); o.error_change_threshold = 0; o.param_change_threshold = 0
    h.Init(o)
    scene.populate(h, sc, imu=True)
    h.Solve(0)
    h.engine().debug_set(6, v)
    h.Solve(3)
    ts = []
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(20):
            h.Solve(1)
        ts.append((time.perf_counter() - t) / 20)
    print("variant %d, %3d poses: Solve(1) %s ms, jtj_schur %.3f ms" % (v, P, " ".join("%.3f" % (1e3 * x) for x in ts), h.timers()["jtj_schur"]))
