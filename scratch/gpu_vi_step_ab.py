"""A/B of the step pass of k_imu on small visual-inertial windows in ONE process (scratch): ba_hip_debug_set key 6 =
1 (a wavefront per residual over the lane-per-sample step pass) against 4 (a wavefront per sample too)."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ba_amd import adjuster, scene
for P, L in ((30, 1500), (100, 5000)):
    sc = scene.make_scene(P, L, 8, lm_dim=1, seed=5)
    scene.add_inertial(sc, period=60.0 * P / 100.0)
    h = adjuster.BundleAdjuster(1, 15)
    o = adjuster.default_options(); o.error_change_threshold = 0; o.param_change_threshold = 0
    h.Init(o)
    scene.populate(h, sc, imu=True)
    h.Solve(2)
    for rep in range(3):
        for v in (tuple(int(x) for x in sys.argv[1:]) or (1, 4)):
            h.engine().debug_set(6, v)
            h.Solve(1)
            t = time.perf_counter()
            for _ in range(20):
                h.Solve(1)
            dt = (time.perf_counter() - t) / 20
            print("%3d poses variant %d: %.3f ms per Solve(1), jtj_schur %.3f ms" % (P, v, 1e3 * dt, h.timers()["jtj_schur"]))
