import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from ba_amd import adjuster, scene
for P, L in ((30, 1500), (100, 5000)):
    sc = scene.make_scene(P, L, 8, lm_dim=1, seed=5)
    scene.add_inertial(sc, period=60.0 * P / 100.0)
    h = adjuster.BundleAdjuster(1, 15)
    o = adjuster.default_options(); o.error_change_threshold = 0; o.param_change_threshold = 0
    h.Init(o)
    scene.populate(h, sc, imu=True)
    h.Solve(1)
    t = time.perf_counter()
    for _ in range(10):
        h.Solve(1)
    dt = (time.perf_counter() - t) / 10
    print(P, L, "ms per Solve(1): %.3f" % (1e3 * dt), h.timers())
    ks = h.engine().kernel_stats() if hasattr(h.engine(), "kernel_stats") else None
