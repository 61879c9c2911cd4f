"""Iteration latency of small visual-inertial windows through ba::BundleAdjuster::Solve(1) (scratch):
wall time per call against the sum of the device phases — the gap is host-side launch cost."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ba_amd import adjuster, scene
for P, L, priors in ((30, 1500, False), (100, 5000, False), (30, 1500, True), (100, 5000, True)):
    sc = scene.make_scene(P, L, 8, lm_dim=1, seed=5)
    scene.add_inertial(sc, period=60.0 * P / 100.0)
    h = adjuster.BundleAdjuster(1, 15)
    o = adjuster.default_options(); o.error_change_threshold = 0; o.param_change_threshold = 0
    h.Init(o)
    scene.populate(h, sc, imu=True, priors=priors, unary_every=10)
    h.Solve(1)
    t = time.perf_counter()
    for _ in range(10):
        h.Solve(1)
    dt = (time.perf_counter() - t) / 10
    tm = h.timers()
    print("%3d poses %5d landmarks priors=%d: %.3f ms per Solve(1); device phases %.3f ms (jtj_schur %.3f, solve %.3f, evaluate %.3f)"
          % (P, L, priors, 1e3 * dt, sum(tm.values()), tm["jtj_schur"], tm["solve"], tm["evaluate_residuals"]))
