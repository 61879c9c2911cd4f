"""Host wall time of every C-ABI phase call against its device time on a small visual-inertial window
(scratch): where the gap between Solve(1) and the sum of the device phases goes."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ba_amd import adjuster, scene
P, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30, 1500)
sc = scene.make_scene(P, L, 8, lm_dim=1, seed=5)
scene.add_inertial(sc, period=60.0 * P / 100.0)
h = adjuster.BundleAdjuster(1, 15)
o = adjuster.default_options(); o.error_change_threshold = 0; o.param_change_threshold = 0
h.Init(o)
scene.populate(h, sc, imu=True)
h.Solve(2)
e = h.engine()
acc = {}
def timed(name, fn, *a):
    t = time.perf_counter(); r = fn(*a); dt = time.perf_counter() - t
    acc.setdefault(name, []).append(dt)
    return r
N = 20
masks = np.zeros(P, dtype=np.uint16); masks[0] = 0x3f
t_all = time.perf_counter()
for it in range(N):
    timed("begin_solve", e.begin_solve)
    timed("set_pose_masks", e.set_pose_masks, masks)
    timed("linearize", e.linearize)
    timed("dogleg_terms(0)", e.dogleg_terms, 0)
    timed("solve_gn", e.solve_gn)
    timed("dogleg_terms(1)", e.dogleg_terms, 1)
    timed("compose_step", e.compose_step, 0.0, 1.0)
    timed("eval(pre)", e.eval_residuals)
    timed("apply_step", e.apply_step)
    timed("eval(post)", e.eval_residuals)
    timed("rollback", e.rollback)
    timed("end_solve", e.end_solve)
t_all = (time.perf_counter() - t_all) / N
tm = e.get_timers()
print("%d poses: %.3f ms per emulated iteration; device timers of the last one: %s" % (P, 1e3 * t_all, {k: round(v, 3) for k, v in tm.items()}))
for k, v in acc.items():
    print("  %-18s wall %.1f us" % (k, 1e6 * np.median(v)))
t = time.perf_counter()
for _ in range(N):
    h.Solve(1)
print("Solve(1): %.3f ms" % (1e3 * (time.perf_counter() - t) / N))
