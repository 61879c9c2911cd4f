import sys
sys.path.insert(0, '/root/repo')
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ba_amd import adjuster, scene
P, L = 30, 1500
sc = scene.make_scene(P, L, 8, lm_dim=1, seed=5)
scene.add_inertial(sc, period=60.0 * P / 100.0)
h = adjuster.BundleAdjuster(1, 15)
o = adjuster.default_options(); o.error_change_threshold = 0; o.param_change_threshold = 0
h.Init(o)
scene.populate(h, sc, imu=True)
h.Solve(1)
