"""BASELINE.json configs[2] at full size on one MI355X (scratch validation run):
5k poses / 500k landmarks / 5M reprojection residuals + IMU pre-integration residuals,
PoseSize = 15 (velocities + biases in the state), Gauss-Newton."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ba_amd import adjuster, scene

P = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
t0 = time.time()
sc = scene.make_scene(P, L, 10, lm_dim=1, seed=3)
scene.add_inertial(sc, period=60.0 * P / 100.0)
print("scene %.1f s" % (time.time() - t0), flush=True)
h = adjuster.BundleAdjuster(1, 15)
o = adjuster.default_options()
o.use_dogleg = 0
o.error_change_threshold = 0
o.param_change_threshold = 0
h.Init(o)
h.SetGravity(sc.gravity)
h.AddCamera(sc.cam_params)
h.add_poses(sc.poses, v_w=sc.init_vel, b=sc.init_bias, time=sc.pose_time)
h.add_landmarks(sc.landmarks, sc.lm_ref_pose)
h.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
for i in range(P - 1):
    h.AddImuResidual(i, i + 1, sc.imu_meas[i])
print("filled %.1f s" % (time.time() - t0), flush=True)
for it in range(3):
    t1 = time.time()
    h.Solve(1)
    s = h.summary()
    print("iter %d: %.3f s result %d proj %.6e inertial %.6e delta_norm %.3e" %
          (it, time.time() - t1, s.result, s.proj_error, s.inertial_error, s.delta_norm), flush=True)
