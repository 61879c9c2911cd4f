"""BASELINE.json configs[4] at full size on ONE MI355X (scratch validation run): 10k poses / 1M
landmarks / 10M reprojection residuals + IMU pre-integration + unary priors on every 100th pose +
binary odometry constraints, PoseSize = 15, dogleg.  The reduced system has n = 150 000: S is
180 GB of the 288 GB of HBM."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ba_amd import adjuster, scene

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
t0 = time.time()
sc = scene.make_scene(P, L, 10, lm_dim=1, seed=3)
scene.add_inertial(sc, period=60.0 * P / 100.0)
print("scene %.1f s" % (time.time() - t0), flush=True)
h = adjuster.BundleAdjuster(1, 15)
o = adjuster.default_options()
o.use_dogleg = 1
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
rng = np.random.default_rng(4)
for i in range(0, P, 100):
    h.AddUnaryConstraint(i, sc.gt_poses[i], np.diag([1e-2] * 3 + [1e-3] * 3), True)
def rel(a, b):  # T_ab = T_wa^-1 T_wb as [t, q]
    ra = scene.quat_to_rot(a[3:7]); rb = scene.quat_to_rot(b[3:7])
    r = ra.T @ rb; t = ra.T @ (b[:3] - a[:3])
    return np.concatenate([t, scene.rot_to_quat(r)]) if hasattr(scene, "rot_to_quat") else None
if hasattr(scene, "rot_to_quat"):
    for i in range(P - 1):
        t12 = rel(sc.gt_poses[i], sc.gt_poses[i + 1])
        t12[:3] += 0.01 * rng.normal(size=3)
        h.AddBinaryConstraint(i, i + 1, t12)
print("filled %.1f s" % (time.time() - t0), flush=True)
for it in range(2):
    t1 = time.time()
    h.Solve(1)
    s = h.summary()
    print("iter %d: %.3f s result %d proj %.6e inertial %.6e delta_norm %.3e" %
          (it, time.time() - t1, s.result, s.proj_error, s.inertial_error, s.delta_norm), flush=True)
