"""Randomised parity sweep of the self-calibration instantiations (scratch soak, not part of the suite):
small scenes with random sizes, calibration kind (T_vs | pinhole parameters | the five parameters of a FOV
camera), PoseSize 6 | 15 (with IMU),
dogleg on/off, random fixed poses / inactive landmarks, random wrong initial calibration; engine vs
oracle over 3 iterations: result codes, errors, the camera, poses, landmarks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ba_amd import adjuster, scene
from oracle import pyoracle as po
from helpers import gn_options, rel_err
po.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for trial in range(N):
    P = int(rng.integers(12, 60)); L = int(rng.integers(30, 250)); K = int(rng.integers(3, 8))
    kind = str(rng.choice(["tvs", "intrinsics", "fov"])); pose_dim = int(rng.choice([6, 6, 15])); dog = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 10000))
    try:
        sc = scene.make_scene(P, L, K, lm_dim=1, seed=seed, roll_amp=0.0 if pose_dim == 15 else 0.6)
    except RuntimeError:
        print("trial %2d skipped (scene generator)" % trial); continue
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    mount = ident if pose_dim == 15 else np.concatenate([rng.normal(0, 0.05, 3), po.so3_exp(rng.normal(0, 0.05, 3))])
    if pose_dim == 15:
        scene.add_inertial(sc, period=60.0 * P / 100.0)
    else:
        sc = scene.mount_camera(sc, mount)
    pa = np.ones(P, dtype=np.uint8); la = np.ones(L, dtype=np.uint8)
    fixed = rng.choice(P, max(2, P // 3), replace=False)
    pa[fixed] = 0
    sc.poses[fixed] = sc.gt_poses[fixed]
    if rng.random() < 0.5:
        la[rng.choice(L, max(1, L // 10), replace=False)] = 0
    t0, cam0 = mount, np.asarray(sc.cam_params, dtype=np.float64)
    if kind == "tvs":
        t0 = po.exp_decoupled(mount, rng.normal(0, 0.02, 6))
        sc.landmarks = scene.remount_landmarks(sc, mount, t0)
    else:
        if kind == "fov":
            scene.to_fov_camera(sc, float(rng.uniform(0.3, 1.2)))
        cam0 = np.asarray(sc.cam_params, dtype=np.float64)
        cam0 = cam0 * (1.0 + rng.normal(0, 0.02, len(cam0)))
    kw = dict(do_tvs=True) if kind == "tvs" else dict(calib_size=len(cam0))
    objs = []
    for cls in (po.OracleBundleAdjuster, adjuster.BundleAdjuster):
        if cls is po.OracleBundleAdjuster:
            opts = gn_options(po, use_dogleg=dog)
        else:
            opts = adjuster.default_options(); opts.use_dogleg = dog; opts.error_change_threshold = 0; opts.param_change_threshold = 0
        b = cls(1, pose_dim, **kw); b.Init(opts)
        if pose_dim == 15:
            b.SetGravity(sc.gravity)
        b.AddCamera(cam0, t0)
        b.add_poses(sc.poses, v_w=getattr(sc, "init_vel", None), b=getattr(sc, "init_bias", None), is_active=pa,
                    time=getattr(sc, "pose_time", None))
        b.add_landmarks(sc.landmarks, sc.lm_ref_pose, is_active=la)
        b.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
        if pose_dim == 15:
            for i in range(P - 1):
                b.AddImuResidual(i, i + 1, sc.imu_meas[i])
        objs.append(b)
    o, h = objs
    ok, cond = True, 1.0
    for it in range(3):
        o.Solve(1); h.Solve(1)
        S0 = o.S()
        w0 = np.linalg.eigvalsh(np.triu(S0) + np.triu(S0, 1).T)
        cond = max(cond, w0.max() / w0.min() if w0.min() > 0 else np.inf)
        if o.summary().result != h.summary().result:
            ok = not np.isfinite(cond) or cond > 1e14
            print("   result codes differ (oracle %d, engine %d) at iteration %d, cond %.1e" % (o.summary().result, h.summary().result, it, cond))
            break
    co = o.camera_pose(0) if kind == "tvs" else o.camera_params(0)
    ch = h.camera_pose(0) if kind == "tvs" else h.camera_params(0)
    d, dl, dc = rel_err(h.poses()[0], o.poses()[0]), rel_err(h.landmarks(), o.landmarks()), rel_err(ch, co)
    tol = max(1e-8, 1e-15 * cond)
    good = ok and max(d, dl, dc) < tol
    flag = "" if good else ("  (singular S: not comparable)" if cond > 1e14 else "  <-- CHECK")
    bad += 0 if (good or cond > 1e14) else 1
    print("trial %2d %-10s P %2d L %3d K %d D %2d dogleg %d  cond %.1e  poses %.1e landmarks %.1e camera %.1e  result %d%s"
          % (trial, kind, P, L, K, pose_dim, dog, cond, d, dl, dc, h.summary().result, flag))
print("mismatches on well-posed trials: %d of %d" % (bad, N))
