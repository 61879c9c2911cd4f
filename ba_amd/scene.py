"""Deterministic synthetic bundle-adjustment scenes.

Numpy only.  Used by tests/, bench.py and __graft_entry__.smoke() to feed the same
seeded problem to the HIP path and to the oracle.

Scene parameters follow the reference's research simulator
(/root/reference/matlab/simulate_vins.py:81-92,116-125,155-156,199): pinhole camera
fx=198.969 fy=198.1284 u0=329.9368 v0=240.1017 at 640x480, a closed "curvy square"
trajectory at height -1.5 m, pixel noise sigma 1.5, and the IMU noise constants of
/root/reference/include/ba/Types.h:33-36.

Poses are 7-vectors [tx,ty,tz,qx,qy,qz,qw]; the camera looks along +z of the pose
frame (T_vs = identity), x right, y down; world z points down.
"""
import numpy as np

CAM_PARAMS = np.array([198.969, 198.1284, 329.9368, 240.1017])
IMG_W, IMG_H = 640.0, 480.0
GRAVITY = np.array([0.0, 0.0, 9.8007])  # world z is down (reference: ba::Gravity, Types.h:39)


def rot_to_quat(R):
    """Rotation matrices (...,3,3) -> quaternions (...,4) in (x,y,z,w) order."""
    R = np.asarray(R)
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    q = np.empty(R.shape[:-2] + (4,))
    tr = m00 + m11 + m22
    # four candidate branches, pick the numerically best per element
    w = np.sqrt(np.maximum(0.0, 1 + tr)) / 2
    x = np.sqrt(np.maximum(0.0, 1 + m00 - m11 - m22)) / 2
    y = np.sqrt(np.maximum(0.0, 1 - m00 + m11 - m22)) / 2
    z = np.sqrt(np.maximum(0.0, 1 - m00 - m11 + m22)) / 2
    best = np.argmax(np.stack([w, x, y, z], -1), -1)
    for k in range(4):
        sel = best == k
        if not np.any(sel):
            continue
        Rs = R[sel]
        if k == 0:
            ww = w[sel]
            q[sel] = np.stack([(Rs[:, 2, 1] - Rs[:, 1, 2]) / (4 * ww),
                               (Rs[:, 0, 2] - Rs[:, 2, 0]) / (4 * ww),
                               (Rs[:, 1, 0] - Rs[:, 0, 1]) / (4 * ww), ww], -1)
        elif k == 1:
            xx = x[sel]
            q[sel] = np.stack([xx, (Rs[:, 0, 1] + Rs[:, 1, 0]) / (4 * xx),
                               (Rs[:, 0, 2] + Rs[:, 2, 0]) / (4 * xx),
                               (Rs[:, 2, 1] - Rs[:, 1, 2]) / (4 * xx)], -1)
        elif k == 2:
            yy = y[sel]
            q[sel] = np.stack([(Rs[:, 0, 1] + Rs[:, 1, 0]) / (4 * yy), yy,
                               (Rs[:, 1, 2] + Rs[:, 2, 1]) / (4 * yy),
                               (Rs[:, 0, 2] - Rs[:, 2, 0]) / (4 * yy)], -1)
        else:
            zz = z[sel]
            q[sel] = np.stack([(Rs[:, 0, 2] + Rs[:, 2, 0]) / (4 * zz),
                               (Rs[:, 1, 2] + Rs[:, 2, 1]) / (4 * zz), zz,
                               (Rs[:, 1, 0] - Rs[:, 0, 1]) / (4 * zz)], -1)
    return q


def quat_to_rot(q):
    q = np.asarray(q)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - w * z)
    R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y)
    R[..., 2, 1] = 2 * (y * z + w * x)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_exp(w):
    w = np.asarray(w)
    th = np.linalg.norm(w, axis=-1, keepdims=True)
    half = 0.5 * th
    s = np.where(th < 1e-10, 0.5 - th * th / 48.0, np.sin(half) / np.where(th < 1e-10, 1.0, th))
    return np.concatenate([s * w, np.cos(half)], -1)


def trajectory(P, radius=7.5, height=-1.5, roll_amp=0.0):
    """Ground-truth poses on a closed curvy square; camera z along the direction of travel.
    roll_amp > 0 banks the frames about the direction of travel by roll_amp sin(5 s): a nearly
    planar trajectory leaves a mount's translation along the turning axis unobservable."""
    s = np.arange(P) / float(P) * 2 * np.pi
    r = radius * (1.0 + 0.15 * np.cos(4 * s))
    pos = np.stack([r * np.cos(s), r * np.sin(s), height + 0.3 * np.sin(3 * s)], -1)
    dr = -radius * 0.6 * np.sin(4 * s)
    vel = np.stack([dr * np.cos(s) - r * np.sin(s), dr * np.sin(s) + r * np.cos(s),
                    0.9 * np.cos(3 * s)], -1)
    zc = vel / np.linalg.norm(vel, axis=-1, keepdims=True)
    down = np.array([0.0, 0.0, 1.0])
    xc = np.cross(np.broadcast_to(down, zc.shape), zc)
    xc /= np.linalg.norm(xc, axis=-1, keepdims=True)
    yc = np.cross(zc, xc)
    if roll_amp:
        a = roll_amp * np.sin(5 * s)
        xc, yc = (np.cos(a)[:, None] * xc + np.sin(a)[:, None] * yc,
                  -np.sin(a)[:, None] * xc + np.cos(a)[:, None] * yc)
    R = np.stack([xc, yc, zc], -1)  # columns = camera axes in the world
    return np.concatenate([pos, rot_to_quat(R)], -1), vel


def project(poses7, pts):
    """Pinhole projection of world points (N,3) into poses (N,7) (T_vs = I). -> (uv, depth)."""
    R = quat_to_rot(poses7[..., 3:7])
    pc = np.einsum('...ji,...j->...i', R, pts - poses7[..., :3])
    z = pc[..., 2]
    zs = np.where(np.abs(z) < 1e-12, 1e-12, z)
    uv = np.stack([CAM_PARAMS[0] * pc[..., 0] / zs + CAM_PARAMS[2],
                   CAM_PARAMS[1] * pc[..., 1] / zs + CAM_PARAMS[3]], -1)
    return uv, z


class Scene:
    """Container of one synthetic problem (ground truth + perturbed initial state)."""
    pass


def make_scene(num_poses, num_landmarks, obs_per_landmark=10, lm_dim=1, seed=0,
               pixel_sigma=1.5, outlier_frac=0.02, trans_sigma=0.05, rot_sigma=0.01,
               depth_sigma=0.05, lm_range=None, window=None, chunk=200000, anchors=None, roll_amp=0.0):
    """Build a scene with exactly `obs_per_landmark` ACCEPTED projection residuals per landmark.

    lm_dim == 1 (inverse depth): the first chosen pose is the landmark's reference pose; its
    own observation only sets z_ref and is rejected by AddProjectionResidual
    (/root/reference/include/ba/BundleAdjuster.h:489-501), so k+1 poses are chosen.
    lm_dim == 3: k poses, the first is still recorded as the (unused) reference pose.

    `lm_range` restricts generation to landmark ids [lo, hi) of the same global scene —
    every landmark is drawn from its own counter-based stream, so shards of one scene can
    be generated independently (multi-GPU landmark sharding).
    """
    P, Ltot, k = int(num_poses), int(num_landmarks), int(obs_per_landmark)
    lo, hi = (0, Ltot) if lm_range is None else lm_range
    L = hi - lo
    nsel = k + 1 if lm_dim == 1 else k
    gt_poses, vel = trajectory(P, roll_amp=roll_amp)
    if window is None:
        window = max(nsel + 2, min(P // 2 - 1, max(24, P // 8)))
    ncand = min(2 * window, max(4 * nsel, 48))

    sc = Scene()
    sc.lm_dim, sc.num_poses, sc.num_landmarks_total, sc.lm_lo = lm_dim, P, Ltot, lo
    sc.cam_params = CAM_PARAMS.copy()
    sc.gt_poses = gt_poses
    sc.gt_vel = vel

    # perturbed initial poses; the anchor poses (default: pose 0 and the diametrically
    # opposite pose P/2) are left at ground truth so that holding them inactive fixes the
    # 7-dof gauge (scale included) with a long baseline and without bias
    rng_p = np.random.Generator(np.random.PCG64([seed, 0xBA5E, 1]))
    dt = rng_p.normal(0.0, trans_sigma, (P, 3))
    dw = rng_p.normal(0.0, rot_sigma, (P, 3))
    anchors = (0, P // 2) if anchors is None else tuple(anchors)
    sc.anchor_poses = np.array(anchors, dtype=np.int64)
    dt[sc.anchor_poses] = 0
    dw[sc.anchor_poses] = 0
    init = gt_poses.copy()
    init[:, :3] += dt
    init[:, 3:7] = quat_mul(gt_poses[:, 3:7], quat_exp(dw))
    init[:, 3:7] /= np.linalg.norm(init[:, 3:7], axis=-1, keepdims=True)
    sc.poses = init

    x_w = np.empty((L, 4))
    ref_pose = np.empty(L, dtype=np.uint32)
    sel_poses = np.empty((L, nsel), dtype=np.uint32)
    z_all = np.empty((L, nsel, 2))
    for c0 in range(0, L, chunk):
        c1 = min(L, c0 + chunk)
        n = c1 - c0
        # one stream per chunk start (chunks are aligned to the global landmark index)
        todo = np.arange(n)
        attempt = 0
        while todo.size:
            rng = np.random.Generator(np.random.PCG64([seed, 0xBA5E, 2, lo + c0, attempt]))
            m = todo.size
            anchor = rng.integers(0, P, m)
            uv = np.stack([rng.uniform(20, IMG_W - 20, m), rng.uniform(20, IMG_H - 20, m)], -1)
            depth = rng.uniform(2.0, 40.0, m)
            ap = gt_poses[anchor]
            ray = np.stack([(uv[:, 0] - CAM_PARAMS[2]) / CAM_PARAMS[0],
                            (uv[:, 1] - CAM_PARAMS[3]) / CAM_PARAMS[1], np.ones(m)], -1)
            pts = ap[:, :3] + np.einsum('nij,nj->ni', quat_to_rot(ap[:, 3:7]), ray * depth[:, None])
            # candidate poses: distinct offsets around the anchor
            offs = np.argsort(rng.random((m, 2 * window)), axis=1)[:, :ncand] - window
            offs = np.where(offs >= 0, offs + 1, offs)  # skip 0 (the anchor itself)
            cand = (anchor[:, None] + offs) % P
            cuv, cz = project(gt_poses[cand], pts[:, None, :])
            vis = (cz > 0.5) & (cuv[..., 0] > 0) & (cuv[..., 0] < IMG_W) & \
                  (cuv[..., 1] > 0) & (cuv[..., 1] < IMG_H)
            # distinct candidates only (wrap-around may alias when 2*window >= P)
            order = np.argsort(~vis, axis=1, kind='stable')  # visible first, original order kept
            nvis = vis.sum(1)
            ok = nvis >= (nsel - 1)
            good = np.nonzero(ok)[0]
            if good.size:
                gi = todo[good]
                picks = np.take_along_axis(cand[good], order[good, :nsel - 1], 1)
                sel = np.concatenate([anchor[good, None], picks], 1)
                # reject rows with duplicate poses
                srt = np.sort(sel, 1)
                dup = (srt[:, 1:] == srt[:, :-1]).any(1)
                keep = ~dup
                gi, sel, good = gi[keep], sel[keep], good[keep]
                x_w[c0 + gi, :3] = pts[good]
                x_w[c0 + gi, 3] = 1.0
                ref_pose[c0 + gi] = sel[:, 0]
                sel_poses[c0 + gi] = sel
                zz, _ = project(gt_poses[sel], pts[good][:, None, :])
                z_all[c0 + gi] = zz
                done = np.zeros(m, dtype=bool)
                done[good] = True
                todo = todo[~done]
            attempt += 1
            if attempt > 200:
                raise RuntimeError("scene generation did not converge")
    # measurement noise + gross outliers
    rng_n = np.random.Generator(np.random.PCG64([seed, 0xBA5E, 3, lo]))
    z_all += rng_n.normal(0.0, pixel_sigma, z_all.shape)
    out = rng_n.random(z_all.shape[:2]) < outlier_frac
    out[:, 0] = False  # the reference observation defines the feature: mismatches happen in the other frames
    zo = np.stack([rng_n.uniform(0, IMG_W, z_all.shape[:2]), rng_n.uniform(0, IMG_H, z_all.shape[:2])], -1)
    z_all = np.where(out[..., None], zo, z_all)
    # landmark initialisation as a tracker does it: back-project the (noisy) reference
    # observation from the current ESTIMATE of the reference pose at a perturbed depth,
    # so the inverse-depth ray of lm_dim == 1 agrees with z_ref.
    z_ref = z_all[:, 0, :]
    ray = np.stack([(z_ref[:, 0] - CAM_PARAMS[2]) / CAM_PARAMS[0],
                    (z_ref[:, 1] - CAM_PARAMS[3]) / CAM_PARAMS[1], np.ones(L)], -1)
    gp = gt_poses[ref_pose]
    depth = np.einsum('nij,nj->ni', np.transpose(quat_to_rot(gp[:, 3:7]), (0, 2, 1)),
                      x_w[:, :3] - gp[:, :3])[:, 2]
    scale = 1.0 / (1.0 + rng_n.normal(0.0, depth_sigma, L))
    ip = init[ref_pose]
    x_init = x_w.copy()
    x_init[:, :3] = ip[:, :3] + np.einsum('nij,nj->ni', quat_to_rot(ip[:, 3:7]),
                                          ray * (depth * scale)[:, None])
    sc.gt_landmarks = x_w
    sc.landmarks = x_init
    sc.lm_ref_pose = ref_pose
    sc.num_landmarks = L
    # observation table in insertion order: for each landmark, its chosen poses in order
    sc.obs_pose = sel_poses.reshape(-1)
    sc.obs_lm = np.repeat(np.arange(L, dtype=np.uint32), nsel)
    sc.obs_z = z_all.reshape(-1, 2)
    sc.obs_per_landmark = k
    return sc


def add_scene_to(ba, sc, pose_dim=6, active=None):
    """Feed a Scene through a reference-style API object (oracle or ba_amd adjuster)."""
    ba.AddCamera(sc.cam_params)
    P = sc.num_poses
    act = np.ones(P, dtype=np.uint8) if active is None else np.asarray(active, dtype=np.uint8)
    v = getattr(sc, 'init_vel', None)
    ba.add_poses(sc.poses, v_w=v, is_active=act, time=getattr(sc, 'pose_time', None))
    ba.add_landmarks(sc.landmarks, sc.lm_ref_pose)
    return ba.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)


# ---------------------------------------------------------------------------------------
# visual-inertial extension (BASELINE.json configs[2] / configs[4])
def _traj_at(t, period, radius=7.5, height=-1.5):
    """Position and rotation matrix of the curvy-square trajectory at time(s) t."""
    s = np.asarray(t, dtype=np.float64) / period * 2 * np.pi
    r = radius * (1.0 + 0.15 * np.cos(4 * s))
    pos = np.stack([r * np.cos(s), r * np.sin(s), height + 0.3 * np.sin(3 * s)], -1)
    dr = -radius * 0.6 * np.sin(4 * s)
    vel = np.stack([dr * np.cos(s) - r * np.sin(s), dr * np.sin(s) + r * np.cos(s),
                    0.9 * np.cos(3 * s)], -1)
    zc = vel / np.linalg.norm(vel, axis=-1, keepdims=True)
    down = np.array([0.0, 0.0, 1.0])
    xc = np.cross(np.broadcast_to(down, zc.shape), zc)
    xc /= np.linalg.norm(xc, axis=-1, keepdims=True)
    yc = np.cross(zc, xc)
    return pos, np.stack([xc, yc, zc], -1)


def add_inertial(sc, period=60.0, samples_per_interval=10, seed=0, vel_sigma=0.02,
                 gyro_sigma=5.3088444e-5, accel_sigma=0.001883649, gravity=None):
    """Attach pose times, velocities, IMU samples (one residual per consecutive pose pair,
    /root/reference/matlab/simulate_vins.py:155-156) to a Scene.  Body rates and specific
    force come from central differences of the analytic trajectory; the integrator model
    is the reference's:  v' = R (a_m + b_a) - g,  q' = exp(R (w_m + b_g) dt) q
    (/root/reference/include/ba/Types.h:376-416)."""
    P = sc.num_poses
    g = GRAVITY if gravity is None else np.asarray(gravity, dtype=np.float64)
    tp = np.arange(P) * (period / P)
    h = 1e-4
    n = samples_per_interval
    # sample times: n+1 samples spanning each pose interval (shared end points)
    ts = (tp[:-1, None] + (tp[1] - tp[0]) * np.arange(n + 1)[None, :] / n)  # (P-1, n+1)
    p0, R0 = _traj_at(ts, period)
    pp, Rp = _traj_at(ts + h, period)
    pm, Rm = _traj_at(ts - h, period)
    acc = (pp - 2 * p0 + pm) / (h * h)
    dR = (Rp - Rm) / (2 * h)
    W = np.einsum('...ji,...jk->...ik', R0, dR)  # R^T dR = [w_body]x
    w_body = np.stack([W[..., 2, 1], W[..., 0, 2], W[..., 1, 0]], -1)
    a_body = np.einsum('...ji,...j->...i', R0, acc + g)
    rng = np.random.Generator(np.random.PCG64([seed, 0xBA5E, 7]))
    w_body = w_body + rng.normal(0, gyro_sigma, w_body.shape)
    a_body = a_body + rng.normal(0, accel_sigma, a_body.shape)
    sc.imu_meas = np.concatenate([w_body, a_body, ts[..., None]], -1)  # (P-1, n+1, 7)
    _, Rg = _traj_at(tp, period)
    vel = (_traj_at(tp + h, period)[0] - _traj_at(tp - h, period)[0]) / (2 * h)
    sc.gt_vel = vel
    sc.init_vel = vel + rng.normal(0, vel_sigma, vel.shape)
    sc.init_bias = np.zeros((P, 6))
    sc.pose_time = tp
    sc.gravity = g
    return sc


def fov_factor(r, w):
    """r_d / r_u of the FOV camera model (Devernay & Faugeras): atan(2 r tan(w/2)) / (r w)."""
    r = np.asarray(r, dtype=np.float64)
    m = 2.0 * np.tan(0.5 * w)
    rs = np.where(r * r < 1e-5, 1.0, r)
    return np.where(r * r < 1e-5, m / w, np.arctan(rs * m) / (rs * w))


def to_fov_camera(sc, w):
    """Turn a pinhole scene into the same scene seen through a FOV camera (fx, fy, u0, v0, w):
    every measured pixel is moved to where that camera images its ray.  The landmarks are
    back-projections of the reference pixels, so they stay where they are."""
    fx, fy, u0, v0 = sc.cam_params[:4]
    px, py = (sc.obs_z[:, 0] - u0) / fx, (sc.obs_z[:, 1] - v0) / fy
    f = fov_factor(np.hypot(px, py), w)
    sc.obs_z = np.stack([fx * f * px + u0, fy * f * py + v0], -1)
    sc.cam_params = np.array([fx, fy, u0, v0, w])
    return sc


def relative_pose(a, b):
    """T_ab = T_wa^-1 T_wb for poses [t, q(xyzw)]."""
    ra, rb = quat_to_rot(np.asarray(a)[3:7]), quat_to_rot(np.asarray(b)[3:7])
    return np.concatenate([ra.T @ (np.asarray(b)[:3] - np.asarray(a)[:3]), rot_to_quat(ra.T @ rb)])


def mount_camera(sc, t_vs):
    """Re-express a Scene for a camera mounted at T_vs on the vehicle: the scene's poses are the
    camera frames the observations were rendered from, so the vehicle poses are T_wc T_vs^-1
    (ground truth and initial guess alike).  Returns sc; sc.gt_t_vs holds the mount."""
    t_vs = np.asarray(t_vs, dtype=np.float64)
    rv = quat_to_rot(t_vs[3:])
    inv = np.concatenate([-rv.T @ t_vs[:3], t_vs[3:] * np.array([-1.0, -1.0, -1.0, 1.0])])
    for arr in (sc.poses, sc.gt_poses):
        for i in range(arr.shape[0]):
            r = quat_to_rot(arr[i, 3:7])
            arr[i, :3] = r @ inv[:3] + arr[i, :3]
            arr[i, 3:7] = quat_mul(arr[i, 3:7], inv[3:])
    sc.gt_t_vs = t_vs
    return sc


def _se3_mul(a, b):
    return np.concatenate([quat_to_rot(a[3:7]) @ b[:3] + a[:3], quat_mul(a[3:7], b[3:7])])


def _se3_apply(t, x3, w=1.0):
    return quat_to_rot(t[3:7]) @ x3 + t[:3] * w


def remount_landmarks(sc, t_vs_from, t_vs_to):
    """World points of the landmarks for a different mount guess, keeping every landmark's
    coordinates in the frame of its reference CAMERA (what a front end that unprojects z_ref with
    its current T_vs guess hands to AddLandmark).  Returns the new (L, 4) array."""
    out = sc.landmarks.copy()
    for l in range(out.shape[0]):
        ref = sc.poses[sc.lm_ref_pose[l]]
        t_ws0 = _se3_mul(ref, np.asarray(t_vs_from, dtype=np.float64))
        r0 = quat_to_rot(t_ws0[3:7])
        xs = r0.T @ (out[l, :3] - t_ws0[:3] * out[l, 3])
        out[l, :3] = _se3_apply(_se3_mul(ref, np.asarray(t_vs_to, dtype=np.float64)), xs, out[l, 3])
    return out


def populate(ba, sc, active=None, imu=False, priors=False, unary_every=100, seed=4, lm_range=None, pose_pose=True,
             lm_ids=None):
    """Feed a Scene through a reference-style API object (the oracle's or ba_amd.adjuster's
    BundleAdjuster — same method names): camera, poses, landmarks, projection residuals;
    imu: one inertial residual per consecutive pose pair (needs add_inertial);
    priors: a unary prior on every `unary_every`-th pose (cov diag(1e-2 I3, 1e-3 I3)) and a binary
    odometry constraint between consecutive poses from the ground truth + 1 cm noise, identity
    covariance — the configs[4] recipe of SURVEY.md §8d.
    lm_range = (lo, hi) or lm_ids = ascending landmark ids: only that landmark shard and its observations (multi-GPU:
    every rank holds all poses); pose_pose = False: no inertial / unary / binary residuals (they live on rank 0)."""
    if hasattr(sc, "gravity") and imu:
        ba.SetGravity(sc.gravity)
    ba.AddCamera(sc.cam_params)
    ba.add_poses(sc.poses, v_w=getattr(sc, "init_vel", None), b=getattr(sc, "init_bias", None),
                 is_active=active, time=getattr(sc, "pose_time", None))
    if lm_ids is not None:   # an arbitrary (ascending) set of landmark ids: shards dealt along the trajectory
        ids = np.asarray(lm_ids)
        new_id = np.full(sc.num_landmarks, -1, dtype=np.int64)
        new_id[ids] = np.arange(len(ids))
        sel = new_id[sc.obs_lm] >= 0
        ba.add_landmarks(sc.landmarks[ids], sc.lm_ref_pose[ids])
        n = ba.add_projection_residuals(sc.obs_z[sel], sc.obs_pose[sel], new_id[sc.obs_lm[sel]].astype(np.uint32))
    elif lm_range is None:
        ba.add_landmarks(sc.landmarks, sc.lm_ref_pose)
        n = ba.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
    else:
        lo, hi = lm_range
        sel = (sc.obs_lm >= lo) & (sc.obs_lm < hi)
        ba.add_landmarks(sc.landmarks[lo:hi], sc.lm_ref_pose[lo:hi])
        n = ba.add_projection_residuals(sc.obs_z[sel], sc.obs_pose[sel], sc.obs_lm[sel] - lo)
    P = sc.num_poses
    if not pose_pose:
        return n
    if imu:
        for i in range(P - 1):
            ba.AddImuResidual(i, i + 1, sc.imu_meas[i])
    if priors:
        rng = np.random.default_rng(seed)
        for i in range(0, P, unary_every):
            ba.AddUnaryConstraint(i, sc.gt_poses[i], np.diag([1e-2] * 3 + [1e-3] * 3), True)
        for i in range(P - 1):
            t12 = relative_pose(sc.gt_poses[i], sc.gt_poses[i + 1])
            t12[:3] += 0.01 * rng.normal(size=3)
            ba.AddBinaryConstraint(i, i + 1, t12)
    return n
