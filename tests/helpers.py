"""Shared helpers for the parity tests (numpy only)."""
import numpy as np

from ba_amd import scene


def gn_options(po, **kw):
    """Options for fixed-count Gauss-Newton runs (exit tests disabled)."""
    o = po.default_options()
    o.use_dogleg = 0
    o.error_change_threshold = 0
    o.param_change_threshold = 0
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def fill(ba, sc, active=None, lm_active=None):
    ba.AddCamera(sc.cam_params)
    ba.add_poses(sc.poses, v_w=getattr(sc, "init_vel", None), b=getattr(sc, "init_bias", None),
                 is_active=active, time=getattr(sc, "pose_time", None))
    ba.add_landmarks(sc.landmarks, sc.lm_ref_pose, is_active=lm_active)
    return ba.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)


def accepted_obs(sc):
    """(meas pose, ref pose, landmark) per ACCEPTED residual id, in residual-id order."""
    nsel = sc.obs_per_landmark + (1 if sc.lm_dim == 1 else 0)
    out = []
    for i in range(len(sc.obs_pose)):
        if sc.lm_dim == 1 and i % nsel == 0:
            continue  # the reference-frame observation is rejected (BundleAdjuster.h:489-501)
        out.append((int(sc.obs_pose[i]), int(sc.lm_ref_pose[sc.obs_lm[i]]), int(sc.obs_lm[i])))
    return out


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
