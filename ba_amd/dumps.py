"""Loader / cross-check for the reference's debug dumps.

`options.write_reduced_camera_matrix` makes Solve() write, per iteration, s.txt, rhs.txt, j_pr.txt,
r_pr.txt and j_l.txt in Eigen's full-precision CSV format
(/root/reference/src/BundleAdjuster.cpp:600-616, include/ba/Utils.h:66) into the working directory.
This module reads such a directory — written by this repo's ba::BundleAdjuster or by a build of the
original arpg/ba — so that the two can be diffed, and checks the files against each other:

    S   = U - W V^-1 W^T        with U = j_pr^T j_pr, W = j_pr^T j_l, V = j_l^T j_l
    rhs = j_pr^T r_pr - W V^-1 j_l^T r_pr                      (BundleAdjuster.cpp:337-354, 409-485)

numpy only; nothing here touches the GPU.
"""
import os

import numpy as np

FILES = ("s.txt", "rhs.txt", "j_pr.txt", "r_pr.txt", "j_l.txt")


def _load(path):
    a = np.loadtxt(path, delimiter=",", ndmin=2)
    return a


def load_reduced_system(directory="."):
    """{'s': (n,n), 'rhs': (n,), 'j_pr': (2O, 6P), 'r_pr': (2O,), 'j_l': (2O, lm L)} for the files present."""
    out = {}
    for name in FILES:
        p = os.path.join(directory, name)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            a = _load(p)
            out[name[:-4]] = a[:, 0] if name in ("rhs.txt", "r_pr.txt") else a
    return out


def schur_from_jacobians(j_pr, r_pr, j_l, lm_dim, pose_dim=6, v_guard=True):
    """Reduced system from the dumped Jacobians, dense numpy (the reference's algebra)."""
    U = j_pr.T @ j_pr
    W = j_pr.T @ j_l
    V = j_l.T @ j_l
    nl = V.shape[0] // lm_dim
    Vi = np.zeros_like(V)
    for k in range(nl):
        sl = slice(k * lm_dim, (k + 1) * lm_dim)
        blk = V[sl, sl].copy()
        if v_guard:
            if lm_dim == 1:
                if abs(blk[0, 0]) < 1e-6:
                    blk[0, 0] += 1e-6                      # BundleAdjuster.cpp:431-434
            elif np.linalg.norm(blk) < 1e-6:
                blk += 1e-6 * np.eye(lm_dim)              # :435-439
        Vi[sl, sl] = np.linalg.inv(blk)
    S = U - W @ Vi @ W.T
    rhs = j_pr.T @ r_pr - W @ Vi @ (j_l.T @ r_pr)
    return S, rhs


def check_consistency(d, lm_dim, pose_dim=6, masked_diag=1e6):
    """Relative mismatch of s.txt / rhs.txt against the system rebuilt from j_pr / r_pr / j_l
    (projection-only problems with PoseSize 6; the upper block triangle when the dump is
    triangular; masked parameters carry `masked_diag` on the diagonal, :587-598)."""
    S, rhs = schur_from_jacobians(d["j_pr"], d["r_pr"], d["j_l"], lm_dim, pose_dim)
    s = d["s"]
    n = s.shape[0]
    tri = not np.any(np.tril(s, -pose_dim))   # use_triangular_matrices: blocks below the diagonal are empty
    mask = np.ones((n, n), dtype=bool)
    if tri:
        nb = n // pose_dim
        for i in range(nb):
            mask[i * pose_dim:(i + 1) * pose_dim, :i * pose_dim] = False
    fixed = np.isclose(np.diag(s), masked_diag) & np.isclose(np.diag(S), 0.0)
    S = S.copy()
    S[np.diag_indices(n)] = np.where(fixed, masked_diag, np.diag(S))
    err_s = np.abs((s - S)[mask]).max() / max(np.abs(S).max(), 1e-300)
    err_r = np.abs(d["rhs"] - rhs).max() / max(np.abs(rhs).max(), 1e-300)
    return err_s, err_r


def diff(dir_a, dir_b):
    """Relative max-norm difference per file of two dump directories (e.g. this repo vs arpg/ba)."""
    a, b = load_reduced_system(dir_a), load_reduced_system(dir_b)
    out = {}
    for k in a:
        if k in b and a[k].shape == b[k].shape:
            out[k] = float(np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-300))
        else:
            out[k] = None
    return out
