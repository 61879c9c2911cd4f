"""Randomised soak of the distributed reduced solve on thread-emulated ranks (one GPU): random scene sizes, rank
counts 2..8, ownership layouts, panel widths, landmark shards contiguous or dealt along the trajectory.  Every
trial: the step is bitwise equal across the ranks, errors / step norms agree with ONE engine on the whole scene to
1e-8, final poses to 1e-9, and the bytes the ranks moved equal the message plan.
    python scratch/gpu_dist_soak.py <seed> <trials> [mid]      (mid: 1000-2600 poses, 94-244 tiles: the 128-block kernels)
(test infrastructure for DESIGN section 6a; the pytest case test_distributed_solve_matches_single is the fixed-seed form)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ba_amd import hipapi, scene, sharding  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
MID = len(sys.argv) > 3 and sys.argv[3] == "mid"


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run_steps(eng, iters, out, key):
    try:
        res = []
        for _ in range(iters):
            e0 = eng.linearize()
            rc = eng.solve_gn()
            if not res:
                out[(key, "delta_p")] = eng.get_delta_gn()[0]
            nrm = eng.compose_step(0.0, 1.0)
            pre = eng.eval_residuals()
            eng.apply_step()
            post = eng.eval_residuals()
            if post.total() > pre.total():
                eng.rollback()
            res.append((rc, e0.proj_error, pre.total(), post.total(), nrm.step_p_norm, nrm.step_l_norm))
        out[key] = res
    except Exception as exc:
        out[key] = exc


bad = 0
t_start = time.time()
for trial in range(N):
    nranks = int(rng.choice([2, 3, 4, 5, 6, 8]))
    P = int(rng.integers(1000, 2600)) if MID else int(rng.integers(60, 420))
    k = int(rng.integers(4, 9))
    L = int(rng.integers(6, 14)) * P
    kout = int(rng.choice([4, 6, 8, 16] if MID else [2, 3, 4, 5, 6]))
    lay = str(rng.choice(["auto", "auto", "row", "col", "grid"]))
    along = bool(rng.integers(0, 2))
    seed = int(rng.integers(1, 10**6))
    os.environ["BA_HIP_KOUT"] = str(kout)
    if lay == "auto":
        os.environ.pop("BA_HIP_DIST_LAYOUT", None)
    else:
        os.environ["BA_HIP_DIST_LAYOUT"] = lay
    sc = scene.make_scene(P, L, k, lm_dim=1, seed=seed)
    pa = np.ones(sc.num_poses, dtype=np.uint8)
    pa[sc.anchor_poses] = 0
    nsel = sc.obs_per_landmark + 1
    keep = np.ones(len(sc.obs_pose), dtype=bool)
    keep[::nsel] = False
    if along:
        ids_by_rank = sharding.landmark_shards_along_trajectory(sc.lm_ref_pose, np.full(L, sc.obs_per_landmark), nranks)
    else:
        ids_by_rank = [np.arange(a, b) for a, b in sharding.landmark_shards(np.full(L, sc.obs_per_landmark), nranks)]

    def make(ids):
        ids = np.asarray(ids)
        remap = -np.ones(L, dtype=np.int64)
        remap[ids] = np.arange(len(ids))
        sel = keep & (remap[sc.obs_lm] >= 0)
        eng = hipapi.Engine(1, 6)
        eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1])
        eng.set_poses(sc.poses, is_active=pa)
        eng.set_landmarks(sc.landmarks[ids], sc.lm_ref_pose[ids])
        eng.set_projection_residuals(sc.obs_z[sel], sc.obs_pose[sel], remap[sc.obs_lm[sel]].astype(np.uint32))
        eng.finalize()
        eng.begin_solve()
        eng.set_pose_masks(np.zeros(sc.num_poses, dtype=np.uint16))
        return eng

    tag = "trial %3d  ranks %d  poses %3d  lms %5d  k %d  KOUT %d  layout %-4s  shards %s  seed %6d" % (
        trial, nranks, P, L, k, kout, lay, "trajectory" if along else "contiguous", seed)
    try:
        hipapi.dist_plan_stats(16, None, nranks, lay, kout)  # (a layout that does not exist for this rank count: refused)
        single = make(np.arange(L))
        out = {}
        iters = 2
        run_steps(single, iters, out, "single")
        engs = [make(ids_by_rank[r]) for r in range(nranks)]
        ar = sharding.ThreadAllReduce(nranks)
        for r in range(nranks):
            engs[r].set_allreduce(ar.hook(r), r, nranks)
            engs[r].set_collectives(ar.collectives(r))
        if not all(e.solve_is_distributed() for e in engs):
            raise RuntimeError("solve not distributed")
        th = [threading.Thread(target=run_steps, args=(engs[r], iters, out, r)) for r in range(nranks)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        if ar.failed:
            raise RuntimeError("hook failure")
        for key in ["single"] + list(range(nranks)):
            if isinstance(out[key], Exception):
                raise out[key]
        worst = 0.0
        for it in range(iters):
            a = out["single"][it]
            for r in range(nranks):
                b = out[r][it]
                if not (a[0] == b[0] == 0):
                    raise RuntimeError("status %r / %r" % (a[0], b[0]))
                for x, y in zip(a[1:], b[1:]):
                    worst = max(worst, abs(x - y) / max(abs(x), 1e-12))
        ps = single.get_poses(sc.num_poses)[0]
        p0 = engs[0].get_poses(sc.num_poses)[0]
        bitwise = True
        dpose = 0.0
        for r in range(nranks):
            pr = engs[r].get_poses(sc.num_poses)[0]
            dpose = max(dpose, rel_err(pr, ps))
            bitwise &= bool(np.array_equal(pr, p0)) and bool(np.array_equal(out[(r, "delta_p")], out[(0, "delta_p")]))
        nzL = single.factor_tile_pattern()
        plan = hipapi.dist_plan_stats(nzL.shape[0], nzL, nranks, lay, kout)
        cs = [engs[r].comm_stats() for r in range(nranks)]
        bytes_ok = (abs(sum(c["chain_bytes_recv"] for c in cs) - iters * plan["chain_recv_total"]) <= 1e-9 * max(1.0, iters * plan["chain_recv_total"])
                    and abs(sum(c["side_bytes_recv"] for c in cs) - iters * plan["side_recv_total"]) <= 1e-9 * max(1.0, iters * plan["side_recv_total"]))
        ok = bitwise and worst <= 1e-8 and dpose <= 1e-9 and bytes_ok
        print("%s  tiles %3d  scalars %.1e  poses %.1e  bitwise %s  bytes %s  %s" % (
            tag, nzL.shape[0], worst, dpose, bitwise, bytes_ok, "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
        for e_ in engs + [single]:
            e_.end_solve()
            e_.close()
    except ValueError as exc:  # a layout that does not exist for this rank count is refused, by design
        print("%s  refused: %s" % (tag, str(exc)[:80]), flush=True)
    except Exception as exc:
        print("%s  ERROR %r" % (tag, exc), flush=True)
        bad += 1
print("trials %d, mismatches or errors %d, %.0f s" % (N, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
