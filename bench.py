#!/usr/bin/env python3
"""Gauss-Newton iterations/sec of the MI355X BundleAdjuster path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one full Gauss-Newton iteration of ba::BundleAdjuster::Solve
(/root/reference/src/BundleAdjuster.cpp:298-663) on a synthetic scene already resident in
HBM: linearise (residuals, Huber median, Jacobians, V/W, gather S and rhs) -> dense
Cholesky solve -> landmark back-substitution -> EvaluateResiduals -> ApplyUpdate ->
EvaluateResiduals -> accept / roll back.  Exit tests are disabled (fixed step count).

Workload (default, `--config 3`): the scene BASELINE.json's metric is quoted on — configs[3],
10k poses / 1M landmarks / 10M reprojection residuals, pinhole camera, inverse-depth landmarks
(LmSize = 1), n = 59 988 reduced unknowns; it fits one MI355X (S is 28.8 GB of the 288 GB).
`--config 1` selects configs[1] (1k poses / 100k landmarks / 1M residuals).  For N > 1 the SAME
scene is sharded by landmark across the ranks (every rank holds all poses); the partial
reduced pose systems are reduce-scattered (RCCL over xGMI) onto the owners of S's column
panels, every panel is factorised by its owner and broadcast, and each rank applies the
trailing updates to the panels it owns; the right-hand side and a few scalars/histograms
are all-reduced ("scaling": "strong").

Rank 0 prints one JSON line with the driver's contract keys plus `roofline` (dominant
kernel, measured live with HIP events on the engine's stream) and `cpu_baseline` (the
oracle — a CPU restatement of the reference — timed on the host cores for ONE iteration
of the same scene).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ba_amd import hipapi, scene, sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md chip table: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6   # AMD datasheet FP64 matrix peak (not listed in the local guide)


def build_engine(sc, lm_dim, lo, hi, device, stream=None):
    """Upload poses (all) and the landmark shard [lo, hi) with its accepted residuals."""
    nsel = sc.obs_per_landmark + (1 if lm_dim == 1 else 0)
    keep = np.ones(len(sc.obs_pose), dtype=bool)
    if lm_dim == 1:
        keep[::nsel] = False  # the reference-frame observation is rejected (BundleAdjuster.h:489-501)
    sel = keep & (sc.obs_lm >= lo) & (sc.obs_lm < hi)
    pa = np.ones(sc.num_poses, dtype=np.uint8)
    pa[sc.anchor_poses] = 0
    eng = hipapi.Engine(lm_dim, 6, device=device, stream=stream)
    eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1])
    eng.set_poses(sc.poses, is_active=pa)
    eng.set_landmarks(sc.landmarks[lo:hi], sc.lm_ref_pose[lo:hi])
    eng.set_projection_residuals(sc.obs_z[sel], sc.obs_pose[sel], sc.obs_lm[sel] - lo)
    eng.finalize()
    eng.begin_solve()
    eng.set_pose_masks(np.zeros(sc.num_poses, dtype=np.uint16))
    return eng, int(sel.sum())


def gn_step(eng):
    """One Gauss-Newton iteration (BundleAdjuster.cpp:298-663, GN branch :1084-1159)."""
    eng.linearize()
    rc = eng.solve_gn()
    if rc != 0:
        raise RuntimeError("reduced system not SPD (rc=%d)" % rc)
    eng.compose_step(0.0, 1.0)
    pre = eng.eval_residuals()
    eng.apply_step()
    post = eng.eval_residuals()
    if post.total() > pre.total():
        eng.rollback()
        return pre.total(), False
    return post.total(), True


def cpu_baseline(sc, lm_dim):
    """The oracle (CPU restatement of the reference, 1 thread as the reference runs its
    projection loop serially, BundleAdjuster.cpp:1345-1347) on ONE iteration of the scene `sc`."""
    from oracle import pyoracle as po
    po.build()
    ba = po.OracleBundleAdjuster(lm_dim, 6)
    o = po.default_options()
    o.use_dogleg = 0
    o.error_change_threshold = 0
    o.param_change_threshold = 0
    ba.Init(o)
    pa = np.ones(sc.num_poses, dtype=np.uint8)
    pa[sc.anchor_poses] = 0
    ba.AddCamera(sc.cam_params)
    ba.add_poses(sc.poses, is_active=pa)
    ba.add_landmarks(sc.landmarks, sc.lm_ref_pose)
    ba.add_projection_residuals(sc.obs_z, sc.obs_pose, sc.obs_lm)
    t0 = time.time()
    ba.Solve(1)
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "iterations/sec", "cores": 1, "kind": "port",
            "sample": "1 Gauss-Newton iteration of the same scene (%.1f s)" % dt,
            "phases_s": {k: round(v, 3) for k, v in ba.timers().items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=[1, 3],
                    help="BASELINE.json configs index: 3 = 10k poses / 1M landmarks / 10M residuals "
                         "(the metric's scene, default), 1 = 1k / 100k / 1M")
    ap.add_argument("--poses", type=int, default=None)
    ap.add_argument("--landmarks", type=int, default=None)
    ap.add_argument("--obs-per-landmark", type=int, default=10)
    ap.add_argument("--lm-dim", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    preset = {1: (1000, 100000, 20, 3), 3: (10000, 1000000, 5, 1)}[args.config]
    if args.poses is None:
        args.poses = preset[0]
    if args.landmarks is None:
        args.landmarks = preset[1]
    if args.steps is None:
        args.steps = preset[2]
    if args.warmup is None:
        args.warmup = preset[3]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # BA_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than
        # ranks (ranks share devices; gloo stages the device tensors through the host)
        backend = os.environ.get("BA_BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)

    P, L, K, lm_dim = args.poses, args.landmarks, args.obs_per_landmark, args.lm_dim
    sc = scene.make_scene(P, L, K, lm_dim=lm_dim, seed=2)
    # landmark shards: contiguous, equal counts (every landmark has K residuals, so the
    # Schur work  sum k(k+1)/2  is balanced too)
    lo, hi = sharding.landmark_shards(np.full(L, K), world)[rank]
    t_setup = time.perf_counter()
    eng, n_obs_local = build_engine(sc, lm_dim, lo, hi, local_rank if world > 1 else 0)
    t_setup = time.perf_counter() - t_setup  # host -> device uploads + structure build (once per Solve())
    if world > 1:
        eng.set_allreduce(sharding.torch_allreduce_hook(dist, "cuda"), rank, world)
        # distributed reduced solve: reduce-scatter of S to the panel owners, per-panel
        # factorisation + broadcast (BA_BENCH_REPLICATED_SOLVE=1 keeps the replicated solve)
        if not os.environ.get("BA_BENCH_REPLICATED_SOLVE"):
            eng.set_collectives(sharding.torch_collectives_hook(dist, "cuda"))

    def barrier():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        gn_step(eng)
    eng.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    accepted = 0
    err = 0.0
    for _ in range(args.steps):
        err, ok = gn_step(eng)
        accepted += int(ok)
    barrier()
    elapsed = time.perf_counter() - t0
    ks = eng.kernel_stats()
    timers = eng.get_timers()
    eng.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng.end_solve()

    if rank == 0:
        n = eng.num_pose_params()
        O = L * K
        ms = 1e3 * elapsed / args.steps
        # dominant kernel: the trailing update of the dense LDL^T factorisation (FP64 MFMA);
        # timed with HIP events on the stream it is launched on (the engine's second stream)
        syrk_tf = ks.syrk_flops / (ks.syrk_ms * 1e-3) / 1e12 if ks.syrk_ms > 0 else 0.0
        # HBM-bound kernels, algorithmic bytes per launch (DESIGN.md §Roofline accounting)
        ell = lm_dim
        b_landmarks = O / world * 32 + (L / world) * 36 + P * 56 + (L / world) * 8 * (ell * ell + ell)
        b_gather = 8.0 * n * (n + 1) / 2 + 8 * n
        lm_gbs = b_landmarks * ks.landmarks_launches / (ks.landmarks_ms * 1e-3) / 1e9 if ks.landmarks_ms > 0 else 0.0
        ga_gbs = b_gather * ks.gather_launches / (ks.gather_ms * 1e-3) / 1e9 if ks.gather_ms > 0 else 0.0
        # HBM-side traffic of the dominant kernel: not measurable live; taken from the committed
        # rocprofv3 --pmc passes of this same command (profiles/, FETCH_SIZE x2 + WRITE_SIZE,
        # per launch), null if the summary is absent
        traffic = None
        pmc = {}
        # 128x128 blocks on trailing matrices of >= 128 tiles, the capped 64-tile kernel below
        bulk_kernel = "k_update128<false>" if n >= 128 * 64 else "k_update2<true>"
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_cfg%d.json" % args.config)) as f:
                pmc = json.load(f)["kernels"]
            bulk = "bae::k_update128<false>" if n >= 128 * 64 else "bae::k_update2<true>"
            traffic = pmc[bulk]["traffic_bytes_per_launch_corrected"]
        except (OSError, KeyError, ValueError):
            pass

        def pmc_bytes(name):
            try:
                return pmc[name]["traffic_bytes_per_launch_corrected"]
            except KeyError:
                return None
        out = {
            "metric": "Gauss-Newton iterations/sec",
            "value": args.steps / elapsed,
            "unit": "iterations/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: %d poses / %d landmarks / %d reprojection "
                                   "residuals, pinhole, LmSize=%d, PoseSize=6, Gauss-Newton (no dogleg), "
                                   "2 anchor poses inactive" % (args.config, P, L, O, lm_dim),
                       "poses": P, "landmarks": L, "residuals": O, "reduced_system_n": n,
                       "parallelism": ("landmark-sharded x%d, reduce-scatter of S to panel owners, distributed LDL^T "
                                       "(panel broadcast)" % world) if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": "%s (dense LDL^T trailing update, v_mfma_f64_16x16x4_f64; the look-ahead's bulk launches)" % bulk_kernel,
                         "achieved": syrk_tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": syrk_tf / FP64_MFMA_PEAK_TF, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_traffic_cfg%d.json (rocprofv3 --pmc, bytes per launch)" % args.config,
                         "flops_per_launch": ks.syrk_flops / max(ks.syrk_launches, 1),
                         "launches": ks.syrk_launches,
                         "avg_launch_us": 1e3 * ks.syrk_ms / max(ks.syrk_launches, 1)},
            "hbm_kernels": {
                # achieved_GBs: compulsory (algorithmic) bytes / time; pmc_traffic_bytes: what the
                # kernel really moved (factor rows are materialised), from the committed PMC passes
                "k_landmarks": {"achieved_GBs": lm_gbs, "frac_of_8TBs": lm_gbs / HBM_PEAK_GBS,
                                "avg_launch_us": 1e3 * ks.landmarks_ms / max(ks.landmarks_launches, 1),
                                "algorithmic_bytes": b_landmarks,
                                "pmc_traffic_bytes": pmc_bytes("bae::k_landmarks<%d>" % lm_dim)},
                "k_gather_S": {"achieved_GBs": ga_gbs, "frac_of_8TBs": ga_gbs / HBM_PEAK_GBS,
                               "avg_launch_us": 1e3 * ks.gather_ms / max(ks.gather_launches, 1),
                               "algorithmic_bytes": b_gather,
                               "pmc_traffic_bytes": pmc_bytes("bae::k_gather_S")}},
            "phase_ms_last_step": {k: round(v, 4) for k, v in timers.items()},
            # one-off per Solve(): PCIe uploads of the scene + host-side structure build (gather
            # lists, tile pattern); NOT part of `value` (inputs are resident when the timed region starts)
            "setup_s_rank0": round(t_setup, 3),
            "accepted_steps": accepted,
            "final_error": err,
        }
        for hk in out["hbm_kernels"].values():  # real (PMC) traffic over the live launch time
            if hk.get("pmc_traffic_bytes") and hk["avg_launch_us"] > 0:
                hk["pmc_GBs"] = hk["pmc_traffic_bytes"] / (hk["avg_launch_us"] * 1e-6) / 1e9
                hk["pmc_frac_of_8TBs"] = hk["pmc_GBs"] / HBM_PEAK_GBS
        if not args.no_cpu_baseline and world == 1:
            if P <= 1000:
                out["cpu_baseline"] = cpu_baseline(sc, lm_dim)
            else:
                # One oracle iteration of THIS scene needs a dense n = 60k LDL^T on one core
                # (~72 TFLOP: over an hour).  Bounded sample: one oracle iteration of the
                # 10x smaller configs[1] scene (same generator, same densities); its O(N)
                # phases are scaled by the residual count, its dense solve by n^3.
                small = scene.make_scene(1000, 100000, K, lm_dim=lm_dim, seed=2)
                cb = cpu_baseline(small, lm_dim)
                ph = cb["phases_s"]
                lin = ph["total"] - ph["solve"]
                f_lin = float(O) / (100000 * K)
                f_sol = (float(n) / (6.0 * 998)) ** 3
                est = lin * f_lin + ph["solve"] * f_sol
                cb["sample"] = ("1 Gauss-Newton iteration of the configs[1] scene (1k poses / 100k landmarks / "
                                "1M residuals, %.1f s measured), extrapolated to this scene: linear phases "
                                "x%.0f (residual count), dense LDL^T x%.0f (n^3) -> %.0f s per iteration"
                                % (ph["total"], f_lin, f_sol, est))
                cb["measured_sample_iterations_per_sec"] = cb["value"]
                cb["value"] = 1.0 / est
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
