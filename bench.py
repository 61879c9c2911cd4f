#!/usr/bin/env python3
"""Gauss-Newton iterations/sec of the MI355X BundleAdjuster path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4]

A "step" is one full iteration of ba::BundleAdjuster::Solve
(/root/reference/src/BundleAdjuster.cpp:298-663) on a synthetic scene already resident in
HBM: linearise (residuals, Huber median, Jacobians, V/W, reduced system S and rhs) -> dense
LDL^T solve -> landmark back-substitution -> EvaluateResiduals -> ApplyUpdate ->
EvaluateResiduals -> accept / roll back.  Exit tests are disabled (fixed step count).

Workloads (`--config` = index into BASELINE.json `configs`):
  3 (default) 10k poses / 1M landmarks / 10M reprojection residuals, LmSize 1, PoseSize 6, GN —
              the scene the metric is quoted on; n = 59 988, S = 28.8 GB: fits one MI355X
  1           1k / 100k / 1M, same shape
  2           5k / 500k / 5M + IMU pre-integration residuals, PoseSize 15 (n = 75 000), GN
  4           10k / 1M / 10M + IMU + unary priors + binary odometry, PoseSize 15 (n = 150 000,
              S = 180 GB on ONE GPU), dogleg trust region
Drivers: configs 1 / 3 time the phase calls of the C-ABI (include/ba_hip.h) — `value` — and, at
N = 1, additionally `api_solve_ms`: wall time of ba::BundleAdjuster::Solve(1) on a warm object,
the number a user of the C++ class sees.  Configs 2 / 4 are timed through that C++ API path
only (Solve(1) per step on a warm object).

For N > 1 the SAME scene (config 1 / 3) is sharded by landmark across the ranks (every rank
holds all poses); the partial reduced pose systems are reduce-scattered (RCCL over xGMI) onto the
owners of S's column panels, every panel is factorised by its owner and broadcast, and each rank
applies the trailing updates to the panels it owns; the right-hand side and a few
scalars/histograms are all-reduced ("scaling": "strong").

Rank 0 prints one JSON line with the driver's contract keys plus `roofline` (dominant kernel,
measured live with HIP events on the stream it runs on) and `cpu_baseline` (the oracle — a CPU
restatement of the reference — timed on the host cores on a bounded sample of the workload).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ba_amd import adjuster, hipapi, scene, sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md chip table: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6   # AMD datasheet FP64 matrix peak (not listed in the local guide)

CONFIGS = {
    # poses, landmarks, pose_dim, imu, priors, dogleg, default steps, warmup, CPU-sample poses / landmarks
    1: dict(P=1000, L=100000, D=6, imu=False, priors=False, dogleg=False, steps=20, warmup=3, sP=1000, sL=100000),
    2: dict(P=5000, L=500000, D=15, imu=True, priors=False, dogleg=False, steps=4, warmup=1, sP=500, sL=50000),
    3: dict(P=10000, L=1000000, D=6, imu=False, priors=False, dogleg=False, steps=5, warmup=1, sP=1000, sL=100000),
    4: dict(P=10000, L=1000000, D=15, imu=True, priors=True, dogleg=True, steps=2, warmup=1, sP=500, sL=50000),
}


def log(msg):
    """progress on stderr (the JSON line on stdout stays alone)"""
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def usable_cpus(cap=16):
    """CPU threads this process may really use: affinity mask, cgroup quota, and the box's share
    (16 for one GPU) — oversubscribed OpenMP teams spin against the CFS quota and crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def make_workload(cfg, P, L, K, lm_dim, fov_w=None):
    sc = scene.make_scene(P, L, K, lm_dim=lm_dim, seed=2)
    if fov_w:
        scene.to_fov_camera(sc, fov_w)  # the same scene seen through a FOV camera (not a BASELINE configuration)
    if cfg["imu"]:
        scene.add_inertial(sc, period=60.0 * P / 100.0)
    return sc


def active_mask(cfg, sc):
    """configs 1 / 3: two anchor poses inactive fix the monocular gauge; the visual-inertial
    scenes are all-active (gravity + velocities observable; root pose auto-regularised,
    BundleAdjuster.cpp:1285-1330) or carry priors."""
    if cfg["imu"]:
        return None
    pa = np.ones(sc.num_poses, dtype=np.uint8)
    pa[sc.anchor_poses] = 0
    return pa


def build_engine(sc, lm_dim, lo, hi, device, stream=None, ids=None):
    """C-ABI driver: upload poses (all) and the landmark shard [lo, hi) — or the landmark ids `ids` (ascending) —
    with its accepted residuals."""
    nsel = sc.obs_per_landmark + (1 if lm_dim == 1 else 0)
    keep = np.ones(len(sc.obs_pose), dtype=bool)
    if lm_dim == 1:
        keep[::nsel] = False  # the reference-frame observation is rejected (BundleAdjuster.h:489-501)
    if ids is not None:
        return _build_engine_ids(sc, lm_dim, keep, np.asarray(ids), device, stream)
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


def _build_engine_ids(sc, lm_dim, keep, ids, device, stream):
    new_id = np.full(sc.num_landmarks, -1, dtype=np.int64)
    new_id[ids] = np.arange(len(ids))
    sel = keep & (new_id[sc.obs_lm] >= 0)
    pa = np.ones(sc.num_poses, dtype=np.uint8)
    pa[sc.anchor_poses] = 0
    eng = hipapi.Engine(lm_dim, 6, device=device, stream=stream)
    eng.set_cameras(sc.cam_params, [0, 0, 0, 0, 0, 0, 1])
    eng.set_poses(sc.poses, is_active=pa)
    eng.set_landmarks(sc.landmarks[ids], sc.lm_ref_pose[ids])
    eng.set_projection_residuals(sc.obs_z[sel], sc.obs_pose[sel], new_id[sc.obs_lm[sel]].astype(np.uint32))
    eng.finalize()
    eng.begin_solve()
    eng.set_pose_masks(np.zeros(sc.num_poses, dtype=np.uint16))
    return eng, int(sel.sum())


def gn_step(eng):
    """One Gauss-Newton iteration (BundleAdjuster.cpp:298-663, GN branch :1084-1159)."""
    eng.linearize()
    rc = eng.solve_gn()
    if rc != 0 and not os.environ.get("BA_BENCH_IGNORE_RC"):   # (measurement builds with wrong numbers set it)
        raise RuntimeError("reduced system not SPD (rc=%d)" % rc)
    eng.compose_step(0.0, 1.0)
    pre = eng.eval_residuals()
    eng.apply_step()
    post = eng.eval_residuals()
    if post.total() > pre.total():
        eng.rollback()
        return pre.total(), False
    return post.total(), True


def api_options(mod, cfg):
    o = mod.default_options()
    o.use_dogleg = 1 if cfg["dogleg"] else 0
    o.error_change_threshold = 0
    o.param_change_threshold = 0
    return o


def build_adjuster(cfg, sc, lm_dim, lm_range=None, pose_pose=True, device=0, lm_ids=None):
    """C++ API driver: ba::BundleAdjuster<double, LmSize, PoseSize> through include/ba_capi.h.
    lm_range / pose_pose: this rank's landmark shard; inertial / unary / binary residuals on rank 0 only."""
    h = adjuster.BundleAdjuster(lm_dim, cfg["D"])
    o = api_options(adjuster, cfg)
    o.device = device
    h.Init(o)
    scene.populate(h, sc, active=active_mask(cfg, sc), imu=cfg["imu"], priors=cfg["priors"], lm_range=lm_range,
                   pose_pose=pose_pose, lm_ids=lm_ids)
    return h


def cpu_baseline(cfg, K, lm_dim, threads):
    """The oracle (CPU restatement of the reference) on ONE iteration of a bounded sample of the
    workload: the same generator and densities at cfg['sP'] poses / cfg['sL'] landmarks.
    threads = 1: reference-faithful (projection loop and LDL^T are serial in the reference,
    BundleAdjuster.cpp:1345-1347, 752-799); threads > 1: best-effort CPU mode (dense LDL^T on all
    cores, BASELINE.md §3 ii)."""
    from oracle import pyoracle as po
    po.build()
    po.set_num_threads(threads)
    sc = make_workload(cfg, cfg["sP"], cfg["sL"], K, lm_dim)
    ba = po.OracleBundleAdjuster(lm_dim, cfg["D"])
    ba.Init(api_options(po, cfg))
    scene.populate(ba, sc, active=active_mask(cfg, sc), imu=cfg["imu"], priors=cfg["priors"])
    t0 = time.time()
    ba.Solve(1)
    dt = time.time() - t0
    po.set_num_threads(1)
    return dt, {k: round(v, 3) for k, v in ba.timers().items()}, ba.num_pose_params()


def cpu_ldlt_rate(n, threads):
    """FP64 GFLOP/s of the oracle's dense LDL^T at size n from the committed three-point fit
    (profiles/r02_cpu_ldlt_fit.json, measured on the GPU box's host cores by scratch/cpu_ldlt_fit.py),
    or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_cpu_ldlt_fit.json")) as f:
            fit = json.load(f)
        pts = fit["threads_%d" % threads] if ("threads_%d" % threads) in fit else fit["all_cores" if threads > 1 else "threads_1"]
        ns = np.array([p["n"] for p in pts], dtype=float)
        gf = np.array([p["gflops"] for p in pts], dtype=float)
        return float(np.interp(min(n, ns.max()), ns, gf)), fit.get("cpu_model", "")
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline_record(cfg, args, n_full, O_full, threads):
    dt, ph, n_s = cpu_baseline(cfg, args.obs_per_landmark, args.lm_dim, threads)
    O_s = cfg["sL"] * args.obs_per_landmark
    rec = {"unit": "iterations/sec", "cores": threads, "kind": "port", "phases_s": ph,
           "mode": "reference-faithful (1 thread)" if threads == 1 else "best-effort (dense LDL^T on %d threads)" % threads}
    if cfg["sP"] == args.poses and cfg["sL"] == args.landmarks:
        rec["value"] = 1.0 / dt
        rec["sample"] = "1 iteration of the same scene (%.1f s)" % dt
        return rec
    # extrapolation to the full scene: O(residuals) phases by the residual count, the dense LDL^T
    # by its flop count at the CPU's measured LDL^T rate (three-point fit under profiles/, else the
    # rate of this sample's own solve)
    lin = ph["total"] - ph["solve"]
    f_lin = float(O_full) / O_s
    rate_sample = (n_s ** 3 / 3.0) / max(ph["solve"], 1e-9) / 1e9
    fit = cpu_ldlt_rate(n_full, threads)
    rate = fit[0] if fit else rate_sample
    t_solve = (float(n_full) ** 3 / 3.0) / (rate * 1e9)
    est = lin * f_lin + t_solve
    rec["value"] = 1.0 / est
    rec["measured_sample_iterations_per_sec"] = 1.0 / dt
    rec["extrapolated"] = True
    rec["sample"] = ("1 iteration of a %d-pose / %d-landmark / %d-residual scene of the same generator (n = %d, %.1f s "
                     "measured), extrapolated: linear phases x%.0f (residual count), dense LDL^T n^3/3 = %.2e flop at "
                     "%.1f GFLOP/s (%s) -> %.0f s per iteration"
                     % (cfg["sP"], cfg["sL"], O_s, n_s, dt, f_lin, float(n_full) ** 3 / 3.0, rate,
                        "profiles/r02_cpu_ldlt_fit.json" if fit else "this sample's solve", est))
    return rec


def newest_pmc(config):
    """Committed rocprofv3 --pmc summary for this configuration (profiles/rNN_pmc_traffic_cfgC.json)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_cfg%d.json" % config)))
    if not files:
        return {}, None
    try:
        with open(files[-1]) as f:
            return json.load(f)["kernels"], os.path.relpath(files[-1], ROOT)
    except (OSError, KeyError, ValueError):
        return {}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs index (3 = the metric's scene, default)")
    ap.add_argument("--poses", type=int, default=None)
    ap.add_argument("--landmarks", type=int, default=None)
    ap.add_argument("--obs-per-landmark", type=int, default=10)
    ap.add_argument("--lm-dim", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fov-w", type=float, default=None,
                    help="variant workload: the scene through a calibu::FovCamera with this distortion parameter "
                         "(times the FOV instantiations of the projection kernels; configs 1 / 3)")
    ap.add_argument("--live-pmc", action="store_true",
                    help="after the timed region, run the two rocprofv3 --pmc passes of this configuration as child "
                         "processes (scratch/pmc_traffic.py; minutes) and report THEIR traffic instead of the committed summary")
    ap.add_argument("--no-api", action="store_true", help="skip the api_solve_ms pass of configs 1 / 3")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.poses is None:
        args.poses = cfg["P"]
    if args.landmarks is None:
        args.landmarks = cfg["L"]
    if args.steps is None:
        args.steps = cfg["steps"]
    if args.warmup is None:
        args.warmup = cfg["warmup"]
    api_driver = cfg["imu"] or cfg["priors"]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = torch = None
    # BA_BENCH_COMM=native (default): the engine's own RCCL communicator (ba_hip_comm_init) carries
    # every collective; torch.distributed (gloo) is only the launcher's control plane (unique-id
    # exchange, barriers, the max over ranks).  BA_BENCH_COMM=torch: collectives through
    # torch.distributed hooks (backend nccl = RCCL; BA_BENCH_BACKEND=gloo for a CPU-staged rehearsal
    # on a box with fewer GPUs than ranks).
    comm_mode = os.environ.get("BA_BENCH_COMM", "native")
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("BA_BENCH_BACKEND", "gloo" if comm_mode == "native" else "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)

    P, L, K, lm_dim = args.poses, args.landmarks, args.obs_per_landmark, args.lm_dim
    sc = make_workload(cfg, P, L, K, lm_dim, args.fov_w)
    log("scene ready: %d poses, %d landmarks" % (P, L))

    def barrier():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    api_ms = first_solve_s = None
    accepted = 0
    err = 0.0
    if not api_driver:
        # landmark shards: contiguous, equal counts (every landmark has K residuals, so the
        # Schur work  sum k(k+1)/2  is balanced too)
        lo, hi = sharding.landmark_shards(np.full(L, K), world)[rank]
        # N > 1: shards dealt ALONG THE TRAJECTORY (landmarks in the order of their reference pose), so that a shard's
        # partial S is a band and the sparse exchange of S moves a fraction of the matrix (BA_BENCH_SHARDS=id: contiguous ids)
        ids = None
        if world > 1 and os.environ.get("BA_BENCH_SHARDS", "trajectory") != "id":
            ids = sharding.landmark_shards_along_trajectory(sc.lm_ref_pose, np.full(L, K), world)[rank]
        t_setup = time.perf_counter()
        eng, _ = build_engine(sc, lm_dim, lo, hi, local_rank if world > 1 else 0, ids=ids)
        t_setup = time.perf_counter() - t_setup  # host -> device uploads + structure build (once per graph)
        if world > 1 and comm_mode == "native":
            # native RCCL inside the engine: all-reduce, reduce-scatter of S to the panel owners,
            # per-panel broadcast in stream order (BA_HIP_NO_DIST_SOLVE=1 keeps the replicated solve)
            ids = [hipapi.Engine.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            eng.comm_init(ids[0], rank, world)
        elif world > 1:
            eng.set_allreduce(sharding.torch_allreduce_hook(dist, "cuda"), rank, world)
            # distributed reduced solve: reduce-scatter of S to the panel owners, per-panel
            # factorisation + broadcast (BA_BENCH_REPLICATED_SOLVE=1 keeps the replicated solve)
            if not os.environ.get("BA_BENCH_REPLICATED_SOLVE"):
                eng.set_collectives(sharding.torch_collectives_hook(dist, "cuda"))
        log("engine ready (setup %.2f s)" % t_setup)
        for _ in range(args.warmup):
            gn_step(eng)
        log("warmup done")
        eng.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            err, ok = gn_step(eng)
            accepted += int(ok)
        barrier()
        elapsed = time.perf_counter() - t0
        log("timed region done: %.1f ms / step" % (1e3 * elapsed / args.steps))
        ks = eng.kernel_stats()
        timers = eng.get_timers()
        stats = eng.structure_stats() if world == 1 else {}
        eng.set_profiling(False)
        n = eng.num_pose_params()
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        eng.end_solve()
        eng.close()
        if world == 1 and not args.no_api:
            # the same scene through ba::BundleAdjuster::Solve(1): first call = upload + structure
            # build + one iteration; later calls on the warm object must not rebuild anything
            h = build_adjuster(cfg, sc, lm_dim)
            t1 = time.perf_counter()
            h.Solve(1)
            first_solve_s = time.perf_counter() - t1
            h.Solve(1)
            reps = max(2, min(args.steps, 5))
            t1 = time.perf_counter()
            for _ in range(reps):
                h.Solve(1)
            api_ms = 1e3 * (time.perf_counter() - t1) / reps
            log("C++ API path: first Solve(1) %.2f s, warm Solve(1) %.1f ms" % (first_solve_s, api_ms))
            del h
    else:
        # configs 2 / 4 through ba::BundleAdjuster.  N > 1: every rank holds all poses and its landmark shard,
        # the pose-pose residuals (inertial, unary, binary) live on rank 0, the gauge masks follow from
        # GLOBAL residual counts, and the class joins the engine-owned RCCL communicator (SetCommunicator)
        if world > 1:
            nsel = K + (1 if lm_dim == 1 else 0)
            lo, hi = sharding.landmark_shards(np.full(L, K), world)[rank]
            ids = None
            if os.environ.get("BA_BENCH_SHARDS", "trajectory") != "id":   # shards along the trajectory (see the C-ABI driver)
                ids = sharding.landmark_shards_along_trajectory(sc.lm_ref_pose, np.full(L, K), world)[rank]
            h = build_adjuster(cfg, sc, lm_dim, lm_range=(lo, hi), pose_pose=(rank == 0), device=local_rank, lm_ids=ids)
            if comm_mode == "native":
                ids = [hipapi.Engine.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                h.set_communicator(ids[0], rank, world, distributed_solve=not os.environ.get("BA_BENCH_REPLICATED_SOLVE"))
            else:  # rehearsal on a box with fewer GPUs than ranks: the class's all-reduce hook (replicated solve)
                h.set_allreduce(sharding.torch_allreduce_hook(dist, "cuda"), rank, world)
        else:
            h = build_adjuster(cfg, sc, lm_dim)
        log("graph built on the host")
        barrier()
        t_setup = time.perf_counter()
        h.Solve(1)       # upload + structure build + the first iteration
        t_setup = time.perf_counter() - t_setup
        log("first Solve(1): %.2f s" % t_setup)
        first_solve_s = t_setup
        for _ in range(max(args.warmup - 1, 0)):
            h.Solve(1)
        ev = h.engine()
        ev.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            h.Solve(1)
            s = h.summary()
            accepted += int(adjuster.RESULT_NAMES[s.result] == "Success")
            err = s.post_solve_norm if cfg["dogleg"] else s.proj_error + s.inertial_error + s.unary_error + s.binary_error
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        log("timed region done: %.1f ms / step" % (1e3 * elapsed / args.steps))
        ks = ev.kernel_stats()
        timers = ev.get_timers()
        stats = ev.structure_stats() if world == 1 else {}
        ev.set_profiling(False)
        n = ev.num_pose_params()
        api_ms = 1e3 * elapsed / args.steps
        t_setup -= api_ms * 1e-3

    if rank == 0:
        O = L * K
        ms = 1e3 * elapsed / args.steps
        # dominant kernel: the trailing update of the dense LDL^T factorisation (FP64 MFMA);
        # timed with HIP events on the stream it is launched on (the engine's second stream)
        syrk_tf = ks.syrk_flops / (ks.syrk_ms * 1e-3) / 1e12 if ks.syrk_ms > 0 else 0.0
        # HBM-bound kernels, algorithmic bytes per launch: every input read once,
        # every output written once.  k_linearize: observation records, landmarks, poses in; the
        # observation-major factor rows, weights, scaled residuals and landmark blocks out.
        # k_assemble_tiles: the rank-1 term lists, the two 48-byte rows of every term and the tile
        # references in; every 64x64 tile of the FACTOR's pattern out (the zero-fill is part of it).
        # k_pose_blocks: its term list (12 B) and two rows per term in.
        ell = lm_dim
        R = 6 if lm_dim == 1 else 8
        Ow, Lw = O / world, L / world
        b_landmarks = (Ow * 32 + Lw * 36 + P * 56) + Ow * (R * 48 + 8 + 16 + 16 * ell) + Lw * ((96 if ell == 1 else 0) + 8 * (ell * ell + 2 * ell))
        if stats:
            b_gather = stats["tiles_L"] * 32768.0 + stats["pair_entries"] * (8 + 96) + stats["tile_refs"] * 8
            b_pose = stats["pose_entries"] * (12 + 96)
        else:
            b_gather = b_pose = 0.0
        lm_gbs = b_landmarks * ks.landmarks_launches / (ks.landmarks_ms * 1e-3) / 1e9 if ks.landmarks_ms > 0 else 0.0
        ga_gbs = b_gather * ks.gather_launches / (ks.gather_ms * 1e-3) / 1e9 if ks.gather_ms > 0 else 0.0
        po_gbs = b_pose * ks.pose_blocks_launches / (ks.pose_blocks_ms * 1e-3) / 1e9 if ks.pose_blocks_ms > 0 else 0.0
        # HBM-side traffic of the dominant kernel: not measurable live; taken from the committed
        # rocprofv3 --pmc passes of this same command (profiles/, FETCH_SIZE x2 + WRITE_SIZE,
        # per launch), null if no summary exists for this configuration
        pmc, pmc_file = newest_pmc(args.config) if not args.fov_w else ({}, None)
        if args.live_pmc and world == 1 and not api_driver:
            # this process has released its engine; the passes are separate processes under rocprofv3
            import subprocess
            live = os.path.join(ROOT, "gpurun_out", "r03_pmc_traffic_cfg%d.json" % args.config)
            if os.path.exists(live):
                os.remove(live)
            os.makedirs(os.path.dirname(live), exist_ok=True)
            log("PMC passes (FETCH_SIZE, WRITE_SIZE) as child processes ...")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "pmc_traffic.py"), str(args.config)],
                               stdout=sys.stderr)
            try:
                with open(live) as f:
                    pmc, pmc_file = json.load(f)["kernels"], "measured by this run (--live-pmc: " + os.path.relpath(live, ROOT) + ")"
            except (OSError, KeyError, ValueError):
                log("PMC passes failed (rc %d); the committed summary is reported" % r.returncode)
        # 128x128 blocks on trailing matrices of >= 128 tiles, the capped 64-tile kernel below
        bulk_kernel = "k_update128<false>" if n >= 128 * 64 else "k_update2<true>"

        def pmc_entry(name):
            # exact kernel name, else the one instantiation that starts with it (template arguments
            # added since the pass was recorded)
            if name in pmc:
                return pmc[name]
            hits = [k for k in pmc if k.startswith(name.rstrip(">"))]
            return pmc[hits[0]] if len(hits) == 1 else None

        def pmc_iterations(tab):
            # the PMC passes profile one iteration: launches of the linearisation kernel = iterations covered
            for k, v in tab.items():
                if k.startswith("bae::k_linearize"):
                    return v.get("launches", 1)
            return 1

        def pmc_bytes(name):
            e = pmc_entry(name)
            return e["traffic_bytes_per_launch_corrected"] if e else None

        def pmc_raw(name):
            e = pmc_entry(name)
            return (e.get("FETCH_SIZE_KB_per_launch", 0.0) + e.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024.0 if e else None
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
            "config": {"workload": "BASELINE.json configs[%d]: %d poses / %d landmarks / %d reprojection residuals%s%s, "
                                   "%s, LmSize=%d, PoseSize=%d, %s%s"
                                   % (args.config, P, L, O, " + %d IMU pre-integration residuals" % (P - 1) if cfg["imu"] else "",
                                      " + unary priors every 100th pose + %d binary odometry constraints" % (P - 1) if cfg["priors"] else "",
                                      ("VARIANT: FOV camera w = %g" % args.fov_w) if args.fov_w else "pinhole",
                                      lm_dim, cfg["D"], "dogleg trust region" if cfg["dogleg"] else "Gauss-Newton (no dogleg)",
                                      "" if cfg["imu"] else ", 2 anchor poses inactive"),
                       "poses": P, "landmarks": L, "residuals": O, "reduced_system_n": n,
                       "driver": ("ba::BundleAdjuster::Solve(1) per step on a warm object (C++ API path, include/ba_capi.h)"
                                  if api_driver else "phase calls of the C-ABI (include/ba_hip.h)"),
                       "parallelism": ("landmark-sharded x%d (all poses on every rank%s), %s; collectives: %s"
                                       % (world, ", pose-pose residuals on rank 0" if api_driver else "",
                                          "replicated LDL^T (all-reduce of S)" if (os.environ.get("BA_BENCH_REPLICATED_SOLVE") or (api_driver and comm_mode != "native"))
                                          else "sparse exchange of S onto tile-block owners (shards along the trajectory), distributed LDL^T (square broadcast + point-to-point block rows, DESIGN.md 6a)",
                                          "engine-owned RCCL communicator" + (" (ba::BundleAdjuster::SetCommunicator)" if api_driver else "")
                                          if comm_mode == "native" else "torch.distributed hooks (%s)" % backend))
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": "%s (dense LDL^T trailing update, v_mfma_f64_16x16x4_f64; the look-ahead's bulk launches)" % bulk_kernel,
                         "achieved": syrk_tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": syrk_tf / FP64_MFMA_PEAK_TF, "traffic": pmc_bytes("bae::" + bulk_kernel),
                         "traffic_source": ("%s (rocprofv3 --pmc, bytes per launch)" % pmc_file) if pmc_file else None,
                         "flops_per_launch": ks.syrk_flops / max(ks.syrk_launches, 1),
                         "launches": ks.syrk_launches,
                         "avg_launch_us": 1e3 * ks.syrk_ms / max(ks.syrk_launches, 1),
                         # the WHOLE reduced solve (bulk updates + the chain they share the matrix pipes with): every
                         # tile product of the factorisation on the tile pattern over the `solve` phase of the last step
                         "solve_total": ({"flops": stats["factor_tile_products"] * 2.0 * 64 ** 3, "ms": timers.get("solve", 0.0),
                                          "achieved": stats["factor_tile_products"] * 2.0 * 64 ** 3 / (timers["solve"] * 1e-3) / 1e12,
                                          "frac": stats["factor_tile_products"] * 2.0 * 64 ** 3 / (timers["solve"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF}
                                         if stats and stats.get("factor_tile_products") and timers.get("solve", 0) > 0 else None)},
            "hbm_kernels": {
                # achieved_GBs: algorithmic bytes / live launch time; pmc_traffic_bytes: what the kernel
                # really moved, from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE: the guide's
                # gfx950 correction is calibrated for 16-B streaming reads and OVER-counts the 48-byte
                # row gathers of the assembly kernels — pmc_raw_bytes is FETCH + WRITE as reported)
                "k_linearize": {"achieved_GBs": lm_gbs, "frac_of_8TBs": lm_gbs / HBM_PEAK_GBS,
                                "avg_launch_us": 1e3 * ks.landmarks_ms / max(ks.landmarks_launches, 1),
                                "algorithmic_bytes": b_landmarks,
                                "pmc_traffic_bytes": pmc_bytes("bae::k_linearize<%d, 2, false, true>" % lm_dim),
                                "pmc_raw_bytes": pmc_raw("bae::k_linearize<%d, 2, false, true>" % lm_dim)},
                "k_assemble_tiles": {"achieved_GBs": ga_gbs, "frac_of_8TBs": ga_gbs / HBM_PEAK_GBS,
                                     "avg_launch_us": 1e3 * ks.gather_ms / max(ks.gather_launches, 1),
                                     "algorithmic_bytes": b_gather,
                                     "pmc_traffic_bytes": pmc_bytes("bae::k_assemble_tiles<5>"),
                                     "pmc_raw_bytes": pmc_raw("bae::k_assemble_tiles<5>")},
                "k_pose_blocks": {"achieved_GBs": po_gbs, "frac_of_8TBs": po_gbs / HBM_PEAK_GBS,
                                  "avg_launch_us": 1e3 * ks.pose_blocks_ms / max(ks.pose_blocks_launches, 1),
                                  "algorithmic_bytes": b_pose, "concurrent_with": "k_assemble_tiles (second stream)",
                                  "pmc_traffic_bytes": pmc_bytes("bae::k_pose_blocks"),
                                  "pmc_raw_bytes": pmc_raw("bae::k_pose_blocks")}},
            "phase_ms_last_step": {k: round(v, 4) for k, v in timers.items()},
            "structure": stats,
            # one-off per graph: PCIe uploads of the scene + host-side structure build (gather
            # lists, tile pattern); NOT part of `value` (inputs are resident when the timed region starts)
            "setup_s_rank0": round(t_setup, 3),
            "accepted_steps": accepted,
            "final_error": err,
        }
        if cfg["imu"] and ks.imu_launches:
            # per residual: two pose states and the samples in; 3 blocks of 15x15, two gradients, two
            # Jacobians, the information matrix out (one lane per residual: latency-bound, not HBM-bound)
            M = sc.imu_meas.shape[1]
            b_imu = (P - 1) * (2 * 16 * 8 + M * 56 + (3 * 225 + 30 + 2 * 225 + 225) * 8)
            us = 1e3 * ks.imu_ms / ks.imu_launches
            out["hbm_kernels"]["k_imu"] = {"avg_launch_us": us, "algorithmic_bytes": b_imu, "residuals": P - 1,
                                           "achieved_GBs": b_imu / (us * 1e-6) / 1e9,
                                           "frac_of_8TBs": b_imu / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                           "pmc_traffic_bytes": pmc_bytes("bae::k_imu")}
        # SURVEY.md §8d group figure for the Jacobian / Schur kernels (phases j_evaluation + robust_weights +
        # jtj_schur of the last step): COMPULSORY bytes — every input read once, every output written once, no
        # materialised Jacobians — over the time those phases took, against 8 TB/s; with and without the one
        # write of the upper triangle of S (14.4 GB at configs[3], the dominant term).  pmc_bytes: what the
        # group's kernels really moved (committed rocprofv3 --pmc summary), null if none.
        Dp = cfg["D"]
        b_lin = Ow * 32 + Lw * 36 + P * (56 + (72 if Dp == 15 else 0)) + Ow * 16
        b_out_noS = 8.0 * n + Lw * 8 * (ell * ell + ell)
        b_S = 8.0 * n * (n + 1) / 2
        g_ms = sum(timers.get(k, 0.0) for k in ("j_evaluation", "robust_weights", "jtj_schur"))
        g_pmc = None
        if pmc:
            # launches per iteration that belong to the group (k_residuals: the error pass of the median only — the
            # two evaluation passes of an iteration are the EvaluateResiduals phase)
            names = {"bae::k_linearize<%d, 2, false, true>" % lm_dim: 1, "bae::k_assemble_tiles<5>": 1, "bae::k_pose_blocks": 1,
                     "bae::k_residuals": 1, "bae::k_write_diag": 1, "bae::k_select_hist": None}
            tot = 0.0
            for nm, per_it in names.items():
                en = pmc_entry(nm)
                if en:
                    cnt = per_it if per_it is not None else en.get("launches", 1) / max(pmc_iterations(pmc), 1)
                    tot += en["traffic_bytes_per_launch_corrected"] * cnt
            g_pmc = tot if tot > 0 else None
        out["hbm_group"] = {
            "kernels": "j_evaluation + robust_weights + jtj_schur (k_residuals, k_select_*, k_linearize, k_assemble_tiles, k_pose_blocks, k_write_diag)",
            "bytes_8d": b_lin + b_out_noS + b_S, "bytes_8d_without_S": b_lin + b_out_noS,
            "ms": g_ms,
            "frac_with_S": (b_lin + b_out_noS + b_S) / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if g_ms > 0 else None,
            "frac_without_S": (b_lin + b_out_noS) / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if g_ms > 0 else None,
            "pmc_bytes": g_pmc,
            "pmc_frac": g_pmc / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if (g_pmc and g_ms > 0) else None}
        if api_ms is not None:
            # wall time of adjuster.Solve(1) on a warm object (graph unchanged since the last call)
            out["api_solve_ms"] = api_ms
            out["api_first_solve_s"] = round(first_solve_s, 3)
        for hk in out["hbm_kernels"].values():  # real (PMC) traffic over the live launch time
            if hk.get("pmc_traffic_bytes") and hk["avg_launch_us"] > 0:
                hk["pmc_GBs"] = hk["pmc_traffic_bytes"] / (hk["avg_launch_us"] * 1e-6) / 1e9
                hk["pmc_frac_of_8TBs"] = hk["pmc_GBs"] / HBM_PEAK_GBS
        if not args.no_cpu_baseline and world == 1:
            log("CPU baseline (oracle), 1 thread ...")
            out["cpu_baseline"] = cpu_baseline_record(cfg, args, n, O, 1)
            log("CPU baseline, %d threads ..." % usable_cpus())
            ncpu = usable_cpus()
            if ncpu > 1:
                out["cpu_baseline_all_cores"] = cpu_baseline_record(cfg, args, n, O, ncpu)
            try:
                with open("/proc/cpuinfo") as f:
                    model = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")]
                out["cpu_baseline"]["cpu_model"] = model[0] if model else ""
                out["cpu_baseline"]["nproc"] = ncpu
            except OSError:
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
