"""Round 3: bytes of the sparse exchange of S at configs[3] on 4 thread-emulated ranks (one GPU: 4 x 28.8 GB of S),
landmark shards contiguous in id against shards dealt along the trajectory; one linearisation + distributed solve each,
the Gauss-Newton step compared across ranks.    python scratch/gpu_r03_sparse_scatter.py [nranks]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from ba_amd import hipapi, sharding

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = bench.CONFIGS[3]
P, L, K = cfg["P"], cfg["L"], 10
sc = bench.make_workload(cfg, P, L, K, 1)
print("scene ready", flush=True)
deals = {"id": [np.arange(lo, hi) for lo, hi in sharding.landmark_shards(np.full(L, K), N)],
         "trajectory": sharding.landmark_shards_along_trajectory(sc.lm_ref_pose, np.full(L, K), N)}
ref = None
for name in ("id", "trajectory"):
    engs = [bench.build_engine(sc, 1, 0, 0, 0, ids=deals[name][r])[0] for r in range(N)]
    print(name, "engines ready; local S tiles per shard:", [e.structure_stats()["tiles_S"] for e in engs], flush=True)
    ar = sharding.ThreadAllReduce(N)
    for r in range(N):
        engs[r].set_allreduce(ar.hook(r), r, N)
        engs[r].set_collectives(ar.collectives(r))
    out = {}

    def run(r):
        try:
            engs[r].linearize()
            rc = engs[r].solve_gn()
            out[r] = (rc, engs[r].get_delta_gn()[0])
        except Exception as exc:
            out[r] = exc
    t0 = time.time()
    th = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not ar.failed and all(not isinstance(out[r], Exception) and out[r][0] == 0 for r in range(N)), out
    assert all(np.array_equal(out[0][1], out[r][1]) for r in range(N))
    if ref is None:
        ref = out[0][1]
    cs = [e.comm_stats() for e in engs]
    print("%-10s S exchange: sent per rank %s GB, total %.2f GB; step vs first deal %.2e; chain recv max %.2f GB, side recv max %.2f GB (%.0f s)"
          % (name, ["%.2f" % (c["reduce_scatter_bytes"] / 1e9) for c in cs], sum(c["reduce_scatter_bytes"] for c in cs) / 1e9,
             float(np.linalg.norm(out[0][1] - ref) / np.linalg.norm(ref)), max(c["chain_bytes_recv"] for c in cs) / 1e9,
             max(c["side_bytes_recv"] for c in cs) / 1e9, time.time() - t0), flush=True)
    for e in engs:
        e.end_solve(); e.close()
