"""FP64 GFLOP/s of the oracle's dense LDL^T (the stand-in for the reference's Eigen LDLT /
SimplicialLDLT, BundleAdjuster.cpp:752-799) at n = 6k, 12k, 24k on the host cores of the GPU box,
1 thread (reference-faithful) and all cores (best-effort): the three calibration points
BASELINE.md §3 asks for.  bench.py extrapolates the CPU baseline's solve term with them.

    python scratch/cpu_ldlt_fit.py > gpurun_out/cpu_ldlt_fit.json   (then copy to profiles/r02_cpu_ldlt_fit.json)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402

po.build()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
ncpu = bench.usable_cpus()
model = ""
with open("/proc/cpuinfo") as f:
    for ln in f:
        if ln.startswith("model name"):
            model = ln.split(":", 1)[1].strip()
            break
out = {"cpu_model": model, "nproc": ncpu, "threads_1": [], "all_cores": []}
sizes = [int(a) for a in sys.argv[1:]] or [6000, 12000, 24000]
rng = np.random.default_rng(0)
for n in sizes:
    # SPD, diagonally dominant: the factorisation does the same flops whatever the values
    a = rng.random((n, n))
    a = np.triu(a) + np.diag(np.full(n, float(n)))
    b = rng.random(n)
    for key, th in (("all_cores", ncpu), ("threads_1", 1)):
        po.set_num_threads(th)
        t0 = time.time()
        x = po.dense_solve_upper(a, b)
        dt = time.time() - t0
        gf = (n ** 3 / 3.0 + 2.0 * n * n) / dt / 1e9
        out[key].append({"n": n, "threads": th, "seconds": round(dt, 3), "gflops": round(gf, 3)})
        print("n=%d threads=%d %.1f s %.2f GFLOP/s" % (n, th, dt, gf), file=sys.stderr, flush=True)
    del a
po.set_num_threads(1)
print(json.dumps(out, indent=1))
