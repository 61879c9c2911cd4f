"""Micro-benchmark of the reduced-system assembly on the configs[3] (default) scene: one engine, the
linearisation phase repeated with every variant of k_assemble_tiles / tile order / tile set
(ba_hip_debug_set); per-kernel times from the engine's HIP-event profiling.
    python scratch/gpu_assemble_variants.py [config]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from ba_amd import sharding

cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = bench.CONFIGS[cfgid]
sc = bench.make_workload(cfg, cfg["P"], cfg["L"], 10, 1)
eng, _ = bench.build_engine(sc, 1, 0, cfg["L"], 0)
print("engine ready", eng.structure_stats(), flush=True)
eng.linearize(); eng.solve_gn()   # tile pattern of the factor
ref = None
for order in (0,):
    for alltiles in (0, 1):
        for var in (5,):
            eng.debug_set(4, alltiles)  # linearize variant rides on the same loop index
            eng.debug_set(1, var); eng.debug_set(2, order); eng.debug_set(3, 0)
            eng.linearize()
            eng.set_profiling(True)
            for _ in range(5):
                eng.linearize()
            ks = eng.kernel_stats()
            t = eng.get_timers()
            eng.set_profiling(False)
            rhs = eng.get_rhs()[0]
            if ref is None:
                ref = rhs
            same = bool(np.array_equal(ref, rhs))
            print("order %d all_tiles %d variant %d: assemble %.3f ms  pose_blocks %.3f ms  linearize %.3f ms  jtj_schur %.3f ms  rhs identical %s"
                  % (order, alltiles, var, ks.gather_ms / ks.gather_launches, ks.pose_blocks_ms / ks.pose_blocks_launches,
                     ks.landmarks_ms / ks.landmarks_launches, t["jtj_schur"], same), flush=True)
# correctness of the last variant through the factorisation
eng.debug_set(1, 5); eng.debug_set(2, 0); eng.debug_set(3, 0)
eng.linearize(); print("solve rc", eng.solve_gn())
