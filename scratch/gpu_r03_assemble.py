"""Round 3: tile assembly variants 5 (round 2) and 6 (tile descriptors, four terms per round, 2-wave workgroups)
on the configs[3] / configs[1] scene; per-kernel times from the engine's HIP-event profiling, S compared bitwise
through a checksum of the packed lower storage (ba_hip_device_buffer 0 -> torch view).
    python scratch/gpu_r03_assemble.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = bench.CONFIGS[cfgid]
sc = bench.make_workload(cfg, cfg["P"], cfg["L"], 10, 1)
if os.environ.get("SORT_LANDMARKS"):
    # experiment: landmark ids in the order of their reference pose (what a tracker that creates landmarks as it
    # moves produces) instead of the generator's spatially random ids: locality of the row gathers
    L, ns = sc.num_landmarks, sc.obs_per_landmark + 1
    perm = np.argsort(sc.lm_ref_pose, kind="stable")
    sc.landmarks = sc.landmarks[perm]; sc.lm_ref_pose = sc.lm_ref_pose[perm]
    sc.obs_pose = sc.obs_pose.reshape(L, ns)[perm].reshape(-1)
    sc.obs_z = sc.obs_z.reshape(L, ns, 2)[perm].reshape(-1, 2)
    print("landmarks sorted by reference pose", flush=True)
eng, _ = bench.build_engine(sc, 1, 0, cfg["L"], 0)
print("engine ready", eng.structure_stats(), flush=True)
eng.linearize(); eng.solve_gn()   # tile pattern of the factor
import torch
from ba_amd.sharding import _DevArray
import ctypes as C
ptr, cnt = C.c_void_p(), C.c_size_t()
eng._chk(eng.L.ba_hip_device_buffer(eng.h, 0, C.byref(ptr), C.byref(cnt)))
A = torch.as_tensor(_DevArray(ptr.value, cnt.value, "<i8"), device="cuda")
ref = None
for var in (5, 6, 5, 6):
    eng.debug_set(1, var)
    eng.linearize()
    torch.cuda.synchronize()
    chk = int(A.sum().item())    # integer sum of the bit patterns: any differing bit shows
    eng.set_profiling(True)
    for _ in range(6):
        eng.linearize()
    ks = eng.kernel_stats()
    t = eng.get_timers()
    eng.set_profiling(False)
    if ref is None:
        ref = chk
    print("variant %d: assemble %.3f ms  pose_blocks %.3f ms  linearize %.3f ms  j_evaluation %.3f  robust %.3f  jtj_schur %.3f ms  S bitwise identical %s"
          % (var, ks.gather_ms / ks.gather_launches, ks.pose_blocks_ms / ks.pose_blocks_launches,
             ks.landmarks_ms / ks.landmarks_launches, t["j_evaluation"], t["robust_weights"], t["jtj_schur"], chk == ref), flush=True)
eng.debug_set(1, 6)
eng.linearize(); print("solve rc", eng.solve_gn())
