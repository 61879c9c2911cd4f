#!/bin/bash
# Round 3, VERDICT item 4: k_linearize beside k_imu.  A/B of the inertial kernels on the second stream
# (default) against the same kernels serialised on the main stream (BA_HIP_IMU_SERIAL=1), configs 2 and 4.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for c in 2 4; do
  timeout -k 10 300 python $ROOT/bench.py --config $c --no-cpu-baseline --steps 2 > $OUT/r03_imu_ab_cfg${c}_concurrent.json 2> $OUT/r03_imu_ab_cfg${c}_concurrent.err || exit 1
  BA_HIP_IMU_SERIAL=1 timeout -k 10 300 python $ROOT/bench.py --config $c --no-cpu-baseline --steps 2 > $OUT/r03_imu_ab_cfg${c}_serial.json 2> $OUT/r03_imu_ab_cfg${c}_serial.err || exit 1
done
python3 - <<PY
import json
for c in (2, 4):
    for v in ("concurrent", "serial"):
        d = json.load(open("$OUT/r03_imu_ab_cfg%d_%s.json" % (c, v)))
        hk = d["hbm_kernels"]
        print(c, v, "ms/step %.1f" % d["ms_per_step"], {k: round(x["avg_launch_us"]) for k, x in hk.items()},
              {k: d["phase_ms_last_step"][k] for k in ("j_evaluation", "jtj_schur") if k in d["phase_ms_last_step"]})
PY
