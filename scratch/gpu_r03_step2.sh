#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python scratch/gpu_r03_assemble.py 3 > $OUT/r03_assemble_cfg3.log 2>&1 || { tail -20 $OUT/r03_assemble_cfg3.log; exit 1; }
grep -E "variant|solve" $OUT/r03_assemble_cfg3.log
timeout -k 10 200 python scratch/gpu_r03_assemble.py 1 > $OUT/r03_assemble_cfg1.log 2>&1 || { tail -20 $OUT/r03_assemble_cfg1.log; exit 1; }
grep -E "variant|solve" $OUT/r03_assemble_cfg1.log
timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline --steps 2 > $OUT/r03_cfg4_imu_late.json 2> $OUT/r03_cfg4_imu_late.err || { tail -5 $OUT/r03_cfg4_imu_late.err; exit 1; }
python3 -c "
import json; d=json.load(open('$OUT/r03_cfg4_imu_late.json')); print('cfg4', d['ms_per_step'], {k: round(x['avg_launch_us']) for k,x in d['hbm_kernels'].items()}, d['phase_ms_last_step'])"
