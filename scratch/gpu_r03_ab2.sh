#!/bin/bash
# same-box check that the round-3 library (tile-block ownership maps and row lists in the factorisation kernels) costs
# the single-GPU solve nothing against the round-2 library
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
  for v in r02 cur; do
    case $v in
      r02) lib=scratch/ab/libba_hip_r02.so;;
      cur) lib=ba_amd/lib/libba_hip.so;;
    esac
    BA_AMD_LIB=$lib timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 4 > $OUT/r03_ab2_${v}_$round.json 2> $OUT/r03_ab2_${v}_$round.err || { tail -5 $OUT/r03_ab2_${v}_$round.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$OUT/r03_ab2_${v}_$round.json')); p=d['phase_ms_last_step']; print('$v $round', round(d['ms_per_step'],1), 'solve', round(p['solve'],1), 'j_eval', round(p['j_evaluation'],2), 'jtj', round(p['jtj_schur'],2), 'bulk TF', round(d['roofline']['achieved'],2), {k: round(x['avg_launch_us']) for k,x in d['hbm_kernels'].items()})"
  done
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl or class_level_native" 2>&1 | tail -3
