#!/bin/bash
# same-box A/B of library builds at configs[3]: round-2 library, current (64-byte rows + fill-skip), current with packed rows;
# alternating, two rounds, so that box drift shows
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
  for v in r02 new p6 new_nofillskip; do
    case $v in
      r02) lib=scratch/ab/libba_hip_r02.so; dbg=;;
      new) lib=ba_amd/lib/libba_hip.so; dbg=;;
      p6) lib=scratch/ab/p6/libba_hip.so; dbg=;;
      new_nofillskip) lib=ba_amd/lib/libba_hip.so; dbg=1;;
    esac
    BA_AMD_LIB=$lib BA_BENCH_NO_FILL_SKIP=$dbg timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 4 > $OUT/r03_ab_${v}_$round.json 2> $OUT/r03_ab_${v}_$round.err || { tail -5 $OUT/r03_ab_${v}_$round.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$OUT/r03_ab_${v}_$round.json')); p=d['phase_ms_last_step']; print('$v $round', round(d['ms_per_step'],1), 'solve', round(p['solve'],1), 'j_eval', round(p['j_evaluation'],2), 'jtj', round(p['jtj_schur'],2), 'bulk TF', round(d['roofline']['achieved'],2), {k: round(x['avg_launch_us']) for k,x in d['hbm_kernels'].items()})"
  done
done
