#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for round in 1 2; do
for v in base $VARIANTS; do
  if [ $v = base ]; then lib=ba_amd/lib/libba_hip.so; else lib=scratch/ab/$v/libba_hip.so; fi
  BA_BENCH_IGNORE_RC=1 BA_AMD_LIB=$lib timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 3 > $OUT/r03_var_${v}_$round.json 2> $OUT/r03_var_${v}_$round.err || { tail -3 $OUT/r03_var_${v}_$round.err; continue; }
  python3 -c "
import json; d=json.load(open('$OUT/r03_var_${v}_$round.json')); print('$v $round', round(d['ms_per_step'],1), 'solve', round(d['phase_ms_last_step']['solve'],1), 'bulk', round(d['roofline']['achieved'],2), 'err', d['final_error'])"
done
done
