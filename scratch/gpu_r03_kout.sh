#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for round in 1 2; do
for k in 16 20 24; do
  BA_HIP_KOUT=$k timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 4 > $OUT/r03_kout_${k}_$round.json 2> /dev/null || exit 1
  python3 -c "
import json; d=json.load(open('$OUT/r03_kout_${k}_$round.json')); print('KOUT $k round $round', round(d['ms_per_step'],1), 'ms  solve', round(d['phase_ms_last_step']['solve'],1), ' bulk', round(d['roofline']['achieved'],2), 'TF', d['roofline']['launches'], 'launches')"
done
done
