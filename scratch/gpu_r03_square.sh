#!/bin/bash
# configs[1] with the per-column chain (BA_HIP_SQUARE=0) and with k_square / k_rowpanel, alternating; then the GPU suite
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for round in 1 2; do
  for v in 0 1; do
    BA_HIP_SQUARE=$v timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/r03_sq_${v}_$round.json 2> $OUT/r03_sq_${v}_$round.err || { tail -5 $OUT/r03_sq_${v}_$round.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/r03_sq_${v}_$round.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('square=$v round $round', round(d['ms_per_step'],3), 'solve', round(p['solve'],3))"
  done
done
if [ -n "$1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q $1 > $OUT/r03_sq_pytest.log 2>&1; rc=$?
  tail -5 $OUT/r03_sq_pytest.log
  exit $rc
fi
