#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
bash scratch/gpu_r03_square_time.sh || exit 1
for v in "0 4" "1 4" "1 2" "1 3" "1 1"; do
  set -- $v
  BA_HIP_SQUARE=$1 BA_HIP_SQ_W=$2 timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/r03_sq3.json 2> $OUT/r03_sq3.err || { tail -5 $OUT/r03_sq3.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/r03_sq3.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('square=$1 w=$2', round(d['ms_per_step'],3), 'solve', round(p['solve'],3))"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r03_sq_pytest.log 2>&1; rc=$?
tail -5 $OUT/r03_sq_pytest.log
exit $rc
