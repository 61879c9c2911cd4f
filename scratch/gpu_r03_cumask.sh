#!/bin/bash
# experiment: the bulk stream of a small system on a CU mask that leaves n CUs to the chain (BA_HIP_BULK_MASK=n)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for v in "0 1" "32 1" "64 1" "96 1" "64 4" "32 8" "0 1"; do
  set -- $v
  BA_HIP_BULK_MASK=$1 BA_HIP_BULK_MASK_STRIDE=$2 timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/r03_cumask.json 2> $OUT/r03_cumask.err || { tail -5 $OUT/r03_cumask.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/r03_cumask.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('reserved=$1 stride=$2', round(d['ms_per_step'],3), 'solve', round(p['solve'],3), 'final', d.get('final_error'))"
done
