#!/bin/bash
# one launch per tile column inside a sub-panel (k_step_fused) against the two-launch chain (BA_HIP_FUSED=0), alternating
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for round in 1 2 3; do
  for v in 0 1; do
    BA_HIP_FUSED=$v timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/r03_fused_${v}_$round.json 2> $OUT/r03_fused_${v}_$round.err || { tail -5 $OUT/r03_fused_${v}_$round.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/r03_fused_${v}_$round.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('cfg1 fused=$v round $round', round(d['ms_per_step'],3), 'solve', round(p['solve'],3), 'final', d.get('final_error'))"
  done
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "dense_cholesky or tile_sparse or reduced_system_and_step or distributed_solve or config1_scale or golden or indefinite or marginal or blocked_128" > $OUT/r03_fused_pytest.log 2>&1; rc=$?
tail -3 $OUT/r03_fused_pytest.log
exit $rc
