#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/r03_sq4.json 2> $OUT/r03_sq4.err || { tail -5 $OUT/r03_sq4.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/r03_sq4.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('default', round(d['ms_per_step'],3), 'solve', round(p['solve'],3))"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "one_workgroup_square or dense_cholesky or blocked_128 or distributed_solve_matches_single" > $OUT/r03_sq_pytest.log 2>&1; rc=$?
tail -5 $OUT/r03_sq_pytest.log
exit $rc
