#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/r03_pytest_fill.log 2>&1
rc=$?
tail -6 $OUT/r03_pytest_fill.log
[ $rc -eq 0 ] || exit $rc
for c in 3 1; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-api > $OUT/r03_fill_cfg$c.json 2> $OUT/r03_fill_cfg$c.err || { tail -5 $OUT/r03_fill_cfg$c.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/r03_fill_cfg$c.json')); print('cfg $c', round(d['ms_per_step'],2), d['phase_ms_last_step'], {k: round(x['avg_launch_us']) for k,x in d['hbm_kernels'].items()}, round(d['hbm_group']['frac_with_S'],4))"
done
