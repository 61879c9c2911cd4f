#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/r03_pytest_final.log 2>&1
rc=$?
tail -4 $OUT/r03_pytest_final.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python bench.py > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err || exit 1
tail -4 $OUT/r03_bench_default.err
timeout -k 10 400 python scratch/gpu_calib_parity_sweep.py 42 120 > $OUT/r03_calib_sweep.log 2>&1; echo "calib sweep rc $?"; tail -2 $OUT/r03_calib_sweep.log
