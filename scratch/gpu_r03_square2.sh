#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
bash scratch/gpu_r03_square_time.sh || exit 1
bash scratch/gpu_r03_square.sh || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r03_sq_pytest.log 2>&1; rc=$?
tail -5 $OUT/r03_sq_pytest.log
exit $rc
