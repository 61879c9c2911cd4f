#!/bin/bash
# final verification of the round: GPU suite, smoke, default bench line, bench + kernel statistics of configs 3 and 1
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
bash scratch/gpu_profile_round.sh bench 3 1 || exit 1
bash scratch/gpu_profile_round.sh stats 3 1 || exit 1
