#!/bin/bash
# Round 3: the redesigned distributed solve on thread-emulated ranks + the RCCL single-rank test, then the
# configs[3] tile-pattern fixture.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "distributed or rccl or shards or sharding" > $OUT/r03_dist_pytest.log 2>&1
rc=$?
tail -25 $OUT/r03_dist_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/golden/make_config3_pattern.py > $OUT/r03_pattern.log 2>&1 || { tail -5 $OUT/r03_pattern.log; exit 1; }
tail -2 $OUT/r03_pattern.log
