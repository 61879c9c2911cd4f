#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_distributed or rccl" > $OUT/r03_pytest_fullsize_dist.log 2>&1
rc=$?
tail -6 $OUT/r03_pytest_fullsize_dist.log
[ $rc -eq 0 ] || exit $rc
bash scratch/gpu_profile_round.sh bench 2 4 || exit 1
timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline > $OUT/r03_bench_cfg3_again.json 2> $OUT/r03_bench_cfg3_again.err || exit 1
tail -3 $OUT/r03_bench_cfg3_again.err
