#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -u scratch/gpu_dist_soak.py ${1:-51} ${2:-60} $3 > $OUT/r03_dist_soak_$3.log 2>&1; rc=$?
tail -6 $OUT/r03_dist_soak_$3.log
exit $rc
