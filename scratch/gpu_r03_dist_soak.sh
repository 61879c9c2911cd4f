#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -u scratch/gpu_dist_soak.py ${1:-51} ${2:-60} > $OUT/r03_dist_soak.log 2>&1; rc=$?
tail -6 $OUT/r03_dist_soak.log
exit $rc
