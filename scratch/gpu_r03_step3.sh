#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SORT_LANDMARKS=1 timeout -k 10 400 python scratch/gpu_r03_assemble.py 3 > $OUT/r03_assemble_cfg3_sorted.log 2>&1 || { tail -20 $OUT/r03_assemble_cfg3_sorted.log; exit 1; }
grep -E "variant|solve|sorted" $OUT/r03_assemble_cfg3_sorted.log
