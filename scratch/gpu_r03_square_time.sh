#!/bin/bash
# cycle stamps of k_square's phases (measurement build -DBAE_TIME_SQ) at configs[1]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
BA_AMD_LIB=scratch/ab/tsq/libba_hip.so timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 3 --warmup 1 > $OUT/r03_tsq.log 2>&1 || { tail -5 $OUT/r03_tsq.log; exit 1; }
grep TSQ $OUT/r03_tsq.log | tail -4
