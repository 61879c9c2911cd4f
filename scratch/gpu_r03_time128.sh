#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
BA_AMD_LIB=scratch/ab/time128/libba_hip.so timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-api --steps 1 --warmup 0 > $OUT/r03_time128.json 2> $OUT/r03_time128.err
grep -h "^T128" $OUT/r03_time128.json $OUT/r03_time128.err | awk '{c[$3]++; p[$3]+=$5; l[$3]+=$7; e[$3]+=$9; pc[$3]+=$11} END {for (k in c) printf "cols %2d  n %5d  prologue %7.0f  loop %9.0f  epilogue %7.0f  per chunk %6.0f\n", k, c[k], p[k]/c[k], l[k]/c[k], e[k]/c[k], pc[k]/c[k]}' | sort -n -k2
