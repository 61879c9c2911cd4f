#!/bin/bash
# per-block trace of k_update128 (measurement build -DBAE_TIME128), one iteration at configs[3], with and without the look-ahead
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for mode in lookahead alone; do
  if [ $mode = alone ]; then export BA_HIP_NO_LOOKAHEAD=1; fi
  BA_HIP_TRACE_FILE=/tmp/t128_$mode.bin BA_AMD_LIB=scratch/ab/time128/libba_hip.so timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-api --steps 1 --warmup 0 > $OUT/r03_time128_$mode.json 2> $OUT/r03_time128_$mode.err || { tail -3 $OUT/r03_time128_$mode.err; exit 1; }
  echo "== $mode"
  python3 scratch/analyze_t128.py /tmp/t128_$mode.bin > $OUT/r03_t128_trace_$mode.txt || exit 1
  tail -4 $OUT/r03_t128_trace_$mode.txt
done
