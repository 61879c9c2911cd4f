#!/bin/bash
# Round profile set (run through gpurun from the repo root): bench lines of the four BASELINE
# configs, rocprofv3 kernel statistics of the same commands, PMC traffic passes.
#   bash scratch/gpu_profile_round.sh bench|stats|pmc  [configs...]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mode=$1; shift
cfgs=${@:-3 1 2 4}
for c in $cfgs; do
  case $mode in
    bench)
      timeout -k 10 420 python $ROOT/bench.py --config $c > $OUT/r03_bench_cfg$c.json 2> $OUT/r03_bench_cfg$c.err || exit 1
      tail -2 $OUT/r03_bench_cfg$c.err ;;
    stats)
      (cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/prof_cfg$c && timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg$c -o cfg$c -- \
        python3 $ROOT/bench.py --config $c --no-cpu-baseline --no-api > $OUT/r03_prof_cfg$c.json 2> $OUT/r03_prof_cfg$c.err) || exit 1
      python3 $ROOT/scratch/rocpd_stats.py $(ls $OUT/prof_cfg$c/*.db | head -1) > $OUT/r03_bench_cfg${c}_kernel_stats.csv
      head -8 $OUT/r03_bench_cfg${c}_kernel_stats.csv ;;
    pmc)
      python3 $ROOT/scratch/pmc_traffic.py $c || exit 1 ;;
  esac
done
