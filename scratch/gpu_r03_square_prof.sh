#!/bin/bash
# rocprofv3 kernel statistics of configs[1] in square mode, widths 2 and 4
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for w in 2 4; do
  export BA_HIP_SQ_W=$w
  (cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/sqprof$w && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/sqprof$w -o sq -- \
    python3 $ROOT/bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/sqprof$w.json 2> $OUT/sqprof$w.err) || { tail -3 $OUT/sqprof$w.err; exit 1; }
  db=$(ls $OUT/sqprof$w/*.db 2>/dev/null | head -1)
  [ -n "$db" ] || { echo "no db for w=$w"; exit 1; }
  python3 $ROOT/scratch/rocpd_stats.py $db > $OUT/r03_sq_w${w}_kernel_stats.csv
  echo "w=$w"; head -7 $OUT/r03_sq_w${w}_kernel_stats.csv | cut -c1-140
done
