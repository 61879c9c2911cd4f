#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/fusedprof && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/fusedprof -o f -- \
  python3 $ROOT/bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/fusedprof.json 2> $OUT/fusedprof.err) || { tail -3 $OUT/fusedprof.err; exit 1; }
db=$(ls $OUT/fusedprof/*.db 2>/dev/null | head -1)
[ -n "$db" ] || { echo "no db"; exit 1; }
python3 $ROOT/scratch/rocpd_stats.py $db > $OUT/r03_fused_kernel_stats.csv
head -8 $OUT/r03_fused_kernel_stats.csv | cut -c1-150
