#!/bin/bash
# the bulk launches as work queues (k_update128q) against plain grids, alternating on one box; then the per-block trace
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for round in 1 2; do
  for q in 0 1; do
    BA_HIP_BULK_QUEUE=$q timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 4 > $OUT/r03_queue_${q}_$round.json 2> $OUT/r03_queue_${q}_$round.err || { tail -5 $OUT/r03_queue_${q}_$round.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/r03_queue_${q}_$round.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('queue=$q round $round', round(d['ms_per_step'],1), 'solve', round(p['solve'],1), 'bulk TF', round(d['roofline']['achieved'],2), 'final', d.get('final_error', d.get('error'))) "
  done
done
for q in 1; do
  BA_HIP_BULK_QUEUE=$q BA_HIP_TRACE_FILE=/tmp/t128_q$q.bin BA_AMD_LIB=scratch/ab/time128/libba_hip.so timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-api --steps 1 --warmup 0 > $OUT/r03_time128_q$q.json 2> $OUT/r03_time128_q$q.err || { tail -3 $OUT/r03_time128_q$q.err; exit 1; }
  python3 scratch/analyze_t128.py /tmp/t128_q$q.bin > $OUT/r03_t128_trace_queue$q.txt || exit 1
  tail -4 $OUT/r03_t128_trace_queue$q.txt
done
