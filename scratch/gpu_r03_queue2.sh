#!/bin/bash
# work-queue bulk launch without the look-ahead (bulk launches alone on the device): does the queue fill the CUs?
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
export BA_HIP_NO_LOOKAHEAD=1
for q in 0 1; do
  BA_HIP_BULK_QUEUE=$q timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 3 > $OUT/r03_queue_alone_$q.json 2> $OUT/r03_queue_alone_$q.err || { tail -5 $OUT/r03_queue_alone_$q.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/r03_queue_alone_$q.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('alone queue=$q', round(d['ms_per_step'],1), 'solve', round(p['solve'],1), 'bulk TF', round(d['roofline']['achieved'],2))"
  BA_HIP_BULK_QUEUE=$q BA_HIP_TRACE_FILE=/tmp/t128_aq$q.bin BA_AMD_LIB=scratch/ab/time128/libba_hip.so timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-api --steps 1 --warmup 0 > $OUT/r03_time128_aq$q.json 2> $OUT/r03_time128_aq$q.err || { tail -3 $OUT/r03_time128_aq$q.err; exit 1; }
  python3 scratch/analyze_t128.py /tmp/t128_aq$q.bin > $OUT/r03_t128_trace_alone_queue$q.txt || exit 1
  sed -n 3,6p $OUT/r03_t128_trace_alone_queue$q.txt; tail -2 $OUT/r03_t128_trace_alone_queue$q.txt
done
