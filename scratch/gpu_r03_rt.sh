#!/bin/bash
# round-trip trimming of the small-system kernels: library at HEAD (scratch/ab/base) against the working tree, alternating
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for round in 1 2 3; do
  for v in base new; do
    lib=ba_amd/lib/libba_hip.so; [ $v = base ] && lib=scratch/ab/base/libba_hip.so
    BA_AMD_LIB=$lib timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/r03_rt_${v}_$round.json 2> $OUT/r03_rt_${v}_$round.err || { tail -5 $OUT/r03_rt_${v}_$round.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/r03_rt_${v}_$round.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('cfg1 $v round $round', round(d['ms_per_step'],3), 'solve', round(p['solve'],3), 'final', d.get('final_error'))"
  done
done
for v in base new; do
  lib=ba_amd/lib/libba_hip.so; [ $v = base ] && lib=scratch/ab/base/libba_hip.so
  BA_AMD_LIB=$lib timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 4 > $OUT/r03_rt3_${v}.json 2> $OUT/r03_rt3_${v}.err || { tail -5 $OUT/r03_rt3_${v}.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/r03_rt3_${v}.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('cfg3 $v', round(d['ms_per_step'],2), 'solve', round(p['solve'],2), 'final', d.get('final_error'))"
done
