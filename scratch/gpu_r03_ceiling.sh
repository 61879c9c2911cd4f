#!/bin/bash
# where the bulk kernel's missing quarter goes: the same library with every operand chunk of k_update128 read from
# tile column 0 (cache-resident; wrong numbers) against the real one, same box
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for v in cur fake cur_nolook fake_nolook; do
  case $v in
    cur) lib=ba_amd/lib/libba_hip.so; nl=;;
    fake) lib=scratch/ab/fake/libba_hip.so; nl=;;
    cur_nolook) lib=ba_amd/lib/libba_hip.so; nl=1;;
    fake_nolook) lib=scratch/ab/fake/libba_hip.so; nl=1;;
  esac
  if [ -n "$nl" ]; then export BA_HIP_NO_LOOKAHEAD=1; else unset BA_HIP_NO_LOOKAHEAD; fi
  BA_BENCH_IGNORE_RC=1 BA_AMD_LIB=$lib timeout -k 10 200 python bench.py --config 3 --no-cpu-baseline --no-api --steps 3 > $OUT/r03_ceil_$v.json 2> $OUT/r03_ceil_$v.err || { tail -5 $OUT/r03_ceil_$v.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/r03_ceil_$v.json')); p=d['phase_ms_last_step']; print('$v', round(d['ms_per_step'],1), 'solve', round(p['solve'],1), 'bulk TF', round(d['roofline']['achieved'],2), 'avg launch us', round(d['roofline']['avg_launch_us']))"
done
