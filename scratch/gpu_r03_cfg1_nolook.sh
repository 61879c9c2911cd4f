#!/bin/bash
# configs[1]: the chain kernels with and without the bulk update beside them (BA_HIP_NO_LOOKAHEAD=1), kernel statistics
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for m in look nolook; do
  if [ $m = nolook ]; then export BA_HIP_NO_LOOKAHEAD=1; fi
  (cd $ROOT && timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/r03_c1_$m.json 2> $OUT/r03_c1_$m.err) || exit 1
  python3 -c "
import json; d=json.loads(open('$OUT/r03_c1_$m.json').read().strip().splitlines()[-1]); p=d['phase_ms_last_step']; print('$m', round(d['ms_per_step'],3), 'solve', round(p['solve'],3))"
  (cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/c1prof_$m && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/c1prof_$m -o f -- \
    python3 $ROOT/bench.py --config 1 --no-cpu-baseline --no-api --steps 20 > $OUT/c1prof_$m.json 2> $OUT/c1prof_$m.err) || exit 1
  db=$(ls $OUT/c1prof_$m/*.db 2>/dev/null | head -1); [ -n "$db" ] || exit 1
  python3 $ROOT/scratch/rocpd_stats.py $db > $OUT/r03_c1_${m}_kernel_stats.csv
  grep "k_step_update\|k_trsm_op\|k_update2\|k_backward2" $OUT/r03_c1_${m}_kernel_stats.csv | cut -d, -f1-4,6,7 | sed 's/_ZN3bae[0-9]*//; s/E[PvS].*\.kd"/"/'
done
