#!/bin/bash
# Round 3 step 4: the whole GPU suite, then KOUT A/B at configs[3], then a 2-process gloo rehearsal of bench.py --gpus 2
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/r03_pytest_gpu.log 2>&1
rc=$?
tail -6 $OUT/r03_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for k in 16 24; do
  BA_HIP_KOUT=$k timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-api --steps 4 > $OUT/r03_kout$k.json 2> $OUT/r03_kout$k.err || { tail -5 $OUT/r03_kout$k.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/r03_kout$k.json')); print('KOUT $k', round(d['ms_per_step'],1), 'ms  bulk', round(d['roofline']['achieved'],2), 'TF', d['phase_ms_last_step'], d['hbm_group'])"
done
# 2 ranks on the one GPU, gloo control plane + hooks (host-staged): configs[1] through the C-ABI driver (distributed solve)
BA_BENCH_COMM=torch BA_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config 1 --steps 3 --warmup 1 > $OUT/r03_rehearsal_cfg1_2ranks.json 2> $OUT/r03_rehearsal_cfg1_2ranks.err || { tail -20 $OUT/r03_rehearsal_cfg1_2ranks.err; exit 1; }
tail -c 1500 $OUT/r03_rehearsal_cfg1_2ranks.json
