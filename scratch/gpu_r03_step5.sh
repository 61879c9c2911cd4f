#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "class_level or distributed_solve or rccl or structure_built" > $OUT/r03_pytest_class.log 2>&1
rc=$?
tail -6 $OUT/r03_pytest_class.log
[ $rc -eq 0 ] || exit $rc
# the C++-class driver of configs[2] on two ranks sharing the one GPU (gloo hooks, replicated solve), reduced size
BA_BENCH_COMM=torch BA_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --config 2 --poses 600 --landmarks 30000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/r03_rehearsal_cfg2_2ranks.json 2> $OUT/r03_rehearsal_cfg2_2ranks.err || { tail -20 $OUT/r03_rehearsal_cfg2_2ranks.err; exit 1; }
python3 -c "
import json; d=json.load(open('$OUT/r03_rehearsal_cfg2_2ranks.json')); print('rehearsal cfg2 x2', d['ms_per_step'], d['accepted_steps'], d['config']['parallelism'])"
timeout -k 10 300 python bench.py --config 2 --poses 600 --landmarks 30000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/r03_rehearsal_cfg2_1rank.json 2> $OUT/r03_rehearsal_cfg2_1rank.err || exit 1
python3 -c "
import json; a=json.load(open('$OUT/r03_rehearsal_cfg2_2ranks.json')); b=json.load(open('$OUT/r03_rehearsal_cfg2_1rank.json')); print('final error 2 ranks', a['final_error'], '1 rank', b['final_error'])"
for c in 2 4; do
  timeout -k 10 900 python3 scratch/pmc_traffic.py $c > $OUT/r03_pmc_cfg$c.log 2>&1 || { tail -10 $OUT/r03_pmc_cfg$c.log; exit 1; }
  tail -2 $OUT/r03_pmc_cfg$c.log
done
