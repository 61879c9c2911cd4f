#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
BA_BENCH_COMM=torch BA_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --config 1 --steps 3 --warmup 1 > $OUT/r03_rehearsal_cfg1_2ranks.json 2> $OUT/r03_rehearsal_cfg1_2ranks.err || { tail -20 $OUT/r03_rehearsal_cfg1_2ranks.err; exit 1; }
BA_BENCH_COMM=torch BA_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --config 2 --poses 600 --landmarks 30000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/r03_rehearsal_cfg2_2ranks.json 2> $OUT/r03_rehearsal_cfg2_2ranks.err || { tail -20 $OUT/r03_rehearsal_cfg2_2ranks.err; exit 1; }
timeout -k 10 300 python bench.py --config 1 --no-cpu-baseline --no-api --steps 3 --warmup 1 > $OUT/r03_rehearsal_cfg1_1rank.json 2> /dev/null || exit 1
python3 - <<PY
import json
def last(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
a = last("$OUT/r03_rehearsal_cfg1_2ranks.json"); b = last("$OUT/r03_rehearsal_cfg1_1rank.json"); c = last("$OUT/r03_rehearsal_cfg2_2ranks.json")
print("cfg1 2 ranks: final error", a["final_error"], "accepted", a["accepted_steps"], a["config"]["parallelism"])
print("cfg1 1 rank : final error", b["final_error"], "accepted", b["accepted_steps"])
print("cfg2 2 ranks (class driver): final error", c["final_error"], c["config"]["parallelism"])
PY
