#!/bin/bash
# scratch: A/B environment settings on the configs[1] bench (same box, back to back)
for e in "$@"; do
  env $e python bench.py --config 1 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab1.json 2> gpurun_out/ab1.err || exit 1
  python - "$e" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab1.json").read().strip().splitlines()[-1])
print("%-28s %.3f ms  solve %.3f  bulk %.1f TF  err %.10g" % (sys.argv[1], d["ms_per_step"], d["phase_ms_last_step"]["solve"], d["roofline"]["achieved"], d["final_error"]), flush=True)
PY
done
