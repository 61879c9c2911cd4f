#!/bin/bash
# scratch: A/B a list of environment settings on the default bench (same box, back to back)
for e in "$@"; do
  env $e python bench.py --steps 3 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
  python - "$e" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("%-28s %.2f ms  solve %.1f  bulk %.1f TF  err %.10g" % (sys.argv[1], d["ms_per_step"], d["phase_ms_last_step"]["solve"], d["roofline"]["achieved"], d["final_error"]), flush=True)
PY
done
