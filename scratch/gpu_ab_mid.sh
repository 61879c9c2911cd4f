#!/bin/bash
# scratch: A/B environment settings on a mid-size scene (3000 / 5000 poses)
P=$1; L=$2; shift 2
for e in "$@"; do
  env $e python bench.py --config 1 --poses $P --landmarks $L --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/abm.json 2> gpurun_out/abm.err || exit 1
  python - "$e" <<PY
import json, sys
d = json.loads(open("gpurun_out/abm.json").read().strip().splitlines()[-1])
print("%-32s n=%d  %.2f ms  solve %.2f  bulk %.1f TF" % (sys.argv[1], d["config"]["reduced_system_n"], d["ms_per_step"], d["phase_ms_last_step"]["solve"], d["roofline"]["achieved"]), flush=True)
PY
done
