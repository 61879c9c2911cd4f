#!/bin/bash
# scratch: sample package power / sclk while the default bench (configs[3]) iterates
python bench.py --steps 12 --warmup 1 --no-cpu-baseline > gpurun_out/pp.json 2> gpurun_out/pp.err &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*: //' | tr '\n' ' ' | awk '{ if ($NF+0 > 500) print }'
  sleep 0.4
done
python -c "
import json;d=json.loads(open('gpurun_out/pp.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])"
