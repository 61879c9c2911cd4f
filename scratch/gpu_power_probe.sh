#!/bin/bash
# scratch: sample power / sclk while the standalone update kernel runs on random vs zero data
for z in 0 1; do
  if [ $z = 1 ]; then export ZERO=1; else unset ZERO; fi
  REPS=500 ./scratch/upd_bench_new 520 8 1 &
  pid=$!
  sleep 1.5
  for k in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo; sleep 0.6; done
  wait $pid
done
