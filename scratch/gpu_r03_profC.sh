#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
bash scratch/gpu_profile_round.sh stats 3 1 2 || exit 1
bash scratch/gpu_profile_round.sh pmc 3 1 || exit 1
timeout -k 10 500 python scratch/gpu_random_parity_sweep.py 41 200 all fov > $OUT/r03_random_sweep.log 2>&1; echo "sweep rc $?"; tail -3 $OUT/r03_random_sweep.log
