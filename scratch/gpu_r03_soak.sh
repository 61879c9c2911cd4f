#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 520 python scratch/gpu_random_parity_sweep.py 43 320 all fov > $OUT/r03_random_sweep_soak.log 2>&1; echo "random soak rc $?"; tail -2 $OUT/r03_random_sweep_soak.log
timeout -k 10 520 python scratch/gpu_calib_parity_sweep.py 44 240 > $OUT/r03_calib_sweep_soak.log 2>&1; echo "calib soak rc $?"; tail -1 $OUT/r03_calib_sweep_soak.log
