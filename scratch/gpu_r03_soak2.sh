#!/bin/bash
# end-of-round parity soak on the final library: new seeds; then the same sweep through the opt-in one-workgroup square path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python -u scratch/gpu_random_parity_sweep.py 45 300 all fov > $OUT/r03_random_sweep_soak2.log 2>&1; rc1=$?; echo "random soak rc $rc1"; tail -2 $OUT/r03_random_sweep_soak2.log
timeout -k 10 400 python -u scratch/gpu_calib_parity_sweep.py 46 200 > $OUT/r03_calib_sweep_soak2.log 2>&1; rc2=$?; echo "calib soak rc $rc2"; tail -1 $OUT/r03_calib_sweep_soak2.log
BA_HIP_SQUARE=1 timeout -k 10 300 python -u scratch/gpu_random_parity_sweep.py 47 150 all fov > $OUT/r03_random_sweep_square.log 2>&1; rc3=$?; echo "square-path soak rc $rc3"; tail -2 $OUT/r03_random_sweep_square.log
[ $rc1 = 0 ] && [ $rc2 = 0 ] && [ $rc3 = 0 ]
