#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
PMC_KERNEL_REGEX="k_linearize|k_assemble_tiles|k_pose_blocks|k_imu|k_residuals|k_select_hist|k_write_diag" timeout -k 10 700 python3 scratch/pmc_traffic.py 4 2>&1 | tee $OUT/r03_pmc_cfg4.log | grep -E "running|pass|linearize|assemble|imu|pose_blocks"
