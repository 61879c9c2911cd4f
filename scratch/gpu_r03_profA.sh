#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
bash scratch/gpu_r03_step6.sh || exit 1
bash scratch/gpu_profile_round.sh bench 3 1 || exit 1
