#!/bin/bash
# build scratch/ab/<name>/libba_hip.so: k_chol.hip recompiled with extra flags, every other object reused
# usage: scratch/build_variant.sh <name> <extra hipcc flags ...>
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
out=$ROOT/scratch/ab/$name
mkdir -p $out
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $BASE "$@" -c $ROOT/ba_amd/csrc/k_chol.hip -o $out/k_chol.o
objs=""
for o in engine k_proj k_reduce k_posepose comm structure_dev; do objs="$objs $ROOT/ba_amd/lib/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libba_hip.so $objs $out/k_chol.o -ldl
echo built $out/libba_hip.so
