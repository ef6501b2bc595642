#!/bin/bash
# Round 6: the in-work-group tile deal against a build without it (-DDPC_ZDEAL_DEFAULT=0), interleaved rounds in ONE process per shape
TAG=${1:-r06p2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
for REP in 1 2 3; do
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0; do
  echo "== $SH (process $REP)" | tee -a "$OUT/ab.txt"
  AB_ROUNDS=9 AB_STEPS=40 AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_nodeal.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
done
