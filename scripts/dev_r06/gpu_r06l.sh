#!/bin/bash
# Round 6: k_zsort held to 64 VGPRs (two 1024-thread work-groups per compute unit) at the training shape (320 views: 1.25 work-groups per CU)
TAG=${1:-r06l}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
for SH in 320,8000,64,21,0.8 320,8000,64,9,0.8 256,8000,64,21,0.8 512,8000,64,21,0.8 32,8000,128,11,1.6; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_zs8.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
