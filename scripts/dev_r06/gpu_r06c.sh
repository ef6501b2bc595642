#!/bin/bash
# Round 6: LDS tile size of k_splat_xy / k_gather_yx (DPC_FUSED_LDS_BYTES 48 K / 24 K / 14 K: strips of 64 / 32 / 16 rows at 128-wide)
TAG=${1:-r06c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 320,8000,64,21,0.8; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_lds24.so $C/libdpc_lds14.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
