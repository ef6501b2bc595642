#!/bin/bash
# Round 6: the transform VJP with one work-group per view, gradient staged in LDS (k_points_bwd_view; DPC_PBWD_VIEW=0 / 1)
TAG=${1:-r06m}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
timeout 600 python -m pytest tests/test_round6_cases.py -x -q -m gpu -k "points_bwd" 2>&1 | tail -3 | tee -a "$OUT/ab.txt"
for SH in 320,8000,64,21,0.8 320,8000,64,9,0.8 128,8000,64,21,0.8 64,8000,64,21,0.8 512,8000,64,21,0.8 32,8000,128,11,1.6 128,8000,128,11,1.6 256,2000,64,11,0.8; do
  for F in 0 1; do
    echo "== $SH DPC_PBWD_VIEW=$F" | tee -a "$OUT/ab.txt"
    DPC_PBWD_VIEW=$F AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
