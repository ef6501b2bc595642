#!/bin/bash
# A/B of the order in which the kernels walk the views (DPC_VIEW_ORDER bits: 1 k_zfwd, 2 k_zbwd, 4 k_gather_yx, 8 k_splat_xy reversed):
# a consumer that starts with the views its producer wrote LAST finds them in the Infinity Cache.
TAG=${1:-r04r}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
for SH in ${AB_SHAPES:-32,8000,128,11,1.6 320,8000,64,21,3.0 320,8000,64,21,0.8 8,16000,256,11,2.0}; do
  for E in ${AB_ORDERS:-0 1 4 5 2 7 13}; do
    echo "== $SH DPC_VIEW_ORDER=$E" | tee -a "$OUT/ab.txt"
    DPC_VIEW_ORDER=$E AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py differentiable-point-clouds_amd/csrc/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
