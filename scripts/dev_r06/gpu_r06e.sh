#!/bin/bash
# Round 6: the strips of a plane on one XCD (DPC_XCD_STRIPS) in k_splat_xy / k_gather_yx
TAG=${1:-r06e}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
DPC_XCD_STRIPS=1 timeout 600 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -x -q -p no:cacheprovider > "$OUT/01_pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/01_pytest.log"; tail -3 "$OUT/01_pytest.log"
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 320,8000,64,21,0.8 32,8000,128,21,3.5; do
  for F in 0 1; do
    echo "== $SH DPC_XCD_STRIPS=$F" | tee -a "$OUT/ab.txt"
    DPC_XCD_STRIPS=$F AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
