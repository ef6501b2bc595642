#!/bin/bash
# Round 6: the WALK instantiation of k_zbwd by rule (128-wide rows up), reload before the dead run; full -m gpu suite; small shapes against r05's numbers
TAG=${1:-r06f}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/01_pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/01_pytest_gpu.log"; tail -8 "$OUT/01_pytest_gpu.log"
DPC_ZWALK=1 DPC_ZBIG=1 timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -q -p no:cacheprovider > "$OUT/01_pytest_forced.log" 2>&1
echo "pytest (walk + big forced) exit $?" | tee -a "$OUT/01_pytest_forced.log"; tail -3 "$OUT/01_pytest_forced.log"
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 32,8000,64,11,1.6 4,1000,64,11,1.0 320,8000,64,21,0.8 320,8000,64,21,0.3; do
  for W in -1 1; do
    echo "== $SH DPC_ZWALK=$W" | tee -a "$OUT/ab.txt"
    if [ $W = -1 ]; then unset DPC_ZWALK; else export DPC_ZWALK=$W; fi
    AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_hip.so@walk0 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
unset DPC_ZWALK
