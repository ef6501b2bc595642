#!/bin/bash
# Round 6: the next plane of a slot put in flight BEHIND the FIR push that consumed the old one (dense streaks as loops of their own):
# the rolling prefetch of the walking z kernels is no longer drained at the top of every group; against the build before (libdpc_prev.so), interleaved rounds in one process
TAG=${1:-r06v2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -x -q -m gpu 2>&1 | tail -2 | tee -a "$OUT/ab.txt"
for REP in 1 2; do
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 32,8000,128,15,2.4 320,8000,64,21,0.8 32,8000,128,7,1.0 4,1000,64,11,1.0; do
  echo "== $SH (process $REP)" | tee -a "$OUT/ab.txt"
  AB_ROUNDS=7 AB_STEPS=40 AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_prev.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
done
