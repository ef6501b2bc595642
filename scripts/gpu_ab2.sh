#!/bin/bash
# quick A/B of library builds at the training shape (N = 8000 and 560); usage: gpu_ab2.sh TAG lib1.so lib2.so ...
TAG=${1:-ab2}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
for SH in ${AB_SHAPES:-320,8000,64,21,3.0 320,560,64,21,3.0}; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
