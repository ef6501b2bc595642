#!/bin/bash
# A/B of library builds (csrc/libdpc_*.so, built with `make OUT=... EXTRA=...`) over the round's shapes; usage: gpu_ab4.sh TAG lib1.so lib2.so ...
TAG=${1:-ab4}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
for SH in ${AB_SHAPES:-32,8000,128,11,1.6 320,8000,64,21,3.0 320,8000,64,21,0.8 320,8000,64,21,0.3}; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
if [ -n "$AB_ENVS" ]; then     # run-time switches of the shipped library, e.g. AB_ENVS="DPC_ZSORT_SPLIT=1 DPC_SAVE_XY_MAXK=17"
  for E in $AB_ENVS; do
    for SH in ${AB_ENV_SHAPES:-32,8000,128,11,1.6}; do
      echo "== $SH with $E" | tee -a "$OUT/ab.txt"
      env $E AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py differentiable-point-clouds_amd/csrc/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
    done
  done
fi
