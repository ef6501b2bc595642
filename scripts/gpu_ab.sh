#!/bin/bash
# A/B of library builds (per-kernel timings, interleaved) at cfg2, the training shape and cfg5, then the GPU parity tests.
# usage: gpu_ab.sh TAG lib1.so lib2.so ... ; set SKIP_TESTS=1 to skip pytest
TAG=${1:-ab}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
LIBS="$@"
[ -z "$LIBS" ] && LIBS=differentiable-point-clouds_amd/csrc/libdpc_hip.so
echo "== cfg2" | tee -a "$OUT/ab.txt"
timeout 300 python scripts/ab_libs.py $LIBS 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
for SH in 320,8000,64,21,3.0 320,560,64,21,3.0; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $LIBS 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
echo "== cfg5" | tee -a "$OUT/ab.txt"
AB_CONFIG=5 timeout 300 python scripts/ab_libs.py $LIBS 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?"; tail -15 "$OUT/pytest_gpu.log"
fi
