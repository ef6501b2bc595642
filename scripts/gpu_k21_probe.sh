#!/bin/bash
# Per-kernel timings of the training-shape projector (B=320, 64^3, K=21) at the dropout sweep's point counts.
TAG=${1:-k21}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
LIB=differentiable-point-clouds_amd/csrc/libdpc_hip.so
for SH in 320,8000,64,21,3.0 320,4000,64,21,3.0 320,560,64,21,3.0 320,8000,64,21,0.5; do
  echo "== $SH"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $LIB 2>&1 | tail -2 | tee -a "$OUT/ab_k21.txt"
done
cd /tmp; export TMPDIR=/tmp
AB_SHAPE=320,8000,64,21,3.0 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o k21 --output-format csv -- \
   python "$REPO/scripts/ab_libs.py" "$REPO/$LIB" > "$OUT/rocprof.log" 2>&1
find "$OUT/prof" -name '*kernel_stats*.csv' | head -1 | xargs -r head -14
find "$OUT" -name '*.db' -delete 2>/dev/null
