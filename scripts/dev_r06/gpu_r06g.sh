#!/bin/bash
# Round 6: zfwd WALK instantiation by rule; forced walk + big work-groups on every shape that has them; then the evidence session
TAG=${1:-r06g}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
DPC_ZWALK=1 DPC_ZBIG=1 timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -q -p no:cacheprovider > "$OUT/01_pytest_forced.log" 2>&1
echo "pytest (walk + big forced) exit $?" | tee -a "$OUT/01_pytest_forced.log"; tail -3 "$OUT/01_pytest_forced.log"
DO="tests bench sigma generic prof pmc sq" bash scripts/gpu_round6.sh r06w
bash scripts/dev_r06/gpu_boxes.sh r06boxes2
