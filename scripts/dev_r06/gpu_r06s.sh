#!/bin/bash
# Round 6: the -m gpu parity suites of the fused path under forced forms of the z kernels (the rule's forms ran in the evidence session)
TAG=${1:-r06s}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for ENV in "DPC_ZDEAL=2" "DPC_ZDEAL=2 DPC_ZBIG=1" "DPC_ZDEAL=0" "DPC_ZDEAL=2 DPC_SPARSE_WALK=2 DPC_ZBIG=1"; do
  echo "== $ENV" | tee -a "$OUT/forced.txt"
  env $ENV timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round6_cases.py tests/test_chunk_sparse.py tests/test_round4_cases.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee -a "$OUT/forced.txt"
done
