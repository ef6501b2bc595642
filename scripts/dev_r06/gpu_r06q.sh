#!/bin/bash
# Round 6: placement probe -- does cfg2's process-to-process spread follow the arena's address?
TAG=${1:-r06q}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for REP in 1 2 3; do
  echo "== process $REP" | tee -a "$OUT/probe.txt"
  DPC_BINDING=ctypes timeout 300 python scripts/dev_r06/placement_probe.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/probe.txt"
done
