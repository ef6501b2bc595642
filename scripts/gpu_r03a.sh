#!/bin/bash
# round-3 session A: SQ counters of the shipped build at the training shape + per-phase ablation of splat / gather
TAG=${1:-r03a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
BENCH_ARGS="--config 3 --projector-only --no-graph" bash scripts/pmc_sq.sh $TAG/sq_cfg3p > /dev/null 2>&1
cp gpurun_out/$TAG/sq_cfg3p/sq_summary.txt gpurun_out/$TAG/sq_cfg3p_summary.txt
find gpurun_out/$TAG -name '*.db' -delete 2>/dev/null
bash scripts/gpu_ablate.sh $TAG/abl
