#!/bin/bash
# Round 6: GradBuckets under a one-rank RCCL group -- where the averaging happens (average="auto": SUM at one rank; DPC_BUCKET_AVG=1: RCCL's AVG; =0: sum, then divide)
TAG=${1:-r06rccl4}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccl or one_rank or world1 or force_dist" -p no:cacheprovider 2>&1 | tail -3 | tee -a "$OUT/rccl.txt"
R() { NAME=$1; shift; timeout 400 python bench.py --gpus 1 --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/21_bench_$NAME.json" 2> "$OUT/21_bench_$NAME.err"; python -c "
import json; j=json.loads([l for l in open('$OUT/21_bench_$NAME.json') if l.startswith('{')][0]); print('$NAME: %.0f views/s %.3f ms/step (median %.3f)' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']))" | tee -a "$OUT/rccl.txt"; }
R plain
R rccl1_auto --force-dist
DPC_BUCKET_AVG=1 R rccl1_collective_avg --force-dist
DPC_BUCKET_AVG=0 R rccl1_divide --force-dist
R rccl1_auto_again --force-dist
R plain_again
