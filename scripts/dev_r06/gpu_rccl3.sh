#!/bin/bash
TAG=${1:-r06rccl3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
R() { NAME=$1; shift; timeout 400 python bench.py --gpus 1 --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/21_bench_$NAME.json" 2> "$OUT/21_bench_$NAME.err"; python -c "
import json; j=json.loads([l for l in open('$OUT/21_bench_$NAME.json') if l.startswith('{')][0]); print('$NAME: %.0f views/s %.3f ms/step (median %.3f)' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']))" | tee -a "$OUT/rccl.txt"; }
R plain
for MB in 64 128 256 64; do DPC_BUCKET_MB=$MB R copy_mb$MB --force-dist; done
DPC_BUCKET_AVG=0 DPC_BUCKET_MB=64 R copy_mb64_sum_then_divide --force-dist
R plain_again
