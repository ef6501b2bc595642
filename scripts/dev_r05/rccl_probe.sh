#!/bin/bash
# dev: the one abort of the first RCCL session -- that build had no collective between init_process_group and the first
# bucket all-reduce (issued from the autograd thread on the warm-up's side stream); try that order again
TAG=${1:-r05probe}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0 DPC_BENCH_TRACE=1 DPC_WATCHDOG_DRAIN_S=0 DPC_INIT_BARRIER=0
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 200 python bench.py --gpus 1 --force-dist --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/nib_$i.out" 2> "$OUT/nib_$i.err"
  echo "no-init-barrier $i rc=$? | $(grep -m1 -o 'HIP error: [a-z ]*' $OUT/nib_$i.err) | $(grep '^\[bench\|^\[graphs' $OUT/nib_$i.err | tail -1)"
done
