#!/bin/bash
# dev: variants of the recorded training step under a one-rank RCCL group, to locate the watchdog's hipErrorCapturedEvent
TAG=${1:-r05diag}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0 DPC_BENCH_TRACE=1
V() { NAME=$1; shift
  ( "$@" timeout 200 python bench.py --gpus 1 --force-dist --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline $EXTRA ) > "$OUT/$NAME.out" 2> "$OUT/$NAME.err"
  echo "$NAME rc=$? $(grep -c oneRank $OUT/$NAME.err) | $(grep -m1 -o 'HIP error: [a-z ]*' $OUT/$NAME.err) | last mark: $(grep '^\[bench\|^\[graphs' $OUT/$NAME.err | tail -1)"
}
V base env DPC_DIAG_HOOK=1
V base2 env
V nocache env TORCH_NCCL_CUDA_EVENT_CACHE=0
V sum env DPC_BUCKET_AVG=0
EXTRA="--burn-in 0" V noburn env
EXTRA="--batch 4" V batch4 env
# (a variant that forced gc.collect() before the capture ran here too: no difference; the hook is gone)
grep -h "GradBuckets\]" "$OUT/base.err" | sort | uniq -c | head -20
