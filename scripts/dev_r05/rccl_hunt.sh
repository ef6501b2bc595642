#!/bin/bash
# the watchdog abort, hunted with numbers: many recordings per process, the remedy off, one thing varied at a time
TAG=${1:-r05hunt}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${RECORDS:-400}; P=${PROCS:-8}
run() { NAME=$1; shift
  died=0; rec=0
  for i in $(seq 1 $P); do
    timeout 200 python scripts/rccl_capture_stress.py --records $N "$@" > "$OUT/${NAME}_$i.out" 2> "$OUT/${NAME}_$i.err"
    if grep -q "^OK:" "$OUT/${NAME}_$i.out"; then rec=$((rec+N)); else died=$((died+1)); r=$(grep -c "stress\] recording" "$OUT/${NAME}_$i.err"); rec=$((rec+r)); fi
  done
  echo "$NAME: $died of $P processes died, ~$rec recordings | $*"
  for i in $(seq 1 $P); do grep -q "^OK:" "$OUT/${NAME}_$i.out" && rm -f "$OUT/${NAME}_$i.err"; done
}
run base      --drain 0
run nobarrier --drain 0 --no-barrier
run mainthr   --drain 0 --main-thread
run pause30   --drain 0 --pause-ms 30
N=100; P=3
run drain     --drain 0.25
