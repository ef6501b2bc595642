#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-tp2}; mkdir -p $OUT
REPO=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o tr --output-format csv -- python $REPO/bench.py --config 3 --steps 30 --warmup 10 --repeats 0 --no-cpu-baseline > $OUT/prof.log 2>&1
echo "rocprof rc=$?"
find $OUT -name '*.db' -delete
