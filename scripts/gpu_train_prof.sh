#!/bin/bash
# rocprofv3 kernel trace of the chair_unsupervised training step (bench.py --config 3), eager; then the same step
# with MIOpen's benchmark (find) mode switched on/off as HIP-graph replays.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-tp}; mkdir -p $OUT
REPO=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o tr --output-format csv -- python $REPO/bench.py --config 3 --steps 30 --warmup 10 --repeats 0 --no-cpu-baseline > $OUT/prof.log 2>&1
echo "rocprof rc=$?"
cd $REPO
for m in 0 1; do
  DPC_CUDNN_BENCHMARK=$m timeout 300 python bench.py --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline > $OUT/graph_bm$m.json 2> $OUT/graph_bm$m.err
  DPC_CUDNN_BENCHMARK=$m timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/eager_bm$m.json 2> $OUT/eager_bm$m.err
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*_bm*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["value"]), "ms", round(j["ms_per_step"],3), "median", round(j["timing"]["ms_per_step_median"],3))
    except Exception as e: print(f,"ERR",e, open(f.replace(".json",".err")).read()[-500:])
PY
find $OUT -name '*.db' -delete
