#!/bin/bash
# training step: fused Adam / MIOpen find mode on and off, eager and as one HIP graph
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-to}; mkdir -p $OUT
for f in 0 1; do for m in 0 1; do
  DPC_ADAM_FUSED=$f DPC_CUDNN_BENCHMARK=$m timeout 300 python bench.py --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline > $OUT/graph_f${f}_bm$m.json 2> $OUT/graph_f${f}_bm$m.err
done; done
timeout 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/eager_default.json 2> $OUT/eager_default.err
timeout 300 python bench.py --config 3 --graph --steps 20 --warmup 5 --keep-prob 0.07 --no-cpu-baseline > $OUT/graph_default_keep007.json 2> $OUT/graph_default_keep007.err
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "training_step or graph_replay" 2>&1 | tail -2
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["value"]), "ms", round(j["ms_per_step"],3), "median", round(j["timing"]["ms_per_step_median"],3))
    except Exception as e: print(f,"ERR",e, open(f.replace(".json",".err")).read()[-500:])
PY
