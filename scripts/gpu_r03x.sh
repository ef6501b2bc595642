#!/bin/bash
# round-3 session X: refresh the training-step lines after the loss-select / sum_views touch-ups (+ the tests that cover them)
TAG=${1:-r03x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/01_pytest_gpu.log" 2>&1; echo "pytest exit $?"; tail -2 "$OUT/01_pytest_gpu.log"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --cpu-seconds 8 > "$OUT/05_bench_cfg3_train.json" 2> "$OUT/05_bench_cfg3_train.err"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --keep-prob 0.07 --no-cpu-baseline > "$OUT/05_bench_cfg3_train_keep007.json" 2> /dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --keep-prob 0.5 --no-cpu-baseline > "$OUT/05_bench_cfg3_train_keep05.json" 2> /dev/null
for KP in 1.0 0.5 0.07; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --graph --keep-prob $KP --no-cpu-baseline > "$OUT/05_bench_cfg3_train_graph_keep$KP.json" 2> /dev/null
done
timeout 300 python examples/chair_unsupervised/train_step.py --steps 40 --warmup 5 --keep-prob 0.07 --scheduled --max-steps 45 --graph > "$OUT/09_train_step_example_graph.json" 2> /dev/null
timeout 300 python examples/chair_unsupervised/train_step.py --steps 40 --warmup 5 --keep-prob 0.07 --scheduled --max-steps 45 > "$OUT/09_train_step_example_eager.json" 2> /dev/null
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "05_bench_*.json"))):
    try:
        j=json.load(open(f)); r=j["roofline"]; t=j["timing"]
        print("%-40s %.3f ms (median %.3f) library %.3f ms" % (os.path.basename(f), j["ms_per_step"], t["ms_per_step_median"], r["step_ms"]), r["kernel_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "09_*.json"))):
    print(os.path.basename(f), open(f).read()[:200])
PY
