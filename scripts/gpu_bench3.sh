#!/bin/bash
# bench.py lines for the headline config, the training step (configs[2]) and the projector alone at that shape
TAG=${1:-b3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"; echo "cfg2 rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --no-cpu-baseline > "$OUT/bench_cfg3_train.json" 2> "$OUT/bench_cfg3_train.err"; echo "cfg3 rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --keep-prob 0.07 --no-cpu-baseline > "$OUT/bench_cfg3_train_keep007.json" 2> "$OUT/bench_cfg3_train_keep007.err"; echo "cfg3 keep rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --projector-only --no-cpu-baseline > "$OUT/bench_cfg3_proj.json" 2> "$OUT/bench_cfg3_proj.err"; echo "cfg3p rc=$?"
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.load(open(f)); r=j["roofline"]; t=j["timing"]
        print(os.path.basename(f), "%.0f views/s %.3f ms/step | median %.3f p10 %.3f p90 %.3f (R=%d) | step_frac %.3f (%.3f ms) dom %s %.3f ms frac %.3f" % (
            j["value"], j["ms_per_step"], t["ms_per_step_median"], t["ms_per_step_p10"], t["ms_per_step_p90"], t["repeats"], r["step_frac"], r["step_ms"], r["kernel"], r["kernel_ms"], r["frac"]))
        print("    ", r["kernel_ms_per_step"])
        if "cpu_baseline" in j: print("    cpu:", {k:v for k,v in j["cpu_baseline"].items() if k!="sample"})
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
