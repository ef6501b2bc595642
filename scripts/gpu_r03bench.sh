#!/bin/bash
# refresh the projector bench lines (bench.py as committed: with the default --burn-in) -> gpurun_out/TAG
TAG=${1:-r03bench}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
timeout 600 python bench.py --gpus 1 --steps 50 --warmup 10 > "$OUT/03_bench_cfg2.json" 2> "$OUT/03_bench_cfg2.err"; echo "cfg2 rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/03_bench_cfg2_steps20.json" 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --config 5 --no-cpu-baseline > "$OUT/04_bench_cfg5.json" 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --no-graph --no-cpu-baseline > "$OUT/04_bench_cfg2_eager.json" 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --config 1 --no-graph --no-cpu-baseline > "$OUT/04_bench_cfg1_eager.json" 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --config 1 --no-cpu-baseline > "$OUT/04_bench_cfg1.json" 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --points ball --no-cpu-baseline > "$OUT/04_bench_cfg2_ball.json" 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --config 3 --projector-only --no-cpu-baseline > "$OUT/05_bench_cfg3_proj.json" 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --config 3 --projector-only --num-points 560 --no-cpu-baseline > "$OUT/05_bench_cfg3_proj_n560.json" 2>/dev/null
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "0[345]_bench_*.json"))):
    try:
        j=json.load(open(f)); r=j["roofline"]; t=j["timing"]
        print("%-34s %9.0f views/s %.4f ms/step | median %.4f | frac %.3f of_ceiling %.2f step_measured/ceiling %.2f ceiling %.0f" % (
            os.path.basename(f), j["value"], j["ms_per_step"], t["ms_per_step_median"], r["frac"], r["frac_of_ceiling"],
            (r["step_measured_achieved"] or 0)/r["copy_ceiling"]["GB/s"], r["copy_ceiling"]["GB/s"]))
    except Exception as e:
        print(f, "ERR", e)
PY
