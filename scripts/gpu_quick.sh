#!/bin/bash
# Short GPU session: parity tests + bench lines (no rocprof).  gpurun -- 'bash scripts/gpu_quick.sh TAG [extra bench args]'
TAG=${1:-quick}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/01_pytest_gpu.log" 2>&1
echo "pytest exit $?"; tail -3 "$OUT/01_pytest_gpu.log"
timeout 600 python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/03_bench_cfg2.json" 2> "$OUT/03_bench_cfg2.err"
python - "$OUT/03_bench_cfg2.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("cfg2: %.0f views/s  %.3f ms/step  step_frac %.3f  dom %s %.3f ms" % (j["value"], j["ms_per_step"], r["step_frac"], r["kernel"], r["kernel_ms"]))
print("   per-step ms:", r["kernel_ms_per_step"])
PY
for B in 8 16 64; do
  timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --batch $B > "$OUT/05_bench_cfg2_b$B.json" 2>> "$OUT/05.err"
  python - "$OUT/05_bench_cfg2_b$B.json" $B <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("cfg2 B=%s: %.0f views/s  %.3f ms/step  step_frac %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], r["step_frac"]), r["kernel_ms_per_step"])
PY
done
for C in 5 1; do
  timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --config $C > "$OUT/04_bench_cfg$C.json" 2>> "$OUT/04.err"
  python - "$OUT/04_bench_cfg$C.json" $C <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("cfg%s: %.0f views/s  %.3f ms/step  step_frac %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], r["step_frac"]), r["kernel_ms_per_step"])
PY
done
