#!/bin/bash
TAG=${1:-r05g}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
B() { NAME=$1; shift
  timeout 300 python bench.py --gpus 1 "$@" --no-cpu-baseline --steps 30 --warmup 5 > "$OUT/03_bench_$NAME.json" 2> "$OUT/03_bench_$NAME.err"
  python - "$OUT/03_bench_$NAME.json" "$NAME" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); r=j["roofline"]
    print("%-30s %.4f ms median | taps %s cs %s xy %s | %s" % (sys.argv[2], j["timing"]["ms_per_step_median"], j["config"].get("taps_run"), r.get("chunk_sparse"), r.get("saves_xy_grid"), r["kernel_ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
B cfg2
B cfg2_k21 --k 21 --sigma 3.5
B cfg5_k21 --config 5 --k 21 --sigma 3.5
B cfg3p_s3.0 --config 3 --projector-only --sigma 3.0
B cfg3p_s0.8 --config 3 --projector-only --sigma 0.8
B cfg5 --config 5
