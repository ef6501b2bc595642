#!/bin/bash
TAG=${1:-r05h}
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
for MODE in 0 1; do
  export DPC_CHUNK_SPARSE_ON=$MODE
  for S in 1.2 1.5 3.0; do B cfg3p_s${S}_cs$MODE --config 3 --projector-only --sigma $S; done
  B d32_k11_cs$MODE --vox 32 --k 11 --sigma 1.6
  B d64_k15_b32_cs$MODE --vox 64 --k 15 --sigma 2.5
done
