#!/bin/bash
# dev session: chunk-sparse rule (forced on / off at the tap counts above the round-4 rule), the new tap counts, SQ counters at cfg2
TAG=${1:-r05e}
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
    print("%-30s %.4f ms median | taps %s cs %s | %s" % (sys.argv[2], j["timing"]["ms_per_step_median"], j["config"].get("taps_run"), r.get("chunk_sparse"), r["kernel_ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
for MODE in 0 1; do
  export DPC_CHUNK_SPARSE_ON=$MODE
  for S in 1.0 1.2 1.5 3.0; do B cfg3p_s${S}_cs$MODE --config 3 --projector-only --sigma $S; done
  B cfg2_k15_cs$MODE --k 15 --sigma 2.5
  B cfg2_k21_cs$MODE --k 21 --sigma 3.5
  B cfg5_k21_cs$MODE --config 5 --k 21 --sigma 3.5
done
unset DPC_CHUNK_SPARSE_ON
B cfg2_k21 --k 21 --sigma 3.5
B cfg2_k23 --k 23 --sigma 4.0
B cfg2_k31 --k 31 --sigma 5.0
B cfg2_vox96 --vox 96
B cfg2_vox64 --vox 64
B cfg2 
BENCH_ARGS="--no-graph" bash scripts/pmc_sq.sh $TAG/sq_cfg2 > /dev/null 2>&1
cp "$OUT/sq_cfg2/sq_summary.txt" "$OUT/10_sq_counters_cfg2.txt"; rm -rf "$OUT/sq_cfg2"/pass*/
grep -A3 -E "^k_" "$OUT/10_sq_counters_cfg2.txt" | head -5
