#!/bin/bash
# dev session: G2 saved through the z-widened chunk mask (DPC_SAVE_G2_SPARSE=1, the new default) against the xy grid saved (=0)
TAG=${1:-r05f}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_chunk_sparse.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider \
   -k "chunk or goldens or knife or cfg2_full_batch or cfg5_full or degenerate or fused_dropout or fused_candidate_loss or training_shape_at or fused_l2 or asymmetric or fused_path_against or d256 or edge_planes or replay" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
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
  export DPC_SAVE_G2_SPARSE=$MODE
  B cfg2_g$MODE
  B cfg5_g$MODE --config 5
  B cfg2_k15_g$MODE --k 15 --sigma 2.5
  B cfg2_k21_g$MODE --k 21 --sigma 3.5
  for S in 0.3 0.8 1.0; do B cfg3p_s${S}_g$MODE --config 3 --projector-only --sigma $S; done
done
