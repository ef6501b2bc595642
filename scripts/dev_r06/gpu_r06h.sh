#!/bin/bash
# Round 6: widths that end inside a lane on the fused path; then the evidence session of the final build
TAG=${1:-r06h}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for V in 100 96 52 48 200 192 132; do
  timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --vox $V --no-cpu-baseline > "$OUT/bench_vox$V.json" 2> "$OUT/bench_vox$V.err"
  python -c "
import json; j=json.load(open('$OUT/bench_vox$V.json')); print('vox $V: %.0f views/s %.4f ms' % (j['value'], j['ms_per_step']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/vox.txt"
done
DPC_ZWALK=1 DPC_ZBIG=1 timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -q -p no:cacheprovider > "$OUT/01_pytest_forced.log" 2>&1
echo "pytest (walk + big forced) exit $?" | tee -a "$OUT/01_pytest_forced.log"; tail -3 "$OUT/01_pytest_forced.log"
DO="tests bench sigma generic prof pmc sq" bash scripts/gpu_round6.sh r06v
bash scripts/dev_r06/gpu_boxes.sh r06boxes3
