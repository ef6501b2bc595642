#!/bin/bash
# Round 6: the tile plan (DPC_ZPERM 0 = Latin-square deal, 1 / 2 = dealt by measured cost, the two wavefront -> SIMD hypotheses)
TAG=${1:-r06i}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
for Z in 1 2; do
  DPC_ZPERM=$Z timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -x -q -p no:cacheprovider > "$OUT/01_pytest_zperm$Z.log" 2>&1
  echo "pytest DPC_ZPERM=$Z exit $?" | tee -a "$OUT/01_pytest_zperm$Z.log"; tail -2 "$OUT/01_pytest_zperm$Z.log"
done
DPC_ZPERM=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "cfg2_full_batch or cfg5_full or goldens or fused_path_against or unnudged" > "$OUT/01_pytest_parity.log" 2>&1
echo "pytest parity exit $?" | tee -a "$OUT/01_pytest_parity.log"; tail -2 "$OUT/01_pytest_parity.log"
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 32,8000,128,7,1.0; do
  for Z in 0 1 2; do
    echo "== $SH DPC_ZPERM=$Z" | tee -a "$OUT/ab.txt"
    DPC_ZPERM=$Z AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
for Z in 0 1 2; do
  DPC_ZPERM=$Z timeout 300 python bench.py --no-cpu-baseline > "$OUT/bench_zperm$Z.json" 2> "$OUT/bench_zperm$Z.err"
  python -c "
import json; j=json.load(open('$OUT/bench_zperm$Z.json')); print('DPC_ZPERM=$Z value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/ab.txt"
done
