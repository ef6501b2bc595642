#!/bin/bash
# Round 6: the 16 tiles of a 1024-thread work-group re-dealt to its wavefronts by cost inside k_zbwd (zdeal_tiles; DPC_ZDEAL=0 / 1)
TAG=${1:-r06n}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
C=differentiable-point-clouds_amd/csrc
timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -x -q -m gpu 2>&1 | tail -3 | tee -a "$OUT/ab.txt"
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 32,8000,128,7,1.0 16,8000,256,7,1.5; do
  for F in 0 1; do
    echo "== $SH DPC_ZDEAL=$F" | tee -a "$OUT/ab.txt"
    DPC_ZDEAL=$F AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
for Z in 0 1 0 1 0 1; do
  DPC_ZDEAL=$Z timeout 300 python bench.py --no-cpu-baseline > "$OUT/bench_zdeal$Z.json" 2> "$OUT/bench_zdeal$Z.err"
  python -c "
import json; j=json.load(open('$OUT/bench_zdeal$Z.json')); print('cfg2 DPC_ZDEAL=$Z value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/ab.txt"
done
for Z in 0 1; do
  DPC_ZDEAL=$Z timeout 300 python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench5_zdeal$Z.json" 2> "$OUT/bench5_zdeal$Z.err"
  python -c "
import json; j=json.load(open('$OUT/bench5_zdeal$Z.json')); print('cfg5 DPC_ZDEAL=$Z value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/ab.txt"
done
