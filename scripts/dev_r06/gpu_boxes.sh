#!/bin/bash
# the default bench line (graph replay) several times in fresh processes on one box: how stable is the headline?
TAG=${1:-r06boxes}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
(rocm-smi --showproductname 2>&1 | head -8; sha256sum differentiable-point-clouds_amd/csrc/libdpc_hip.so; git rev-parse HEAD 2>/dev/null) > "$OUT/env.log" 2>&1
for i in 1 2 3 4 5 6; do
  EXTRA=""; [ $((i % 2)) = 0 ] && EXTRA="--steps 20 --warmup 5"
  timeout 300 python bench.py --no-cpu-baseline $EXTRA > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"
  python -c "
import json; j=json.load(open('$OUT/bench_$i.json')); r=j['roofline']
print('run $i [$EXTRA] value %.0f ms_per_step %.4f median %.4f p10 %.4f p90 %.4f bound %s' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median'], j['timing']['ms_per_step_p10'], j['timing']['ms_per_step_p90'], r['bound']), r['kernel_ms_per_step'])" | tee -a "$OUT/boxes.txt"
done
for W in 0; do
  DPC_SPARSE_WALK=$W timeout 300 python bench.py --no-cpu-baseline > "$OUT/bench_walk$W.json" 2> "$OUT/bench_walk$W.err"
  python -c "
import json; j=json.load(open('$OUT/bench_walk$W.json')); print('DPC_SPARSE_WALK=$W value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/boxes.txt"
done
DPC_ZBIG=0 timeout 300 python bench.py --no-cpu-baseline > "$OUT/bench_zbig0.json" 2> "$OUT/bench_zbig0.err"
python -c "
import json; j=json.load(open('$OUT/bench_zbig0.json')); print('DPC_ZBIG=0 value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/boxes.txt"
DPC_ZDEAL=0 timeout 300 python bench.py --no-cpu-baseline > "$OUT/bench_zdeal0.json" 2> "$OUT/bench_zdeal0.err"
python -c "
import json; j=json.load(open('$OUT/bench_zdeal0.json')); print('DPC_ZDEAL=0 value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/boxes.txt"
for C in 5 5; do
  timeout 300 python bench.py --config $C --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg$C.json" 2> "$OUT/bench_cfg$C.err"
  python -c "
import json; j=json.load(open('$OUT/bench_cfg$C.json')); print('cfg$C value %.0f ms_per_step %.4f median %.4f bound %s' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median'], j['roofline']['bound']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/boxes.txt"
done
