#!/bin/bash
# Round 6: does the allocator's segment policy change where the step lands (0.213 or 0.223 ms)?  fresh processes, default bench line
TAG=${1:-r06t}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for CONF in "" "expandable_segments:True" "" "expandable_segments:True" "" "expandable_segments:True" "" "expandable_segments:True"; do
  PYTORCH_CUDA_ALLOC_CONF=$CONF PYTORCH_HIP_ALLOC_CONF=$CONF timeout 300 python bench.py --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
  python -c "
import json; j=json.load(open('$OUT/b.json')); print('conf=[$CONF] value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" 2>&1 | tail -1 | tee -a "$OUT/alloc.txt"
  tail -2 "$OUT/b.err" | grep -i "warn\|error" | head -2 | tee -a "$OUT/alloc.txt"
done
