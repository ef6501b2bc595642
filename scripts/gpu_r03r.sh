#!/bin/bash
# round-3 session R: GPU tests of the fused candidate loss / in-kernel replication, and what they do to the step
TAG=${1:-r03r}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?"; tail -3 "$OUT/pytest_gpu.log"
for A in "" "--config 3 --projector-only" ; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $A 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print(j['config']['workload'][:60], '| %.3f ms (median %.3f)'%(j['ms_per_step'], j['timing']['ms_per_step_median']), r['kernel_ms_per_step'], 'ceiling %.0f GB/s (%s)'%(r['copy_ceiling']['GB/s'], r['copy_ceiling']['variant']))"
done
for KP in 1.0 0.07; do
  timeout 300 python bench.py --steps 20 --warmup 5 --config 3 --graph --keep-prob $KP --no-cpu-baseline 2>"$OUT/train_$KP.err" | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('train graph keep $KP | %.3f ms (median %.3f) library %.3f ms'%(j['ms_per_step'], j['timing']['ms_per_step_median'], r['step_ms']), r['kernel_ms_per_step'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 --config 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('train eager | %.3f ms (median %.3f) library %.3f ms'%(j['ms_per_step'], j['timing']['ms_per_step_median'], r['step_ms']))"
