mkdir -p gpurun_out/r04v
for rep in 1 2; do
for S in 3.0 1.0 0.8; do
  for E in "DPC_VIEW_ORDER=0" "DPC_VIEW_ORDER=5" "DPC_VIEW_ORDER=1"; do
    env $E timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --config 3 --projector-only --sigma $S --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sigma $S $E rep $rep: %.3f ms median %.3f | %s' % (j['ms_per_step'], j['timing']['ms_per_step_median'], j['roofline']['kernel_ms_per_step']))" | tee -a gpurun_out/r04v/ab.txt
  done
done
done
