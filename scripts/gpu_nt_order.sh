# bench.py / kernel-level A/B of run-time switches of the shipped library (edit the lists): DPC_VIEW_ORDER, DPC_NT, DPC_NT_STORES
mkdir -p gpurun_out/r04x
for rep in 1 2 3; do
  for E in "DPC_NT_STORES=0" "DPC_NT_STORES=1" "DPC_NT_STORES=3" "DPC_NT_STORES=2"; do
    env $E timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $E rep $rep: %.4f ms median %.4f | %s' % (j['ms_per_step'], j['timing']['ms_per_step_median'], j['roofline']['kernel_ms_per_step']))" | tee -a gpurun_out/r04x/ab.txt
  done
done
