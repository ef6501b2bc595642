#!/bin/bash
# HIP-graph replay of the training step with the fused dropout's {keep, seed} in device memory.
OUT=gpurun_out/${1:-g}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "graph_replay or fused_dropout or training_step" > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
for kp in 1.0 0.5 0.07; do
  timeout 300 python bench.py --config 3 --graph --keep-prob $kp --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_graph_keep$kp.json 2> $OUT/bench_graph_keep$kp.err || tail -5 $OUT/bench_graph_keep$kp.err
done
timeout 300 python examples/chair_unsupervised/train_step.py --steps 40 --warmup 5 --keep-prob 0.07 --scheduled --max-steps 45 --graph > $OUT/example_graph.json 2> $OUT/example_graph.err || tail -5 $OUT/example_graph.err
timeout 300 python examples/chair_unsupervised/train_step.py --steps 40 --warmup 5 --keep-prob 0.07 --scheduled --max-steps 45 > $OUT/example_eager.json 2> $OUT/example_eager.err || tail -5 $OUT/example_eager.err
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j["value"],1), j["unit"], "ms/step", round(j["ms_per_step"],3), j.get("timing",{}).get("ms_per_step_median"), j.get("config",{}).get("loss"))
    except Exception as e: print(f, "ERR", e)
PY
