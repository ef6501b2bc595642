#!/bin/bash
# Round-3 GPU session: parity tests, smoke, bench lines for every config, and per config (cfg2, the training-shape
# projector "cfg3p", cfg5) the rocprofv3 kernel statistics plus the FETCH_SIZE / WRITE_SIZE PMC passes.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_round3.sh r03x'       (SKIP_TESTS=1 / SKIP_PMC=1 to shorten)
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
(rocm-smi --showproductname 2>&1 | head -12; lscpu | head -16; nproc) > "$OUT/00_env.log" 2>&1
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/01_pytest_gpu.log" 2>&1
  echo "pytest exit $?" | tee -a "$OUT/01_pytest_gpu.log"; tail -3 "$OUT/01_pytest_gpu.log"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/02_smoke.log" 2>&1
  echo "smoke exit $?" | tee -a "$OUT/02_smoke.log"; tail -1 "$OUT/02_smoke.log"
fi
echo "== bench lines"
timeout 600 python bench.py --gpus 1 --steps 50 --warmup 10 > "$OUT/03_bench_cfg2.json" 2> "$OUT/03_bench_cfg2.err"; echo "cfg2 rc=$?"
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --config 5 --no-cpu-baseline > "$OUT/04_bench_cfg5.json" 2> "$OUT/04_bench_cfg5.err"
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --no-graph --no-cpu-baseline > "$OUT/04_bench_cfg2_eager.json" 2> "$OUT/04_bench_cfg2_eager.err"
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --config 1 --no-graph --no-cpu-baseline > "$OUT/04_bench_cfg1_eager.json" 2> "$OUT/04_bench_cfg1_eager.err"
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --config 1 --no-cpu-baseline > "$OUT/04_bench_cfg1.json" 2> "$OUT/04_bench_cfg1.err"
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --points ball --no-cpu-baseline > "$OUT/04_bench_cfg2_ball.json" 2> "$OUT/04_bench_cfg2_ball.err"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --cpu-seconds 8 > "$OUT/05_bench_cfg3_train.json" 2> "$OUT/05_bench_cfg3_train.err"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --keep-prob 0.07 --no-cpu-baseline > "$OUT/05_bench_cfg3_train_keep007.json" 2> "$OUT/05_bench_cfg3_train_keep007.err"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --keep-prob 0.5 --no-cpu-baseline > "$OUT/05_bench_cfg3_train_keep05.json" 2> "$OUT/05_bench_cfg3_train_keep05.err"
for KP in 1.0 0.5 0.07; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config 3 --graph --keep-prob $KP --no-cpu-baseline > "$OUT/05_bench_cfg3_train_graph_keep$KP.json" 2> "$OUT/05_bench_cfg3_train_graph_keep$KP.err"
done
timeout 300 python examples/chair_unsupervised/train_step.py --steps 40 --warmup 5 --keep-prob 0.07 --scheduled --max-steps 45 --graph > "$OUT/09_train_step_example_graph.json" 2> "$OUT/09_train_step_example_graph.err"
timeout 300 python examples/chair_unsupervised/train_step.py --steps 40 --warmup 5 --keep-prob 0.07 --scheduled --max-steps 45 > "$OUT/09_train_step_example_eager.json" 2> "$OUT/09_train_step_example_eager.err"
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --config 3 --projector-only --no-cpu-baseline > "$OUT/05_bench_cfg3_proj.json" 2> "$OUT/05_bench_cfg3_proj.err"
timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --config 3 --projector-only --num-points 560 --no-cpu-baseline > "$OUT/05_bench_cfg3_proj_n560.json" 2> "$OUT/05_bench_cfg3_proj_n560.err"
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "0[345]_bench_*.json"))):
    try:
        j=json.load(open(f)); r=j["roofline"]; t=j["timing"]
        print("%-36s %9.0f views/s %.3f ms/step | median %.3f p10 %.3f p90 %.3f (R=%d) | step_frac %.3f (%.3f ms) dom %s %.3f ms frac %.3f" % (
            os.path.basename(f), j["value"], j["ms_per_step"], t["ms_per_step_median"], t["ms_per_step_p10"], t["ms_per_step_p90"], t["repeats"], r["step_frac"], r["step_ms"], r["kernel"], r["kernel_ms"], r["frac"]))
        print("      ", r["kernel_ms_per_step"])
        if "cpu_baseline" in j: print("       cpu:", {k:v for k,v in j["cpu_baseline"].items() if k not in ("sample","cpu_model")})
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-800:])
PY
cd /tmp
prof() {  # name, bench args
  NAME=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats_$NAME" -o $NAME --output-format csv -- \
      python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --repeats 0 --no-graph --no-cpu-baseline "$@" > "$OUT/06_rocprof_stats_$NAME.log" 2>&1
  echo "rocprof stats $NAME exit $?"
  if [ -z "$SKIP_PMC" ]; then
    for CTR in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/prof_pmc_${CTR}_$NAME" -o $NAME --output-format csv -- \
          python "$REPO/bench.py" --gpus 1 --steps 5 --warmup 2 --repeats 0 --no-graph --no-cpu-baseline "$@" > "$OUT/07_rocprof_pmc_${CTR}_$NAME.log" 2>&1
      echo "pmc $CTR $NAME exit $?"
    done
  fi
}
prof cfg2
prof cfg3p --config 3 --projector-only
prof cfg5 --config 5
if [ -z "$SKIP_PMC" ]; then
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/prof_calib_$CTR" -o calib --output-format csv -- \
        python "$REPO/scripts/pmc_calibrate.py" > "$OUT/07_calib_$CTR.log" 2>&1
    echo "calib $CTR exit $?"
  done
fi
cd "$REPO"
if [ -z "$SKIP_SQ" ]; then
  BENCH_ARGS="--config 3 --projector-only --no-graph" bash scripts/pmc_sq.sh $TAG/sq_cfg3p > /dev/null 2>&1
  cp "$OUT/sq_cfg3p/sq_summary.txt" "$OUT/10_sq_counters_cfg3p.txt"
  BENCH_ARGS="--no-graph" bash scripts/pmc_sq.sh $TAG/sq_cfg2 > /dev/null 2>&1
  cp "$OUT/sq_cfg2/sq_summary.txt" "$OUT/10_sq_counters_cfg2.txt"
fi
python scripts/summarize_profiles2.py "$OUT" > "$OUT/08_summary.txt" 2>&1
tail -60 "$OUT/08_summary.txt"
find "$OUT" -name '*.db' -delete 2>/dev/null
du -sh "$OUT"
