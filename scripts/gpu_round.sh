#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats and
# PMC (HBM byte) passes.  Everything lands under gpurun_out/$TAG/.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r01a'
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
echo "== $(date) rocm-smi" | tee "$OUT/00_env.log"
(rocm-smi --showproductname 2>&1 | head -20; lscpu | head -20; nproc) >> "$OUT/00_env.log" 2>&1

echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/01_pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/01_pytest_gpu.log"
tail -5 "$OUT/01_pytest_gpu.log"

echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/02_smoke.log" 2>&1
echo "smoke exit $?" | tee -a "$OUT/02_smoke.log"
tail -2 "$OUT/02_smoke.log"

echo "== bench cfg2"
timeout 600 python bench.py --gpus 1 --steps 50 --warmup 10 > "$OUT/03_bench_cfg2.json" 2> "$OUT/03_bench_cfg2.err"
echo "bench exit $?"; cat "$OUT/03_bench_cfg2.json"
echo "== bench cfg5 / cfg1"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config 5 --no-cpu-baseline > "$OUT/04_bench_cfg5.json" 2> "$OUT/04_bench_cfg5.err"
cat "$OUT/04_bench_cfg5.json"
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --config 1 --no-cpu-baseline > "$OUT/04_bench_cfg1.json" 2> "$OUT/04_bench_cfg1.err"
cat "$OUT/04_bench_cfg1.json"

echo "== rocprofv3 kernel stats (cfg2)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats" -o cfg2 --output-format csv -- \
    python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/05_rocprof_stats.log" 2>&1
echo "rocprof stats exit $?"
find "$OUT/prof_stats" -name '*kernel_stats*.csv' | head -1 | xargs -r head -25

for CTR in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 pmc $CTR"
  timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/prof_pmc_$CTR" -o cfg2 --output-format csv -- \
      python "$REPO/bench.py" --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/06_rocprof_pmc_$CTR.log" 2>&1
  echo "pmc $CTR exit $?"
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/prof_calib_$CTR" -o calib --output-format csv -- \
      python "$REPO/scripts/pmc_calibrate.py" > "$OUT/07_calib_$CTR.log" 2>&1
  echo "calib $CTR exit $?"
done
cd "$REPO"
python scripts/summarize_profiles.py "$OUT" > "$OUT/08_summary.txt" 2>&1
cat "$OUT/08_summary.txt"
# keep the merge-back small: drop bulky raw traces, keep csv summaries
find "$OUT" -name '*.db' -delete 2>/dev/null
du -sh "$OUT"
