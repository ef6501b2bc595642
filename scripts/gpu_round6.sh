#!/bin/bash
# Round-6 GPU session (round 5's, plus: issue.json from the SQ passes -- bench.py's roofline.issue / bound).  Sections (space-separated in DO): rccl tests bench sigma generic prof pmc sq
#   gpurun --timeout 1500 -- 'DO="rccl tests" bash scripts/gpu_round6.sh r05a'
# `pmc` regenerates gpurun_out/<tag>/traffic.json WITH the stamp (sha256 of the library and of the kernel sources it was
# measured on); copy it to profiles/traffic.json unedited -- bench.py quotes it only for that build.
TAG=${1:-r06}
DO=${DO:-"rccl tests bench prof pmc sq"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
has() { [[ " $DO " == *" $1 "* ]]; }
(rocm-smi --showproductname 2>&1 | head -12; lscpu | head -16; nproc; sha256sum differentiable-point-clouds_amd/csrc/libdpc_hip.so) > "$OUT/00_env.log" 2>&1
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > "$OUT/01_pytest_gpu.log" 2>&1
  RC=$?; echo "pytest exit $RC" | tee -a "$OUT/01_pytest_gpu.log"; tail -15 "$OUT/01_pytest_gpu.log"
  if [ $RC -ne 0 ] && [ -n "$STOP_ON_FAIL" ]; then echo "tests failed: stopping (STOP_ON_FAIL)"; exit 1; fi
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/02_smoke.log" 2>&1
  echo "smoke exit $?" | tee -a "$OUT/02_smoke.log"; tail -1 "$OUT/02_smoke.log"
fi
if has rccl; then
  # RCCL on the one GPU: a forced ONE-rank process group (backend nccl).  Tests first, then the kernel trace of the
  # recorded training step (GradBuckets' all-reduces inside the HIP graph) and of the eager DDP step.
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "one_rank_rccl or rccl_watchdog" > "$OUT/20_pytest_rccl.log" 2>&1
  echo "pytest rccl exit $?" | tee -a "$OUT/20_pytest_rccl.log"; tail -5 "$OUT/20_pytest_rccl.log"
  timeout 300 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/21_bench_cfg2_rccl1.json" 2> "$OUT/21_bench_cfg2_rccl1.err"; echo "bench cfg2 rccl1 rc=$?"
  timeout 400 python bench.py --gpus 1 --force-dist --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/21_bench_cfg3_graph_rccl1.json" 2> "$OUT/21_bench_cfg3_graph_rccl1.err"; echo "bench cfg3 graph rccl1 rc=$?"
  timeout 400 python bench.py --gpus 1 --force-dist --config 3 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/21_bench_cfg3_ddp_rccl1.json" 2> "$OUT/21_bench_cfg3_ddp_rccl1.err"; echo "bench cfg3 ddp rccl1 rc=$?"
  timeout 400 python bench.py --gpus 1 --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/21_bench_cfg3_graph_plain.json" 2> "$OUT/21_bench_cfg3_graph_plain.err"; echo "bench cfg3 graph plain rc=$?"
  (cd /tmp
   timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_rccl_graph" -o rccl_graph --output-format csv -- \
      python "$REPO/bench.py" --gpus 1 --force-dist --config 3 --graph --steps 5 --warmup 2 --repeats 0 --burn-in 0 --no-cpu-baseline > "$OUT/22_rocprof_rccl_graph.log" 2>&1
   echo "rocprof rccl graph exit $?"
   timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_rccl_ddp" -o rccl_ddp --output-format csv -- \
      python "$REPO/bench.py" --gpus 1 --force-dist --config 3 --steps 5 --warmup 2 --repeats 0 --burn-in 0 --no-cpu-baseline > "$OUT/22_rocprof_rccl_ddp.log" 2>&1
   echo "rocprof rccl ddp exit $?")
  # the capture / watchdog race: with the drain every recording goes through; without it the process is expected to die
  timeout 300 python scripts/rccl_capture_stress.py --records 40 > "$OUT/24_stress_drain.log" 2>&1; echo "stress with drain rc=$?"
  for i in 1 2 3; do
    timeout 300 python scripts/rccl_capture_stress.py --records 40 --drain 0 > "$OUT/24_stress_nodrain_$i.log" 2>&1; echo "stress WITHOUT drain, run $i rc=$? (non-zero expected)"
  done
  python scripts/summarize_rccl.py "$OUT" > "$OUT/23_rccl_world1.txt" 2>&1; tail -60 "$OUT/23_rccl_world1.txt"
fi
B() {  # name, bench args...
  NAME=$1; shift
  timeout 400 python bench.py --gpus 1 "$@" > "$OUT/03_bench_$NAME.json" 2> "$OUT/03_bench_$NAME.err"; echo "bench $NAME rc=$?"
}
if has ab; then
  # chunk-sparse layout forced off against the per-shape rule (the default)
  timeout 600 python -m pytest tests/test_chunk_sparse.py -m gpu -q -p no:cacheprovider > "$OUT/30_pytest_chunk_sparse.log" 2>&1
  echo "pytest chunk-sparse exit $?" | tee -a "$OUT/30_pytest_chunk_sparse.log"; tail -3 "$OUT/30_pytest_chunk_sparse.log"
  for MODE in 0 rule; do
    if [ $MODE = 0 ]; then export DPC_CHUNK_SPARSE_ON=0; else unset DPC_CHUNK_SPARSE_ON; fi
    B cfg2_cs$MODE --steps 50 --warmup 10 --no-cpu-baseline
    B cfg5_cs$MODE --steps 30 --warmup 5 --config 5 --no-cpu-baseline
    for S in 3.0 0.8 0.3; do
      B cfg3p_sigma${S}_cs$MODE --steps 30 --warmup 5 --config 3 --projector-only --sigma $S --no-cpu-baseline
    done
  done
  unset DPC_CHUNK_SPARSE_ON
fi
if has bench; then
  B cfg2 --steps 50 --warmup 10
  B cfg2_driver --steps 20 --warmup 5 --no-cpu-baseline
  B cfg5 --steps 30 --warmup 5 --config 5 --no-cpu-baseline
  B cfg1 --steps 50 --warmup 10 --config 1 --no-cpu-baseline
  B cfg2_eager --steps 50 --warmup 10 --no-graph --no-cpu-baseline
  B cfg3_train --steps 20 --warmup 5 --config 3 --no-cpu-baseline
  B cfg3_train_graph --steps 20 --warmup 5 --config 3 --graph --no-cpu-baseline
  timeout 300 python scripts/bench_extras.py 2>/dev/null | tail -1 > "$OUT/09_bench_extras.json"; echo "extras rc=$?"
fi
if has sigma; then
  for S in 3.0 1.5 0.8 0.3; do
    B cfg3p_sigma$S --steps 30 --warmup 5 --config 3 --projector-only --sigma $S --no-cpu-baseline
  done
  B cfg3p_n560 --steps 30 --warmup 5 --config 3 --projector-only --num-points 560 --no-cpu-baseline
fi
if has generic; then
  B cfg2_k15 --steps 30 --warmup 5 --k 15 --sigma 2.5 --no-cpu-baseline
  B cfg2_k21 --steps 30 --warmup 5 --k 21 --sigma 3.5 --no-cpu-baseline
  B cfg2_k23 --steps 30 --warmup 5 --k 23 --sigma 4.0 --no-cpu-baseline
  B cfg2_k31 --steps 30 --warmup 5 --k 31 --sigma 5.0 --no-cpu-baseline
  B cfg2_vox48 --steps 30 --warmup 5 --vox 48 --no-cpu-baseline
  B cfg2_vox96 --steps 30 --warmup 5 --vox 96 --no-cpu-baseline
  B cfg2_vox100 --steps 30 --warmup 5 --vox 100 --no-cpu-baseline
  B cfg2_vox64 --steps 30 --warmup 5 --vox 64 --no-cpu-baseline
fi
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "03_bench_*.json"))):
    try:
        j=json.load(open(f)); r=j["roofline"]; t=j["timing"]
        print("%-40s %9.0f views/s %.3f ms/step | median %.3f p10 %.3f p90 %.3f | taps %s | dom %s %.3f ms frac %.3f vs_ceil %.2f | step_frac %.3f vs_ceil %.2f" % (
            os.path.basename(f)[9:-5], j["value"], j["ms_per_step"], t["ms_per_step_median"], t["ms_per_step_p10"], t["ms_per_step_p90"],
            j["config"].get("taps_run"), r["kernel"], r["kernel_ms"], r["frac"], r["vs_ceiling"] or 0, r["step_frac"], r["step_vs_ceiling"] or 0))
        print("      ", r["kernel_ms_per_step"])
        if "cpu_baseline" in j: print("       cpu:", {k:v for k,v in j["cpu_baseline"].items() if k not in ("sample","cpu_model")})
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-800:])
PY
cd /tmp
stats() {  # name, bench args
  NAME=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats_$NAME" -o $NAME --output-format csv -- \
      python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --repeats 0 --no-graph --no-cpu-baseline "$@" > "$OUT/06_rocprof_stats_$NAME.log" 2>&1
  echo "rocprof stats $NAME exit $?"
}
pmc() {
  NAME=$1; shift
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/prof_pmc_${CTR}_$NAME" -o $NAME --output-format csv -- \
        python "$REPO/bench.py" --gpus 1 --steps 5 --warmup 2 --repeats 0 --no-graph --no-cpu-baseline "$@" > "$OUT/07_rocprof_pmc_${CTR}_$NAME.log" 2>&1
    echo "pmc $CTR $NAME exit $?"
  done
}
if has prof; then
  stats cfg2
  stats cfg3p --config 3 --projector-only
  stats cfg3p_sigma0.8 --config 3 --projector-only --sigma 0.8
  stats cfg5 --config 5
fi
if has pmc; then
  pmc cfg2
  pmc cfg3p --config 3 --projector-only
  pmc cfg3p_sigma0.8 --config 3 --projector-only --sigma 0.8
  pmc cfg5 --config 5
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d "$OUT/prof_calib_$CTR" -o calib --output-format csv -- \
        python "$REPO/scripts/pmc_calibrate.py" > "$OUT/07_calib_$CTR.log" 2>&1
    echo "calib $CTR exit $?"
  done
fi
cd "$REPO"
if has sq; then
  # SQ / GRBM counters per kernel -> issue.json (stamped with the library's sha256; copy to profiles/issue.json unedited)
  BENCH_ARGS="--no-graph" bash scripts/pmc_sq.sh $TAG/sq_cfg2 > /dev/null 2>&1
  cp "$OUT/sq_cfg2/sq_summary.txt" "$OUT/10_sq_counters_cfg2.txt"
  BENCH_ARGS="--config 5 --no-graph" bash scripts/pmc_sq.sh $TAG/sq_cfg5 > /dev/null 2>&1
  cp "$OUT/sq_cfg5/sq_summary.txt" "$OUT/10_sq_counters_cfg5.txt"
  BENCH_ARGS="--config 3 --projector-only --no-graph" bash scripts/pmc_sq.sh $TAG/sq_cfg3p > /dev/null 2>&1
  cp "$OUT/sq_cfg3p/sq_summary.txt" "$OUT/10_sq_counters_cfg3p.txt"
  python scripts/summarize_sq.py "$OUT" cfg2=config2:32:8000 cfg5=config5:8:16000 cfg3p=config3:320:8000 > "$OUT/11_issue.txt" 2>&1
  cat "$OUT/11_issue.txt"
fi
if has prof || has pmc; then
  python scripts/summarize_profiles4.py "$OUT" --traffic cfg2=config2:32:8000 cfg3p=config3:320:8000 \
      cfg3p_sigma0.8=config3_sigma0.8:320:8000 cfg5=config5:8:16000 > "$OUT/08_summary.txt" 2>&1
  tail -70 "$OUT/08_summary.txt"
fi
find "$OUT" -name '*.db' -delete 2>/dev/null
du -sh "$OUT"
