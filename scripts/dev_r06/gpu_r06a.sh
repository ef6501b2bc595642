#!/bin/bash
# Round 6, first GPU session: the sparse z walk.  Parity first, then per-kernel A/B (walk on / off / compiled out), bench lines, rocprof stats.
TAG=${1:-r06a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
C=differentiable-point-clouds_amd/csrc
(rocm-smi --showproductname 2>&1 | head -12; sha256sum $C/*.so) > "$OUT/00_env.log" 2>&1
timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -x -q -p no:cacheprovider > "$OUT/01_pytest_r06.log" 2>&1
echo "pytest r06 exit $?" | tee -a "$OUT/01_pytest_r06.log"; tail -5 "$OUT/01_pytest_r06.log"
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 320,8000,64,21,0.8 320,8000,64,21,0.3 32,8000,128,21,3.5 32,8000,128,23,4.0; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_hip.so@walk0 $C/libdpc_noskip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
B() { NAME=$1; shift; timeout 400 python bench.py --gpus 1 "$@" > "$OUT/03_bench_$NAME.json" 2> "$OUT/03_bench_$NAME.err"; echo "bench $NAME rc=$?"; tail -c 600 "$OUT/03_bench_$NAME.json"; echo; }
B cfg2 --steps 50 --warmup 10 --no-cpu-baseline
B cfg5 --steps 30 --warmup 5 --config 5 --no-cpu-baseline
if [ -z "$SKIP_FULL_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > "$OUT/02_pytest_gpu.log" 2>&1
  echo "pytest gpu exit $?" | tee -a "$OUT/02_pytest_gpu.log"; tail -6 "$OUT/02_pytest_gpu.log"
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats_cfg2" -o cfg2 --output-format csv -- \
   python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --repeats 0 --no-graph --no-cpu-baseline > "$OUT/06_rocprof_stats_cfg2.log" 2>&1
echo "rocprof exit $?"
python - "$OUT" <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"prof_stats_cfg2","**","*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("%-70s calls %5s avg %9.1f ns  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
find "$OUT" -name '*.db' -delete 2>/dev/null
du -sh "$OUT"
