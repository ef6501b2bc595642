#!/bin/bash
# Round 6: the tile plan of k_zbwd only (k_zfwd leaves the costs, atomics-free deal ahead of k_zbwd): plan cost and net effect (DPC_ZPERM=0 / 1), rocprof kernel table
TAG=${1:-r06k}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round6_cases.py -x -q -m gpu 2>&1 | tail -3 | tee -a "$OUT/ab.txt"
for Z in 0 1 0 1; do
  DPC_ZPERM=$Z timeout 300 python bench.py --no-cpu-baseline > "$OUT/bench_zperm$Z.json" 2> "$OUT/bench_zperm$Z.err"
  python -c "
import json; j=json.load(open('$OUT/bench_zperm$Z.json')); print('cfg2 DPC_ZPERM=$Z value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/ab.txt"
done
for Z in 0 1; do
  DPC_ZPERM=$Z timeout 300 python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench5_zperm$Z.json" 2> "$OUT/bench5_zperm$Z.err"
  python -c "
import json; j=json.load(open('$OUT/bench5_zperm$Z.json')); print('cfg5 DPC_ZPERM=$Z value %.0f ms_per_step %.4f median %.4f' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['roofline']['kernel_ms_per_step'])" | tee -a "$OUT/ab.txt"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats_cfg2" -o cfg2 --output-format csv -- \
   python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --repeats 0 --no-graph --no-cpu-baseline > "$OUT/06_rocprof_stats_cfg2.log" 2>&1
python - "$OUT" <<'PY'
import csv,glob,sys,os
for f in glob.glob(os.path.join(sys.argv[1],"prof_stats_cfg2","**","*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("%-70s calls %5s avg %9.1f ns  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
find "$OUT" -name '*.db' -delete 2>/dev/null
