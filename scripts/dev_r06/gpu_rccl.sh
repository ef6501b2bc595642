#!/bin/bash
# Round 6: GradBuckets gather="copy" against "accumulate" under a forced one-rank RCCL group (recorded training step)
TAG=${1:-r06rccl}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "one_rank_rccl or rccl_watchdog" > "$OUT/20_pytest_rccl.log" 2>&1
echo "pytest rccl exit $?" | tee -a "$OUT/20_pytest_rccl.log"; tail -4 "$OUT/20_pytest_rccl.log"
R() { NAME=$1; shift; timeout 400 python bench.py --gpus 1 --config 3 --graph --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/21_bench_$NAME.json" 2> "$OUT/21_bench_$NAME.err"; echo "bench $NAME rc=$?"; python -c "
import json; j=json.load(open('$OUT/21_bench_$NAME.json')); print('$NAME: %.0f views/s %.3f ms/step (median %.3f)' % (j['value'], j['ms_per_step'], j['timing']['ms_per_step_median']), j['config']['parallelism'][:90])" | tee -a "$OUT/rccl.txt"; }
R graph_plain
R graph_rccl1_copy --force-dist
DPC_BUCKET_GATHER=accumulate R graph_rccl1_accumulate --force-dist
R graph_rccl1_copy_again --force-dist
R graph_plain_again
(cd /tmp
 timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_rccl_graph" -o rccl_graph --output-format csv -- \
    python "$REPO/bench.py" --gpus 1 --force-dist --config 3 --graph --steps 5 --warmup 2 --repeats 0 --burn-in 0 --no-cpu-baseline > "$OUT/22_rocprof_rccl_graph.log" 2>&1
 echo "rocprof rccl graph exit $?")
python scripts/summarize_rccl.py "$OUT" > "$OUT/23_rccl_world1.txt" 2>&1; tail -40 "$OUT/23_rccl_world1.txt"
find "$OUT" -name '*.db' -delete 2>/dev/null
