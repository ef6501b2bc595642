#!/bin/bash
# re-take the rocprofv3 kernel statistics of bench.py as committed (default --burn-in) into an existing session directory
TAG=${1:-r03z}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
prof() {
  NAME=$1; shift
  rm -rf "$OUT/prof_stats_$NAME"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats_$NAME" -o $NAME --output-format csv -- \
      python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --repeats 0 --no-graph --no-cpu-baseline "$@" > "$OUT/06_rocprof_stats_$NAME.log" 2>&1
  echo "rocprof stats $NAME exit $?"
  find "$OUT/prof_stats_$NAME" -name '*.db' -delete 2>/dev/null
  python - "$OUT/prof_stats_$NAME" <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
agg = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v.sort(); print("   %-60s calls %5d avg %8.2f us median %8.2f us" % (k, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3))
PY
}
prof cfg2
prof cfg3p --config 3 --projector-only
prof cfg5 --config 5
