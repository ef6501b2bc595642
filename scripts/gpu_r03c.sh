#!/bin/bash
# round-3 session C: which of the two splat changes regressed it (A/B of variant builds), LDS conflict counters
TAG=${1:-r03c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
C=differentiable-point-clouds_amd/csrc
LIBS="$C/libdpc_base_r02.so $C/libdpc_hip.so $C/libdpc_v_nofold.so $C/libdpc_v_nou16.so $C/libdpc_v_neither.so"
for SH in 320,8000,64,21,3.0 320,560,64,21,3.0; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $LIBS 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL \
   --kernel-trace -d "$OUT/lds" -o p --output-format csv -- \
   python "$REPO/bench.py" --gpus 1 --steps 4 --warmup 2 --repeats 0 --no-cpu-baseline --config 3 --projector-only --no-graph > "$OUT/lds.log" 2>&1
echo "lds pass exit $?"
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "lds/**/*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")[:44]
        if k.startswith("k_"):
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(os.path.join(out, "lds_summary.txt"), "w") as fh:
    for k in sorted(agg):
        fh.write(k + "\n")
        for c in sorted(agg[k]):
            v = sorted(agg[k][c])
            fh.write("   %-32s median %16.0f  (n=%d)\n" % (c, v[len(v) // 2], len(v)))
print(open(os.path.join(out, "lds_summary.txt")).read())
PY
find "$OUT" -name '*.db' -delete 2>/dev/null
