#!/bin/bash
# SQ / GRBM counter passes over the cfg2 bench (one rocprofv3 run per pass; PMC only with --kernel-trace)
TAG=${1:-pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
i=0
for PASS in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM" \
  "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PASS --kernel-trace -d "$OUT/pass$i" -o p --output-format csv -- \
      python "$REPO/bench.py" --gpus 1 --steps 4 --warmup 2 --repeats 0 --no-cpu-baseline $BENCH_ARGS > "$OUT/pass$i.log" 2>&1
  echo "pass $i exit $?"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*/**/*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")[:44]
        if k.startswith("k_"):
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
with open(os.path.join(out, "sq_summary.txt"), "w") as fh:
    for k in sorted(agg):
        fh.write(k + "\n")
        for c in names:
            v = sorted(agg[k].get(c, []))
            if v:
                fh.write("   %-32s median %16.0f  (n=%d)\n" % (c, v[len(v) // 2], len(v)))
print(open(os.path.join(out, "sq_summary.txt")).read())
PY
