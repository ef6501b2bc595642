#!/usr/bin/env python3
"""Per-kernel breakdown of the training step from a rocprofv3 kernel trace (scripts/gpu_train_prof.sh): launches and
GPU-busy time per step over the last 10 steps (a step = the launches from one k_zsort to the next).
usage: summarize_train_trace.py <dir with *kernel_trace.csv> [header note]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("void k_zsort") or r[2].startswith("k_zsort")]
if len(marks) < 12:
    sys.exit("need at least 12 steps in the trace, found %d" % len(marks))
lo, hi = marks[-11], marks[-1]          # ten whole steps (zsort .. next zsort), the last partial one dropped
steps = 10
sel = rows[lo:hi]


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "", n)
    n = n.replace("vectorized_elementwise_kernel<4, ", "VEW:").replace("elementwise_kernel<128, 4, ", "EW:")
    return n[:86]


agg = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    a = agg[short(n)]
    a[0] += 1
    a[1] += e - s
busy = sum(e - s for s, e, _ in sel) / steps / 1e3
lib = sum(v[1] for k, v in agg.items() if k.startswith("k_")) / steps / 1e3
span = (rows[hi][0] - rows[lo][0]) / steps / 1e3
print(" ".join(sys.argv[2:]))
print("launches per step: %d   GPU-busy per step: %.0f us   of which this library's kernels: %.0f us   (step span %.0f us)"
      % (len(sel) / steps, busy, lib, span))
print()
print("%-88s %8s %9s %10s" % ("kernel", "n/step", "avg us", "us/step"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-88s %8.1f %9.1f %10.1f" % (k, v[0] / steps, v[1] / v[0] / 1e3, v[1] / steps / 1e3))
