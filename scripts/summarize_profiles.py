#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_round.sh into small text/JSON
summaries fit for profiles/: per-kernel launch statistics, PMC byte counters
per launch, and the counter calibration factors from pmc_calibrate.py."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r"\(.*$", "", name)           # drop the argument list
    name = name.replace("void ", "")
    return name.strip()[:80]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def kernel_trace_stats(files):
    agg = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                try:
                    agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                except (KeyError, ValueError):
                    pass
    rows = []
    for k, v in agg.items():
        v.sort()
        rows.append((sum(v), k, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1]))
    rows.sort(reverse=True)
    return rows


def counters(files):
    agg = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                try:
                    agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
                except (KeyError, ValueError):
                    pass
    return agg


summary = {}
print("== kernel trace (cfg2 bench, --kernel-trace --stats run) ==")
rows = kernel_trace_stats(find("prof_stats/**/*kernel_trace.csv"))
tot = sum(r[0] for r in rows) or 1
print("%-72s %6s %12s %12s %6s" % ("kernel", "calls", "avg_us", "median_us", "pct"))
for total, k, n, avg, med, lo, hi in rows[:20]:
    print("%-72s %6d %12.2f %12.2f %6.2f" % (k, n, avg / 1e3, med / 1e3, 100.0 * total / tot))
summary["kernel_trace"] = [dict(kernel=k, calls=n, avg_us=avg / 1e3, median_us=med / 1e3, pct=100.0 * total / tot)
                           for total, k, n, avg, med, lo, hi in rows[:20]]

for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    print("\n== PMC %s per launch (raw counter units; rocprofv3 documents KB) ==" % ctr)
    c = counters(find("prof_pmc_%s/**/*counter_collection.csv" % ctr))
    summary["pmc_" + ctr] = {}
    for k in sorted(c):
        for name, vals in c[k].items():
            vals.sort()
            med = vals[len(vals) // 2]
            print("%-72s %-12s n=%4d median=%14.1f mean=%14.1f" % (k, name, len(vals), med, sum(vals) / len(vals)))
            summary["pmc_" + ctr][k] = dict(n=len(vals), median=med, mean=sum(vals) / len(vals))
    print("\n== calibration %s (512 MiB read + 512 MiB written per copy launch; zero_ writes 512 MiB) ==" % ctr)
    c = counters(find("prof_calib_%s/**/*counter_collection.csv" % ctr))
    summary["calib_" + ctr] = {}
    for k in sorted(c):
        for name, vals in c[k].items():
            vals.sort()
            med = vals[len(vals) // 2]
            print("%-72s %-12s n=%4d median=%14.1f  => bytes/unit if 512MiB: %.1f" %
                  (k, name, len(vals), med, (512 * 1024 * 1024) / med if med else float("nan")))
            summary["calib_" + ctr][k] = dict(n=len(vals), median=med)

with open(os.path.join(out, "08_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
