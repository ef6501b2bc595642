#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_round2.sh: per config the kernel launch statistics and the PMC
HBM bytes per launch (bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reports half of a streaming read on
gfx950, see profiles/README.md and the calibration block), written as text + a traffic.json fragment."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
LABEL = {"k_zsort": "zsort", "k_zhist": "zhist", "k_zscatter": "zscatter", "k_splat_xy": "splat_xy", "k_zfwd": "zfwd",
         "k_zbwd": "zbwd", "k_gather_yx": "gather_yx", "k_points_bwd_sorted": "points_bwd", "k_pose_finalize": "pose_finalize"}


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").strip()
    return name[:70]


def rows_of(pattern):
    for f in sorted(glob.glob(os.path.join(out, pattern), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                yield row


summary = {}
for cfg in ("cfg2", "cfg3p", "cfg5"):
    agg = defaultdict(list)
    for row in rows_of("prof_stats_%s/**/*kernel_trace.csv" % cfg):
        try:
            agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        except (KeyError, ValueError):
            pass
    tot = sum(sum(v) for v in agg.values()) or 1
    print("== %s: kernel trace (rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 ...) ==" % cfg)
    print("%-72s %6s %10s %10s %6s" % ("kernel", "calls", "avg_us", "median_us", "pct"))
    ks = []
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v.sort()
        print("%-72s %6d %10.2f %10.2f %6.2f" % (k, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, 100.0 * sum(v) / tot))
        ks.append(dict(kernel=k, calls=len(v), avg_us=sum(v) / len(v) / 1e3, median_us=v[len(v) // 2] / 1e3))
    ctr = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        a = defaultdict(list)
        for row in rows_of("prof_pmc_%s_%s/**/*counter_collection.csv" % (c, cfg)):
            try:
                a[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
            except (KeyError, ValueError):
                pass
        ctr[c] = {k: sorted(v)[len(v) // 2] for k, v in a.items()}
    traffic = {}
    print("\n-- %s: HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (medians) --" % cfg)
    for k in sorted(set(ctr["FETCH_SIZE"]) | set(ctr["WRITE_SIZE"])):
        base = k.split("<")[0]
        if base not in LABEL:
            continue
        f, w = ctr["FETCH_SIZE"].get(k, 0.0), ctr["WRITE_SIZE"].get(k, 0.0)
        b = int((2 * f + w) * 1024)
        traffic[LABEL[base]] = traffic.get(LABEL[base], 0) + b
        print("%-60s fetch %12.0f KB  write %12.0f KB  => %.1f MB" % (k, f, w, b / 1e6))
    traffic["_step_total"] = sum(v for k, v in traffic.items())
    print("   step total: %.1f MB\n" % (traffic["_step_total"] / 1e6))
    summary[cfg] = dict(kernels=ks, traffic=traffic)
print("== counter calibration (k_copy<W>: 512 MiB read + 512 MiB written per launch) ==")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a = defaultdict(list)
    for row in rows_of("prof_calib_%s/**/*counter_collection.csv" % c):
        try:
            a[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        except (KeyError, ValueError):
            pass
    for k in sorted(a):
        v = sorted(a[k])
        print("%-60s %-11s median %14.1f KB (524288 KB moved)" % (k, c, v[len(v) // 2]))
json.dump(summary, open(os.path.join(out, "08_summary.json"), "w"), indent=1)
