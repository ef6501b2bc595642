#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_round4.sh (any set of workloads): per workload the kernel launch
statistics (prof_stats_<name>) and the PMC HBM bytes per launch (prof_pmc_{FETCH,WRITE}_SIZE_<name>; bytes =
(2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reports half of a streaming read on gfx950, see profiles/README.md and
the calibration block).  Writes the text to stdout and 08_summary.json; with --traffic also profiles/traffic.json,
stamped with the sha256 of the library (and of its sources) the counters were taken on.
usage: summarize_profiles4.py <session dir> [--traffic name=key:B:N ...]"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABEL = {"k_zsort": "zsort", "k_zhist": "zhist", "k_zscatter": "zscatter", "k_splat_xy": "splat_xy", "k_zfwd": "zfwd",
         "k_zbwd": "zbwd", "k_gather_yx": "gather_yx", "k_points_bwd_sorted": "points_bwd", "k_points_bwd_slots": "points_bwd", "k_pose_finalize": "pose_finalize",
         "k_points_fwd": "points_fwd", "k_points_bwd": "points_bwd", "k_blur_plane": "blur_plane", "k_blur_xy_stream": "blur_xy",
         "k_blur_z": "blur_z", "k_blur_z_generic": "blur_z", "k_sum_views": "sum_views"}


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").strip()
    return name[:70]


def rows_of(pattern):
    for f in sorted(glob.glob(os.path.join(out, pattern), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                yield row


names = sorted({os.path.basename(d)[len("prof_stats_"):] for d in glob.glob(os.path.join(out, "prof_stats_*"))} |
               {os.path.basename(d)[len("prof_pmc_FETCH_SIZE_"):] for d in glob.glob(os.path.join(out, "prof_pmc_FETCH_SIZE_*"))})
summary = {}
for cfg in names:
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
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]:
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
    if ctr["FETCH_SIZE"] or ctr["WRITE_SIZE"]:
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
        print("   step total: %.1f MB" % (traffic["_step_total"] / 1e6))
    print()
    summary[cfg] = dict(kernels=ks, traffic=traffic)
if glob.glob(os.path.join(out, "prof_calib_*")):
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


def sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def sha_sources():          # the same digest as bench.py source_sha256()
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "differentiable-point-clouds_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc")) +
                    [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "dpc_hip.h")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


if "--traffic" in sys.argv:
    doc = {"_doc": "HBM bytes per launch (median over launches) from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                   "passes (scripts/gpu_round6.sh -> scripts/summarize_profiles4.py), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: the "
                   "factor 2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (HBM), re-calibrated in the same "
                   "session with k_copy<1|2|4> on 512 MiB buffers.  lib_sha256 / src_sha256: the build of libdpc_hip.so (and "
                   "its sources, for information) the counters were taken on -- bench.py quotes these bytes only for the build with that lib_sha256; written by the script only, never edited.",
           "lib_sha256": sha_file(os.path.join(ROOT, "differentiable-point-clouds_amd", "csrc", "libdpc_hip.so")),
           "src_sha256": sha_sources()}
    for spec in sys.argv[sys.argv.index("--traffic") + 1:]:
        name, rest = spec.split("=")
        key, B, N = rest.split(":")
        if summary.get(name, {}).get("traffic"):
            doc[key] = dict({"B": int(B), "N": int(N)}, **summary[name]["traffic"])
    json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("traffic.json written:", sorted(k for k in doc if not k.startswith("_") and k not in ("lib_sha256", "src_sha256")))
