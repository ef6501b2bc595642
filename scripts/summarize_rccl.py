#!/usr/bin/env python3
"""Summarise the one-rank RCCL session (scripts/gpu_round5.sh, section `rccl`): the bench lines' parallelism notes and,
from the rocprofv3 kernel traces of the recorded / eager training step, which kernels RCCL launched, how often and
where they sit between the step's other kernels."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
RCCL = ("nccl", "rccl", "oneRank", "OneRank", "msccl")


def is_rccl(name):
    return any(k in name for k in RCCL)


for f in sorted(glob.glob(os.path.join(out, "21_bench_*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][0])      # (RCCL's banner may precede the line)
        print("%-32s %9.0f views/s  %.3f ms/step  hip_graph=%s\n    parallelism: %s" % (
            os.path.basename(f)[9:-5], j["value"], j["ms_per_step"], j["config"]["hip_graph"], j["config"]["parallelism"]))
    except Exception as e:   # noqa: BLE001
        print(f, "ERR", e)
        try:
            print(open(f.replace(".json", ".err")).read()[-1500:])
        except OSError:
            pass
for tag in ("rccl_graph", "rccl_ddp"):
    traces = glob.glob(os.path.join(out, "prof_" + tag, "**", "*kernel_trace.csv"), recursive=True)
    if not traces:
        print("\n[%s] no kernel trace" % tag)
        continue
    rows = list(csv.DictReader(open(traces[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    hits = [i for i, n in enumerate(names) if is_rccl(n)]
    print("\n[%s] %d kernel launches traced, %d by RCCL" % (tag, len(rows), len(hits)))
    by = {}
    for i in hits:
        d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        k = names[i][:110]
        by.setdefault(k, []).append(d)
    for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print("   %5d x  mean %8.1f us  %s" % (len(v), sum(v) / len(v), k))
    # the neighbourhood of the LAST step's collectives: what ran right before / after each
    if hits:
        last = hits[-min(len(hits), 4):]
        print("   neighbourhood of the last collectives (stream order by start time; queue id in brackets):")
        for i in last:
            for jx in range(max(0, i - 2), min(len(rows), i + 3)):
                r = rows[jx]
                print("     %s [q%s] %9.1f us  %s" % ("->" if jx == i else "  ", r.get("Queue_Id", "?"),
                                                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, names[jx][:100]))
            print("     --")
    ours = ("k_zsort", "k_zhist", "k_zscatter", "k_splat_xy", "k_zfwd", "k_zbwd", "k_gather_yx", "k_points_bwd", "k_sum_views",
            "k_sil_", "k_student", "k_cs_")
    lib = sum(1 for n in names if any(k in n for k in ours))
    print("   this library's kernels in the same trace: %d launches" % lib)

print()
for f in sorted(glob.glob(os.path.join(out, "24_stress_*.log"))):
    txt = open(f).read()
    ok = [l for l in txt.splitlines() if l.startswith("OK:")]
    err = [l for l in txt.splitlines() if "HIP error:" in l]
    print("%-28s %s" % (os.path.basename(f), ok[0] if ok else ("DIED: " + (err[0].split("HIP error:")[1].strip() if err else txt[-200:]))))
