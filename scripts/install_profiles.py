#!/usr/bin/env python3
"""Dev tool: copy the judged summaries of one scripts/gpu_round2.sh session (gpurun_out/<tag>) into profiles/<round>/
under a prefix, and refresh profiles/traffic.json from its PMC passes.
usage: install_profiles.py gpurun_out/d profiles/r02 d [old_prefix_to_remove]"""
import glob
import json
import os
import shutil
import sys

src, dst, pre = sys.argv[1], sys.argv[2], sys.argv[3]
old = sys.argv[4] if len(sys.argv) > 4 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if old:
    for f in glob.glob(os.path.join(dst, old + "_*")):
        os.remove(f)
names = {"00_env.log": "env.log", "01_pytest_gpu.log": "pytest_gpu.log", "02_smoke.log": "smoke.log",
         "08_summary.txt": "rocprof_summary.txt"}
names["10_sq_counters_cfg2.txt"] = "sq_counters_cfg2.txt"
names["10_sq_counters_cfg3p.txt"] = "sq_counters_cfg3p.txt"
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):        # (a session run with SKIP_TESTS / SKIP_SQ leaves some out)
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (pre, b)))
for f in glob.glob(os.path.join(src, "0[345]_bench_*.json")):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (pre, os.path.basename(f)[3:])))
for f in glob.glob(os.path.join(src, "09_train_step_example_*.json")):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (pre, os.path.basename(f)[3:])))
for cfg in ("cfg2", "cfg3p", "cfg5"):
    shutil.copy(os.path.join(src, "prof_stats_%s" % cfg, "%s_kernel_stats.csv" % cfg),
                os.path.join(dst, "%s_%s_kernel_stats.csv" % (pre, cfg)))
summary = json.load(open(os.path.join(src, "08_summary.json")))
tpath = os.path.join(root, "profiles", "traffic.json")
traffic = json.load(open(tpath))
for cfg, key in (("cfg2", "config2"), ("cfg3p", "config3"), ("cfg5", "config5")):
    new = {"B": traffic[key]["B"], "N": traffic[key]["N"]}
    new.update(summary[cfg]["traffic"])
    traffic[key] = new
if old:
    traffic["_doc"] = traffic["_doc"].replace("%s_rocprof_summary" % old, "%s_rocprof_summary" % pre)
json.dump(traffic, open(tpath, "w"), indent=1)
print("installed", src, "->", dst, pre)
