#!/usr/bin/env python3
"""Dev tool: copy the judged summaries of one scripts/gpu_round6.sh session (gpurun_out/<tag>) into profiles/r06/ under a prefix,
and profiles/traffic.json / profiles/issue.json -- unedited: both carry the sha256 of the library they were measured on.
usage: install_profiles6.py gpurun_out/r06x profiles/r06 z"""
import glob
import os
import shutil
import sys

src, dst, pre = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(dst, pre + "_*")):
    os.remove(f)
names = {"00_env.log": "env.log", "01_pytest_gpu.log": "pytest_gpu.log", "02_smoke.log": "smoke.log", "08_summary.txt": "rocprof_summary.txt",
         "09_bench_extras.json": "bench_extras.json", "11_issue.txt": "issue_summary.txt"}
for cfg in ("cfg2", "cfg3p", "cfg5"):
    names["10_sq_counters_%s.txt" % cfg] = "sq_counters_%s.txt" % cfg
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (pre, b)))
for f in glob.glob(os.path.join(src, "03_bench_*.json")):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (pre, os.path.basename(f)[3:])))
for cfg in ("cfg2", "cfg3p", "cfg3p_sigma0.8", "cfg5"):
    p = os.path.join(src, "prof_stats_%s" % cfg, "%s_kernel_stats.csv" % cfg)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s_kernel_stats.csv" % (pre, cfg)))
for f in ("traffic.json", "issue.json"):
    shutil.copy(os.path.join(src, f), os.path.join(root, "profiles", f))
print(sorted(os.listdir(dst)))
