#!/usr/bin/env python3
"""Per-kernel resource table from the compiler's assembly listing (make -C differentiable-point-clouds_amd/csrc asm):
VGPRs, SGPRs, occupancy, scratch, static LDS and the static instruction mix.  usage: kernel_resources.py [filter ...]"""
import os
import re
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "differentiable-point-clouds_amd", "csrc", os.environ.get("DPC_ASM", "dpc_kernels.gfx950.s"))).read()
flt = sys.argv[1:]
print("%-58s %5s %5s %4s %7s %6s | %5s %5s %5s %4s %5s %8s" % ("kernel", "VGPR", "SGPR", "occ", "scratch", "LDS", "valu", "v_pk", "salu", "ds", "vmem", "readlane"))
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?\.end_amdhsa_kernel.*?; Occupancy: \d+)", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and not any(s in name for s in flt):
        continue
    def f(k):
        r = re.search(r"; %s: (\S+)" % k, body)
        return r.group(1) if r else "?"
    insts = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = Counter()
    for i in insts:
        if i.startswith("v_pk"):
            c["vpk"] += 1
        elif i.startswith("v_"):
            c["valu"] += 1
        elif i.startswith("s_"):
            c["salu"] += 1
        elif i.startswith("ds_"):
            c["ds"] += 1
        elif i.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
    short = re.sub(r"^_Z\d+", "", name)[:58]
    print("%-58s %5s %5s %4s %7s %6s | %5d %5d %5d %4d %5d %8d" % (short, f("NumVgprs"), f("NumSgprs"), f("Occupancy"), f("ScratchSize"), f("LDSByteSize"),
          c["valu"], c["vpk"], c["salu"], c["ds"], c["vmem"], sum(1 for i in insts if "readlane" in i)))
