#!/usr/bin/env python3
"""Condense the SQ / GRBM counter passes of scripts/pmc_sq.sh into the per-kernel ISSUE figures bench.py quotes as
`roofline.issue` (round 6): instructions per wavefront, how busy the SIMDs' vector pipes are, what share of a wavefront's
cycles is spent waiting -- the numbers that say whether a kernel at 0.3 of HBM peak has 70 % headroom or is bound by its
own instruction stream.  Writes <session>/issue.json, stamped with the sha256 of the library the counters were taken on
(bench.py quotes it for that build only; copy it to profiles/issue.json unedited).
usage: summarize_sq.py <session dir> name=key:B:N ...      (name: the sq_<name> directory of pmc_sq.sh)

Counter units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts;
GRBM_GUI_ACTIVE counts shader cycles summed over the 8 XCDs.  Derived:
  valu_per_wave      SQ_INSTS_VALU / SQ_WAVES
  valu_active_frac   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES     (share of a wavefront's life with a vector instruction in the pipe)
  wait_inst_frac     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        (issue stalls: dependent chains, busy pipes)
  wait_any_frac      SQ_WAIT_ANY / SQ_WAVE_CYCLES             (parked on s_waitcnt / barriers: memory and LDS latency)
  valu_busy          SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel quad-cycles), kernel quad-cycles = GRBM_GUI_ACTIVE / 8 / 4
                     -- the fraction of the chip's vector issue slots the kernel fills: its distance to the VALU-issue bound
  waves_per_simd     average resident wavefronts per SIMD = SQ_WAVE_CYCLES / (1024 x kernel quad-cycles)"""
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
LABEL = {"k_zsort": "zsort", "k_zhist": "zhist", "k_zscatter": "zscatter", "k_splat_xy": "splat_xy", "k_zfwd": "zfwd", "k_zbwd": "zbwd",
         "k_gather_yx": "gather_yx", "k_points_bwd_sorted": "points_bwd", "k_points_bwd_slots": "points_bwd", "k_sum_views": "sum_views"}
SIMDS = 256 * 4


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


doc = {"_doc": __doc__.split("usage:")[0].strip() + "  Units and derivations: scripts/summarize_sq.py.",
       "lib_sha256": sha(os.path.join(ROOT, "differentiable-point-clouds_amd", "csrc", "libdpc_hip.so"))}
for spec in sys.argv[2:]:
    name, rest = spec.split("=")
    key, B, N = rest.split(":")
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, "sq_%s" % name, "pass*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"[<(].*$", "", row["Kernel_Name"]).replace("void ", "").strip()
            if k in LABEL:
                agg[LABEL[k]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    ent = {"B": int(B), "N": int(N)}
    for lab, ctr in sorted(agg.items()):
        med = {c: sorted(v)[len(v) // 2] for c, v in ctr.items()}
        need = ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE")
        if any(c not in med for c in need) or med["SQ_WAVES"] <= 0 or med["SQ_WAVE_CYCLES"] <= 0 or med["GRBM_GUI_ACTIVE"] <= 0:
            continue
        kq = med["GRBM_GUI_ACTIVE"] / 8.0 / 4.0
        ent[lab] = {
            "waves": int(med["SQ_WAVES"]),
            "valu_per_wave": med["SQ_INSTS_VALU"] / med["SQ_WAVES"],
            "salu_per_wave": med.get("SQ_INSTS_SALU", 0.0) / med["SQ_WAVES"],
            "vmem_per_wave": (med.get("SQ_INSTS_VMEM_RD", 0.0) + med.get("SQ_INSTS_VMEM_WR", 0.0)) / med["SQ_WAVES"],
            "valu_active_frac": med["SQ_ACTIVE_INST_VALU"] / med["SQ_WAVE_CYCLES"],
            "wait_inst_frac": med["SQ_WAIT_INST_ANY"] / med["SQ_WAVE_CYCLES"],
            "wait_any_frac": med["SQ_WAIT_ANY"] / med["SQ_WAVE_CYCLES"],
            "valu_busy": med["SQ_ACTIVE_INST_VALU"] / (SIMDS * kq),
            "waves_per_simd": med["SQ_WAVE_CYCLES"] / (SIMDS * kq),
            "kernel_cycles": med["GRBM_GUI_ACTIVE"] / 8.0,
        }
    doc[key] = ent
    print("== %s (%s)" % (name, key))
    for lab in sorted(k for k in ent if isinstance(ent[k], dict)):
        e = ent[lab]
        print("  %-10s waves %6d  VALU/wave %7.0f  SALU/wave %6.0f  VALU active %.2f  wait-inst %.2f  wait-any %.2f  | VALU busy %.2f  waves/SIMD %.2f"
              % (lab, e["waves"], e["valu_per_wave"], e["salu_per_wave"], e["valu_active_frac"], e["wait_inst_frac"], e["wait_any_frac"],
                 e["valu_busy"], e["waves_per_simd"]))
with open(os.path.join(out, "issue.json"), "w") as fh:
    json.dump(doc, fh, indent=1, sort_keys=True)
