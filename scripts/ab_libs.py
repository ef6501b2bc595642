#!/usr/bin/env python3
"""Dev tool: A/B per-kernel timings of several builds of the HIP library on the
cfg2 workload (interleaved rounds, one process).  usage: ab_libs.py lib1.so lib2.so ..."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402
import bench  # noqa: E402

cfg_id = int(os.environ.get("AB_CONFIG", "2"))
if os.environ.get("AB_SHAPE"):      # e.g. AB_SHAPE=320,8000,64,21,3.0  (B,N,D,K,sigma) -> ad-hoc config 9
    b_, n_, d_, k_, s_ = os.environ["AB_SHAPE"].split(",")
    dpc_amd.synthetic.CONFIGS[9] = dict(B=int(b_), N=int(n_), D=int(d_), K=int(k_), sigma=float(s_))
    cfg_id = 9
B = os.environ.get("AB_BATCH")
case = bench.build_case(cfg_id, int(B) if B else None, torch.device("cuda"))
# lib.so[@walk0][@cs0]: the same build with the sparse z walk / the chunk-sparse layout switched off (round 6)
libs, opts = [], {}
for spec in sys.argv[1:]:
    path, *flags = spec.split("@")
    name = os.path.basename(path) + "".join("@" + f for f in flags)
    libs.append((name, dpc_amd._capi.DpcLibrary(os.path.abspath(path))))
    opts[name] = flags
rounds, steps = int(os.environ.get("AB_ROUNDS", "3")), int(os.environ.get("AB_STEPS", "20"))
res = {n: {"ms": [], "k": {}} for n, _ in libs}
for r in range(rounds):
    for name, lib in libs:
        dpc_amd._capi.set_library(lib)
        lib.dpc_set_sparse_walk(0 if "walk0" in opts[name] else 1)
        lib.dpc_set_chunk_sparse(0 if "cs0" in opts[name] else -1)
        for _ in range(5):
            bench.step(case)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bench.step(case)
        torch.cuda.synchronize()
        res[name]["ms"].append((time.perf_counter() - t0) / steps * 1e3)
        lib.profile(True)
        for _ in range(5):
            bench.step(case)
        torch.cuda.synchronize()
        for label, ms in lib.profile_records():
            res[name]["k"].setdefault(label, []).append(ms)
        lib.profile(False)
for name, _ in libs:
    k = res[name]["k"]
    per = {lab: sum(v) / (rounds * 5) for lab, v in k.items()}
    print("%-24s step %.3f ms (min %.3f) | " % (name, sorted(res[name]["ms"])[len(res[name]["ms"]) // 2], min(res[name]["ms"])) +
          " ".join("%s=%.3f" % (lab, per[lab]) for lab in sorted(per) if per[lab] > 0.004))
