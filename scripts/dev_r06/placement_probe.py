#!/usr/bin/env python3
"""Dev probe (round 6): is the process-to-process spread of cfg2's step (0.213 or 0.223 ms, k_zbwd 64 or 71 us, the same for every
build) a matter of WHERE the arena lands?  One process; before each trial the caching allocator is emptied and a spacer of a
different size is allocated first, so the op's per-call arena (and the saved grids inside it) sit at different addresses."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402
import bench  # noqa: E402

lib = dpc_amd._capi.get_library()
case = bench.build_case(int(os.environ.get("AB_CONFIG", "2")), None, torch.device("cuda"))
for trial, spacer_kb in enumerate([0, 4, 64, 1024, 2048 + 4, 7 * 1024, 64 * 1024 + 128, 200 * 1024, 0, 512, 3 * 1024 + 64]):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    spacer = torch.empty(spacer_kb * 1024, dtype=torch.uint8, device="cuda") if spacer_kb else None
    for _ in range(10):
        bench.step(case)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        bench.step(case)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    lib.profile(True)
    for _ in range(5):
        bench.step(case)
    torch.cuda.synchronize()
    k = {}
    for label, t in lib.profile_records():
        k.setdefault(label, []).append(t)
    lib.profile(False)
    # where the big buffers of one step sit (the ctypes binding allocates through torch.empty: DPC_BINDING=ctypes)
    seen = []
    real_empty = torch.empty
    def spy(*a, **kw):
        t = real_empty(*a, **kw)
        if t.is_cuda and t.numel() * t.element_size() >= (1 << 20):
            seen.append((t.numel() * t.element_size(), t.data_ptr()))
        return t
    torch.empty = spy
    try:
        bench.step(case)
    finally:
        torch.empty = real_empty
    torch.cuda.synchronize()
    print("   buffers: " + " ".join("%.1fMB@%#x" % (n / 2 ** 20, ptr) for n, ptr in seen), flush=True)
    print("spacer %7d KB  step %.4f ms | " % (spacer_kb, ms) + " ".join("%s=%.4f" % (a, sum(v) / len(v)) for a, v in sorted(k.items()) if sum(v) / len(v) > 0.004), flush=True)
    del spacer
