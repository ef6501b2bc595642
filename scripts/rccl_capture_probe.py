#!/usr/bin/env python3
"""Which event query of ProcessGroupNCCL's watchdog does a HIP-graph capture with a collective inside collide with?

    python scripts/rccl_capture_probe.py <variant>      (one rank, backend nccl; prints PROBE <variant> OK or dies)

The watchdog polls every 100 ms; every variant holds its window open for 0.5 s, so a collision is certain, not a race.
  leftover_unforked  eager barrier, capture begins at once, 0.5 s inside the capture BEFORE any collective
  leftover_forked    eager barrier, capture begins at once, a collective, then 0.5 s inside the capture
  drained_forked     eager barrier, 0.3 s pause, capture, a collective, then 0.5 s inside the capture
  drained_waited     as drained_forked, but work.wait() before the pause (the communication stream has joined again)
"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402

variant = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29647")
rank, world, dev = dpc_amd.distributed.init("nccl", force=True)
x = torch.ones(1 << 20, device=dev)
for _ in range(3):
    dist.all_reduce(x, op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
time.sleep(0.3)
dist.barrier()
torch.cuda.synchronize()
if variant.startswith("drained"):
    time.sleep(0.3)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y = x * 2.0
    if variant == "leftover_unforked":
        time.sleep(0.5)
    w = dist.all_reduce(y, op=dist.ReduceOp.AVG, async_op=True)
    if variant == "drained_waited":
        w.wait()
    if variant != "leftover_unforked":
        time.sleep(0.5)
    w.wait()
    z = y + 1.0
g.replay()
torch.cuda.synchronize()
assert float(z[0]) == 3.0
time.sleep(0.3)
print("PROBE", variant, "OK", flush=True)
dpc_amd.distributed.finalize()
