#!/usr/bin/env python3
"""Which event query of ProcessGroupNCCL's watchdog does a HIP-graph capture with a collective inside collide with?

    python scripts/rccl_capture_probe.py <variant>      (one rank, backend nccl; prints PROBE <variant> OK or dies)

The watchdog polls every 100 ms; every variant holds its window open for 0.5 s, so a collision is certain, not a race.
  leftover_unforked  eager barrier, capture begins at once, 0.5 s inside the capture BEFORE any collective
  leftover_forked    eager barrier, capture begins at once, a collective, then 0.5 s inside the capture
  drained_forked     eager barrier, 0.3 s pause, capture, a collective, then 0.5 s inside the capture
  drained_waited     as drained_forked, but work.wait() before the pause (the communication stream has joined again)
  burst_forked       200 eager all-reduces, a synchronize, then AT ONCE (raw capture_begin: no gc.collect in between) a capture
                     with a collective, held open 0.3 s: the watchdog's list is certainly not empty when it polls inside
  burst_unforked     the same, but the 0.3 s pass BEFORE the capture's collective (RCCL's stream not yet forked)
  burst_side_forked  burst_forked with the eager all-reduces issued (from the capturing thread) on a SIDE stream, like the warm-up
  hooks_held         the stress scenario made certain: 20 eager steps whose bucket all-reduces are issued from GRADIENT HOOKS (the
                     autograd engine's thread) on a side stream, a synchronize, then at once a capture of the same step (hooks
                     again), held open 0.3 s
  hooks_held_drained the same with 0.3 s between the eager steps and the capture (the watchdog's list is empty)
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
if variant.startswith("hooks"):
    torch.manual_seed(0)
    net = torch.nn.Sequential(*[torch.nn.Linear(1024, 1024) for _ in range(8)]).to(dev)
    red = dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=8)
    xin = torch.randn(256, 1024, device=dev)

    def run():
        red.zero_()
        net(xin).square().mean().backward()
        red.finish()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(20):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if variant == "hooks_held_drained":
        time.sleep(0.3)
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        g.capture_begin(capture_error_mode="thread_local")
        run()
        time.sleep(0.3)
        g.capture_end()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    time.sleep(0.3)
    print("PROBE", variant, "OK (%d buckets)" % len(red.buckets), flush=True)
    dpc_amd.distributed.finalize()
    sys.exit(0)
if variant.startswith("burst"):
    if variant == "burst_side_forked":
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(200):
                dist.all_reduce(x, op=dist.ReduceOp.AVG)
        torch.cuda.current_stream().wait_stream(warm)
    else:
        for _ in range(200):
            dist.all_reduce(x, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g.capture_begin(capture_error_mode="thread_local")
        y = x * 2.0
        if variant == "burst_unforked":
            time.sleep(0.3)
        w = dist.all_reduce(y, op=dist.ReduceOp.AVG, async_op=True)
        if variant != "burst_unforked":
            time.sleep(0.3)
        w.wait()
        z = y + 1.0
        g.capture_end()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    time.sleep(0.3)
    print("PROBE", variant, "OK", flush=True)
    dpc_amd.distributed.finalize()
    sys.exit(0)
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
