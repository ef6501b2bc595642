#!/usr/bin/env python3
"""Record a step that holds RCCL collectives into a HIP graph again and again beside the live watchdog thread.

    python scripts/rccl_capture_stress.py --records 40              # with the drain (distributed.drain_watchdog)
    python scripts/rccl_capture_stress.py --records 40 --drain 0    # without: expected to die with hipErrorCapturedEvent

One rank, backend nccl (a forced one-rank group): a small network's backward with GradBuckets' bucket all-reduces,
recorded `--records` times; between recordings a few replays and an eager barrier (whose work the watchdog then holds).
Prints one line per run; exit code 0 iff every recording and replay went through."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=40)
    ap.add_argument("--drain", type=float, default=None, help="seconds (default: the library's 0.25); 0 = off")
    args = ap.parse_args()
    if args.drain is not None:
        os.environ["DPC_WATCHDOG_DRAIN_S"] = str(args.drain)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dd = dpc_amd.distributed
    rank, world, dev = dd.init("nccl", force=True)
    torch.manual_seed(0)
    net = torch.nn.Sequential(*[torch.nn.Linear(1024, 1024) for _ in range(12)]).to(dev)
    red = dd.GradBuckets(net.parameters(), bucket_mb=8)
    x = torch.randn(256, 1024, device=dev)

    def run():
        red.zero_()
        net(x).square().mean().backward()
        red.finish()
    t0 = time.perf_counter()
    step = dpc_amd.graphs.RecordedStep(run, world=world, device=dev, collectives=True)
    for i in range(args.records - 1):
        for _ in range(5):
            step()
        dd.barrier(dev)                  # an eager collective: its work sits in the watchdog's list for up to 100 ms
        step._record()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print("OK: %d recordings of a step with %d bucket collectives, drain %s s, %.1f s [%s]" % (
        step.records, len(red.buckets), os.environ.get("DPC_WATCHDOG_DRAIN_S", "0.25"), time.perf_counter() - t0,
        dd.collective_library()), flush=True)
    dd.finalize()


if __name__ == "__main__":
    main()
