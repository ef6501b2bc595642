#!/usr/bin/env python3
"""Dev probe: does running the projector's kernels of two half batches on two HIP streams -- so that a VALU-bound kernel
of one half (k_splat_xy, k_gather_yx at 21 taps) overlaps a bandwidth-bound one of the other (k_zfwd, k_zbwd) -- beat
one launch chain over the whole batch?  usage: overlap_probe.py [B,N,D,K,sigma]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402
import bench  # noqa: E402

shape = (sys.argv[1] if len(sys.argv) > 1 else "320,8000,64,21,3.0").split(",")
B, N, D, K, sigma = int(shape[0]), int(shape[1]), int(shape[2]), int(shape[3]), float(shape[4])
dpc_amd.synthetic.CONFIGS[9] = dict(B=B, N=N, D=D, K=K, sigma=sigma)
dev = torch.device("cuda")
full = bench.build_case(9, B, dev)
halves = [bench.build_case(9, B // 2, dev, seed_offset=s) for s in (0, 1)]
quarters = [bench.build_case(9, B // 4, dev, seed_offset=s) for s in range(4)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def graphed(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    return g.replay, keep


def one():
    return bench.step(full)


def chunks_serial(cs):
    return [bench.step(c) for c in cs]


def chunks_two_streams(cs, stagger_cycles=0):
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur)
    sB.wait_stream(cur)
    out = []
    for i, c in enumerate(cs):
        st = sA if i % 2 == 0 else sB
        with torch.cuda.stream(st):
            if stagger_cycles and i == 1:
                torch.cuda._sleep(stagger_cycles)
            out.append(bench.step(c))
    cur.wait_stream(sA)
    cur.wait_stream(sB)
    return out


res = {}
for name, fn in [("full batch, one chain", one),
                 ("2 halves, serial", lambda: chunks_serial(halves)),
                 ("2 halves, 2 streams", lambda: chunks_two_streams(halves)),
                 ("2 halves, 2 streams, second delayed 60 us", lambda: chunks_two_streams(halves, 120000)),
                 ("4 quarters, serial", lambda: chunks_serial(quarters)),
                 ("4 quarters, 2 streams", lambda: chunks_two_streams(quarters)),
                 ("4 quarters, 2 streams, second delayed 30 us", lambda: chunks_two_streams(quarters, 60000))]:
    try:
        replay, keep = graphed(fn)
        res[name] = timeit(replay)
    except Exception as e:  # noqa: BLE001
        res[name] = float("nan")
        print("capture failed for", name, type(e).__name__, e)
        torch.cuda.synchronize()
    print("%-46s %.3f ms per %d views" % (name, res[name], B), flush=True)
