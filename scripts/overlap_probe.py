#!/usr/bin/env python3
"""Dev probe: do two half-batches on two HIP streams overlap better than one full batch?
Everything is replayed from HIP graphs so the host is out of the picture."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda")
full = bench.build_case(2, 32, dev)
halves = [bench.build_case(2, 16, dev, seed_offset=i) for i in range(2)]
quarters = [bench.build_case(2, 8, dev, seed_offset=i) for i in range(4)]


def capture(cases):
    streams = [torch.cuda.Stream() for _ in cases]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for c in cases:
            for _ in range(3):
                bench.step(c)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if len(cases) == 1:
            keep.append(bench.step(cases[0]))
        else:
            for s, c in zip(streams, cases):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    keep.append(bench.step(c))
            for s in streams:
                cur.wait_stream(s)
    return g, keep


def timeit(g, n=200):
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, cases in (("1 x 32", [full]), ("2 x 16", halves), ("4 x 8", quarters)):
    g, keep = capture(cases)
    print("%s views on %d stream(s): %.3f ms per 32 views" % (name, len(cases), timeit(g)))
