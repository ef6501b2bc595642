#!/usr/bin/env python3
"""Dev tool: how far ahead of the GPU is the Python host?  (a) enqueue vs total time per step at cfg2,
(b) the host-bound step time: the same call sequence at B = 1, where the GPU work is a few tens of microseconds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for B in (None, 1):
    case = bench.build_case(2, B, torch.device("cuda"))
    for _ in range(20):
        bench.step(case)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        bench.step(case)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%s: enqueue %.3f ms/step, total %.3f ms/step" % (case["B"], (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
