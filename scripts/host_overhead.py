#!/usr/bin/env python3
"""Dev tool: is the Python host ahead of the GPU?  Enqueue time vs total time per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
case = bench.build_case(2, None, torch.device("cuda"))
for _ in range(20):
    bench.step(case)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    bench.step(case)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
