#!/usr/bin/env python3
"""Dev tool: cProfile of the Python host path of one fwd+bwd step (B = 1: the GPU work is negligible)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
case = bench.build_case(2, 1, torch.device("cuda"))
for _ in range(50):
    bench.step(case)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    bench.step(case)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(38)
