#!/usr/bin/env python3
"""Dev tool: where does the fixed cost of the first timed block after a synchronize go?  Per-step HIP-event
timestamps of a 20-step block that starts from an idle GPU, eager and graph-replayed, plus the host-side clock."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
case = bench.build_case(2, None, torch.device("cuda"))
def make_graph():
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): bench.step(case)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        case["g"] = bench.step(case)
    return g.replay
for name, run in (("eager", lambda: bench.step(case)), ("graph", make_graph())):
    for _ in range(10): run()
    for trial in range(3):
        torch.cuda.synchronize()
        if trial == 2: time.sleep(0.05)            # a longer idle gap
        K = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(K):
            run(); ev[i + 1].record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        d = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
        print("%s trial %d: wall %.3f ms (enqueue %.3f) = %.4f ms/step | gpu first-event..last %.3f ms | steps: %s"
              % (name, trial, (t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3 / K, ev[0].elapsed_time(ev[K]),
                 " ".join("%.3f" % x for x in d)))
