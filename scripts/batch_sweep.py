#!/usr/bin/env python3
"""Dev tool: per-view time of every kernel against the batch size (does a smaller resident set help?).
usage: batch_sweep.py [bench args ...]   e.g.  --config 3 --projector-only"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in (int(b) for b in os.environ.get("SWEEP", "40,80,160,320,640").split(",")):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", str(B), "--steps", "30", "--warmup", "5",
                        "--no-cpu-baseline"] + sys.argv[1:], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    k = j["roofline"]["kernel_ms_per_step"]
    med = j["timing"]["ms_per_step_median"]
    print("B=%4d  %.4f ms/step  %.3f us/view | " % (B, med, 1e3 * med / B) + " ".join("%s=%.3f" % (a, 1e3 * b / B) for a, b in sorted(k.items())))
