#!/usr/bin/env python3
"""Dev probe: which combination of (binding, what the recorded step returns / keeps) survives HIP graph capture.
Each variant runs in its own process; prints the exit codes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BODY = r'''
import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, %r)
import dpc_amd
from dpc_amd import synthetic as synth
variant = sys.argv[1]
dev = torch.device("cuda")
c = synth.config_inputs(1)
cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device=dev)
t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
gt = torch.tensor(synth.disk_gt(c["B"], c["D"]), device=dev)
def run():
    kw = {} if "nol2" in variant else dict(l2_target=(gt, 0.25))
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, **kw)
    up = out["proj_l2_grad"] if "nol2" not in variant else torch.ones_like(out["proj"])
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], up)
    if "keepout" in variant:
        return [out["proj"], out["proj_depth"]] + list(g)
    if "detach" in variant:
        return [out["proj"].detach(), out["proj_depth"].detach()] + list(g)
    return list(g)
if "reload" in variant:
    assert dpc_amd._ext.module() is not None
    dpc_amd._ext.reset()
    assert dpc_amd._ext.module() is not None
if "clone" in variant:
    keep = [x.clone() for x in run()]
if "eagerfirst" in variant:
    run(); torch.cuda.synchronize()
step = dpc_amd.graphs.RecordedStep(run, world=1, device=dev)
r = step(); torch.cuda.synchronize()
print("OK", variant, float(r[-1].sum()))
''' % ROOT
for binding in ("compiled", "ctypes"):
    for variant in ("keepout", "keepout_reload", "keepout_clone", "keepout_poison", "keepout_clone_poison_reload"):
        env = dict(os.environ, DPC_BINDING="" if binding == "compiled" else "ctypes")
        if "poison" in variant:
            env.update(DPC_POISON_BUFFERS="1", DPC_TEST_HOOKS="1")
        r = subprocess.run([sys.executable, "-c", BODY, variant], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        tail = (r.stderr.strip().splitlines() or [""])[-1][:200] if r.returncode else ""
        print("%-9s %-20s rc=%d %s %s" % (binding, variant, r.returncode, r.stdout.strip()[-60:], tail), flush=True)
