#!/usr/bin/env python3
"""Dev diagnostic: the point-gradient entries of a config that are furthest (elementwise) from the fp64 CPU oracle."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dpc_amd
from helpers import rcpu, synth, onp
import parity_cases
cfg_id, B = int(sys.argv[1]), int(sys.argv[2])
c = synth.config_inputs(cfg_id, B=B)
c["pc"] = parity_cases._nudge_off_cell_faces({"pc": c["pc"], "pose": c["pose"]}, None, None, c["D"], c["D"])["pc"]
D = c["D"]
cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=c["K"])
t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device="cuda")
out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
gt = torch.tensor(synth.disk_gt(B, D), device="cuda")
dproj = ((out["proj"] - gt) / B).detach()
g = torch.autograd.grad(out["proj"], [pc, pose, scale], dproj)[0].cpu().numpy().astype(np.float64)
rc = rcpu.Cfg(vox_size=D, pc_gauss_kernel_size=c["K"])
ck = rcpu.smoothing_kernel(rc, c["sigma"], torch.float64)
ref = []
tr = []
for b in range(B):
    d = lambda a: torch.tensor(a[b:b + 1], dtype=torch.float64, requires_grad=True)
    cpc, cpose, cscale = d(c["pc"]), d(c["pose"]), d(c["scale"])
    r = rcpu.pointcloud_project_fast(rc, cpc, cpose, None, None, ck, scaling_factor=cscale)
    ref.append(torch.autograd.grad(r["proj"], [cpc], dproj[b:b + 1].cpu().double())[0].numpy())
    tr.append(r["tr_pc"].detach().numpy())
ref = np.concatenate(ref); tr = np.concatenate(tr)
scale_ = np.abs(ref).max()
bound = 2e-5 * scale_ + 1e-3 * np.abs(ref)
ratio = np.abs(g - ref) / bound
print("max|ref| %.3e; entries over the bound: %d of %d; worst ratio %.2f" % (scale_, (ratio > 1).sum(), ratio.size, ratio.max()))
idx = np.argsort(ratio.max(-1).ravel())[::-1][:12]
for i in idx:
    b, n = divmod(i, ref.shape[1])
    gl = (tr[b, n] + 0.5) * (D - 1)
    print("b=%d n=%5d ratio %.2f  got %s ref %s | lattice (z,y,x) = %s frac %s" % (
        b, n, ratio[b, n].max(), np.array2string(g[b, n], precision=4), np.array2string(ref[b, n], precision=4),
        np.array2string(np.floor(gl), precision=0), np.array2string(gl - np.floor(gl), precision=5)))
