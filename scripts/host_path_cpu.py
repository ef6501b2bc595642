#!/usr/bin/env python3
"""Dev tool: Python/ctypes cost of one fwd+bwd step of pointcloud_project_fast WITHOUT a GPU.
The emulation library answers the layout queries; the two compute entry points are replaced by
no-ops, so what is timed is the host path alone (argument checks, allocations, struct marshalling,
autograd bookkeeping).  usage: host_path_cpu.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time

os.environ["DPC_TEST_HOOKS"] = "1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402

emu = dpc_amd._capi.DpcLibrary(os.path.join(ROOT, "tests", "hipemu", "libdpc_emu.so"), host_memory=True)
emu.dpc_project_forward = lambda *a: 0
emu.dpc_project_backward = lambda *a: 0
dpc_amd._capi.set_library(emu)
ext = dpc_amd._ext.module()           # the compiled binding (None with DPC_BINDING=ctypes or when it is not built)
if ext is not None:
    ext.set_dry_run(True)
print("binding:", "compiled (csrc/dpc_torch.cpp)" if ext is not None else "ctypes (ops.ProjectFused)")
torch.set_num_threads(1)

B, N, D, K = 4, 1000, 64, 11
cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
pc = torch.rand(B, N, 3, requires_grad=True)
pose = torch.rand(B, 4, requires_grad=True)
scale = torch.rand(B, 1, requires_grad=True)
kern = dpc_amd.smoothing_kernel(cfg, 1.0, device=pc.device)
gt = torch.zeros(B, D, D)


def step():
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, l2_target=(gt, 1.0 / B))
    return torch.autograd.grad(out["proj"], [pc, pose, scale], out["proj_l2_grad"])


for _ in range(200):
    step()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    step()
print("host path: %.1f us per fwd+bwd step" % ((time.perf_counter() - t0) / n * 1e6))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(1000):
        step()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
