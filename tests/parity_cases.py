"""Parity checks shared by the CPU emulation tier (tests/test_emu_kernels.py,
device "cpu" + emulation library) and the GPU tier (tests/test_gpu_parity.py,
device "cuda" + libdpc_hip.so)."""
import numpy as np
import torch

import dpc_amd
from helpers import maxabs, onp, rcpu, relerr, synth

TOL_PROJ = 2e-5
TOL_DEPTH = 2e-4
TOL_GRAD = 2e-4
ODD_CASES = [(20, 13), (17, 3), (40, 9)]


def stage_level_api_matches_cpu_oracle(dev):
    """The finer-grained reference API (predict.py:130-132 style), each stage
    with its own autograd node, against reference_cpu on the same inputs."""
    rng = np.random.default_rng(3)
    B, N, D, K, sigma = 2, 300, 24, 7, 1.1        # K=7: generic plane path + fixed z path
    inp = synth.make_inputs(B, N, 77)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    rc = rcpu.Cfg(vox_size=D, pc_gauss_kernel_size=K)
    pc = torch.tensor(inp["pc"], device=dev, requires_grad=True)
    pose = torch.tensor(inp["pose"], device=dev, requires_grad=True)
    cpc = torch.tensor(inp["pc"], dtype=torch.float64, requires_grad=True)
    cpose = torch.tensor(inp["pose"], dtype=torch.float64, requires_grad=True)

    tr = dpc_amd.pc_perspective_transform(cfg, pc, pose)
    vox, _ = dpc_amd.pointcloud2voxels3d_fast(cfg, tr, None)
    vox = torch.clamp(vox.unsqueeze(-1), 0.0, 1.0)
    sm = dpc_amd.smoothen_voxels3d(cfg, vox, dpc_amd.smoothing_kernel(cfg, sigma, device=dev))
    proj, p = dpc_amd.drc_projection(sm, cfg)
    depth = dpc_amd.drc_depth_projection(p, cfg)

    ctr = rcpu.pc_perspective_transform(rc, cpc, cpose)
    cvox, _ = rcpu.pointcloud2voxels3d_fast(rc, ctr, None)
    cvox = torch.clamp(cvox.unsqueeze(-1), 0.0, 1.0)
    csm = rcpu.smoothen_voxels3d(rc, cvox, rcpu.smoothing_kernel(rc, sigma, torch.float64))
    cproj, cp = rcpu.drc_projection(csm, rc)
    cdepth = rcpu.drc_depth_projection(cp, rc)

    assert maxabs(tr.detach().cpu().numpy(), ctr.detach().numpy()) < 2e-6
    assert maxabs(sm.detach().cpu().numpy(), csm.detach().numpy()) < 1e-5
    assert maxabs(proj.detach().cpu().numpy(), cproj.detach().numpy()) < TOL_PROJ
    assert maxabs(p.detach().cpu().numpy(), cp.detach().numpy()) < TOL_PROJ
    assert maxabs(depth.detach().cpu().numpy(), cdepth.detach().numpy()) < TOL_DEPTH

    w1 = rng.standard_normal(proj.shape)
    w2 = 0.1 * rng.standard_normal(p.shape)
    loss = (proj * torch.tensor(w1, dtype=torch.float32, device=dev)).sum() + \
           (p * torch.tensor(w2, dtype=torch.float32, device=dev)).sum()
    closs = (cproj * torch.tensor(w1)).sum() + (cp * torch.tensor(w2)).sum()
    g = torch.autograd.grad(loss, [pc, pose])
    cg = torch.autograd.grad(closs, [cpc, cpose])
    assert relerr(g[0].cpu().numpy(), cg[0].numpy()) < TOL_GRAD
    assert relerr(g[1].cpu().numpy(), cg[1].numpy()) < TOL_GRAD


def odd_sizes_and_generic_tap_counts(dev, D, K):
    """Generic (run-time K) kernels, odd D (scalar ray path), D not a multiple
    of the y-tile: fused path vs the NumPy oracle."""
    B, N, sigma = 2, 400, 1.3
    inp = synth.make_inputs(B, N, 500 + D)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    pc = torch.tensor(inp["pc"], device=dev, requires_grad=True)
    pose = torch.tensor(inp["pose"], device=dev, requires_grad=True)
    scale = torch.tensor(inp["scale"], device=dev, requires_grad=True)
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg, sigma, device=dev),
                                          scaling_factor=scale)
    w = np.random.default_rng(D).standard_normal(out["proj"].shape)
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], torch.tensor(w, dtype=torch.float32, device=dev))
    f64 = lambda a: a.astype(np.float64)
    taps = onp.smoothing_taps(D, -1, K, sigma)
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, Dz=D, D=D)
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, fw, dproj=w)
    assert maxabs(out["proj"].detach().cpu().numpy(), fw["proj"]) < TOL_PROJ
    assert relerr(g[0].cpu().numpy(), bw["dpc"]) < TOL_GRAD
    assert relerr(g[1].cpu().numpy(), bw["dpose"]) < TOL_GRAD
    assert relerr(g[2].cpu().numpy(), bw["dscale"]) < TOL_GRAD
