"""Round-6 cases on the CPU tiers (GPU counterparts at the bottom, marked gpu)."""
import ctypes

import numpy as np
import pytest
import torch

import dpc_amd
from helpers import synth
from oracle import dpc_oracle_np as onp


def _max_projection_case(dev, D, K, sigma, B=2, N=400, Dz=-1):
    inp = synth.make_inputs(B, N, 11)
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K, ptn_max_projection=True)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    w = np.random.default_rng(3).standard_normal(tuple(out["proj"].shape))
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], torch.tensor(w, dtype=torch.float32, device=dev))
    return inp, cfg, out["proj"].detach().cpu().numpy(), [x.cpu().numpy() for x in g], w


def max_projection_on_a_fused_shape(lib, dev, D, K, sigma):
    """ptn_max_projection (point_cloud.py:264-267) on a shape whose front / back end is the fused one (k_splat_xy, k_gather_yx)
    while the collapse is k_max_*: the grids between them are DENSE and no chunk marks exist.  Round 5 handed k_gather_yx the
    chunk-sparse flag all the same: with marks read from a zero-filled arena it dropped every chunk.
    Layout forced off == the rule == forced on, and all against the float64 NumPy oracle."""
    S = dpc_amd._capi.DpcShape(2, 400, D, D, K, K, K)
    P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, dpc_amd._capi.DPC_COLLAPSE_MAX, 0, 0, 0)
    lay = lib.dpc_saved_layout(ctypes.byref(S), ctypes.byref(P))
    assert lay & 6 == 6, "expected the fused front end for this shape"
    res = {}
    for mode in (0, -1, 1):
        lib.dpc_set_chunk_sparse(mode)
        assert not lib.dpc_saved_layout(ctypes.byref(S), ctypes.byref(P)) & 16     # no chunk-sparse grids without the fused z pass
        inp, cfg, proj, grads, w = _max_projection_case(dev, D, K, sigma)
        res[mode] = (proj, grads)
    lib.dpc_set_chunk_sparse(-1)
    for mode in (-1, 1):
        assert np.array_equal(res[0][0], res[mode][0])
        assert np.array_equal(res[0][1][0], res[mode][1][0]), float(np.abs(res[0][1][0] - res[mode][1][0]).max())     # dpc bit for bit
        for a, b in zip(res[0][1][1:], res[mode][1][1:]):             # pose / scale sums: float atomics' run-to-run noise
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(a).max(), 1e-30)
    f64 = lambda a: a.astype(np.float64)
    taps = onp.smoothing_taps(D, -1, K, sigma)
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, Dz=D, D=D, max_projection=True)
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, fw, dproj=w, max_projection=True)
    proj, grads = res[-1]
    assert np.abs(proj - fw["proj"]).max() <= 2e-5
    assert np.abs(bw["dpc"]).max() > 0
    for name, g in zip(("dpc", "dpose", "dscale"), grads):
        assert np.abs(g.reshape(bw[name].shape) - bw[name]).max() <= 2e-4 * np.abs(bw[name]).max(), name


@pytest.mark.parametrize("D,K,sigma", [(32, 5, 0.9), (64, 5, 0.9)])
def test_emu_max_projection_on_a_fused_shape(emu, poison_mode, D, K, sigma):
    try:
        max_projection_on_a_fused_shape(emu, "cpu", D, K, sigma)
    finally:
        emu.dpc_set_chunk_sparse(-1)


def uncompiled_z_taps_on_a_fused_shape(lib, dev):
    """vox_size_z > vox_size stretches the z filter beyond the compiled tap counts (gauss_kernel.py:38-50: 11 taps on a 32-wide
    grid with 128 planes -> 45 z taps): fused front / back end, generic z blur, dense grids in between -- the other shape family
    of the same ADVICE item.  Against the float64 NumPy oracle, chunk-sparse rule and forced on."""
    B, N, D, Dz, K, sigma = 1, 300, 32, 128, 11, 1.2
    inp = synth.make_inputs(B, N, 12)
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    taps = onp.smoothing_taps(D, Dz, K, sigma)
    Kz = len(taps[2])
    assert Kz > 31 and lib.dpc_compiled_taps(Kz) != Kz
    f64 = lambda a: a.astype(np.float64)
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, Dz=Dz, D=D)
    w = np.random.default_rng(4).standard_normal((B, D, D, 1))
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, fw, dproj=w)
    for mode in (-1, 1):
        lib.dpc_set_chunk_sparse(mode)
        t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
        pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
        g = torch.autograd.grad(out["proj"], [pc, pose, scale], torch.tensor(w, dtype=torch.float32, device=dev))
        assert np.abs(out["proj"].detach().cpu().numpy() - fw["proj"]).max() <= 2e-5
        for name, x in zip(("dpc", "dpose", "dscale"), g):
            assert np.abs(x.cpu().numpy().reshape(bw[name].shape) - bw[name]).max() <= 2e-4 * np.abs(bw[name]).max(), (mode, name)
    lib.dpc_set_chunk_sparse(-1)


def test_emu_uncompiled_z_taps_on_a_fused_shape(emu, poison_mode):
    try:
        uncompiled_z_taps_on_a_fused_shape(emu, "cpu")
    finally:
        emu.dpc_set_chunk_sparse(-1)


# ---- the same on the device ------------------------------------------------------------------------------------------
@pytest.fixture
def gpu_lib():
    dpc_amd._capi.set_library(None)
    lib = dpc_amd.get_library()
    assert lib.path.endswith("libdpc_hip.so") and not lib.host_memory
    yield lib
    lib.dpc_set_chunk_sparse(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("D,K,sigma", [(32, 5, 0.9), (64, 5, 0.9), (128, 11, 1.6)])
def test_gpu_max_projection_on_a_fused_shape(gpu_lib, poison_mode, D, K, sigma):
    max_projection_on_a_fused_shape(gpu_lib, "cuda", D, K, sigma)


@pytest.mark.gpu
def test_gpu_uncompiled_z_taps_on_a_fused_shape(gpu_lib, poison_mode):
    uncompiled_z_taps_on_a_fused_shape(gpu_lib, "cuda")
