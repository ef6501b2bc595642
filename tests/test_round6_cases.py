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


# ---- the sparse z walk (ZSkip, csrc/k_fused.inc) ----------------------------------------------------------------------
def both_walks_agree_bit_for_bit(lib, dev, B, N, D, K, sigma, Dz=-1, with_depth=False, cs=-1, seed=21):
    """dpc_set_sparse_walk(0 / 1): every wavefront walks every plane step / skips the groups in which none of its rays has anything.
    The skip advances the ray state with the dense walk's own operations in the dense walk's order: images, depth, point gradients
    bit for bit; the per-view pose / scale sums to the run-to-run noise of their float atomics."""
    inp = synth.make_inputs(B, N, seed)
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    gt = torch.tensor(synth.disk_gt(B, D), device=dev)
    res = []
    lib.dpc_set_chunk_sparse(cs)
    try:
        for walk in (0, 1):
            lib.dpc_set_sparse_walk(walk)
            t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
            pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
            out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
            up = [((out["proj"] - gt) / B).detach()]
            outs = [out["proj"]]
            if with_depth:          # a depth gradient comes in: the HAS_GD instantiations
                outs.append(out["proj_depth"])
                up.append((0.1 * torch.sin(out["proj_depth"])).detach())
            g = torch.autograd.grad(outs, [pc, pose, scale], up)
            res.append([out["proj"].detach().cpu().numpy(), out["proj_depth"].detach().cpu().numpy()] + [x.cpu().numpy() for x in g])
    finally:
        lib.dpc_set_sparse_walk(1)
        lib.dpc_set_chunk_sparse(-1)
    for a, b, name in zip(res[0], res[1], ("proj", "depth", "dpc", "dpose", "dscale")):
        if name in ("dpose", "dscale"):
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(a).max(), 1e-30), (name, float(np.abs(a - b).max()))
        else:
            assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))
    assert np.abs(res[0][2]).max() > 0


WALK_CASES_EMU = [
    # B, N, D, K, sigma, Dz, depth gradient, chunk-sparse mode
    (2, 300, 32, 5, 0.9, -1, False, -1),
    (1, 300, 64, 9, 1.4, -1, True, -1),
    (1, 300, 64, 3, 0.6, -1, False, 1),        # 3 taps: k_zfwd walks groups of 12 planes, k_zbwd of 6
    (1, 200, 64, 21, 3.0, -1, False, 0),       # dense layout, 21 taps, one loop body (64 planes: no run of empty planes is long enough)
    (1, 200, 64, 23, 1.2, -1, False, 1),       # G2 saved through the widened mask (the !FROM_T walk)
    (1, 400, 128, 11, 1.6, 32, False, -1),     # the headline's row width, shallow to keep the emulation quick
    (1, 300, 96, 7, 1.0, 24, False, -1),       # padded rows: no chunk maps, plane occupancy only
]


@pytest.mark.parametrize("case", WALK_CASES_EMU)
def test_emu_both_walks_agree_bit_for_bit(emu, poison_mode, case):
    B, N, D, K, sigma, Dz, depth, cs = case
    emu.dpc_emu_dead_groups_take()
    both_walks_agree_bit_for_bit(emu, "cpu", B, N, D, K, sigma, Dz=Dz, with_depth=depth, cs=cs)
    dead = emu.dpc_emu_dead_groups_take()
    assert dead > 0 or K == 21, "no wavefront skipped a group: the case does not exercise the sparse walk"


def test_emu_sparse_walk_against_the_numpy_oracle(emu):
    """the sparse walk (default on) through the fused path against the float64 oracle, incl. depth upstream, translation, focal"""
    import parity_cases
    emu.dpc_emu_dead_groups_take()
    parity_cases.fused_path_against_numpy_oracle("cpu", 2, 300, 64, 64, 5, 0.9, True, True)
    assert emu.dpc_emu_dead_groups_take() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(32, 8000, 128, 11, 1.6, -1, False, -1),     # BASELINE configs[1]
                                  (4, 16000, 256, 11, 2.0, -1, False, -1),     # configs[4]'s grid
                                  (40, 8000, 64, 21, 0.8, -1, False, -1),      # the training shape, 9 taps run
                                  (40, 8000, 64, 21, 3.0, -1, False, -1),      # ... 21 taps: dense layout
                                  (8, 8000, 128, 11, 1.6, -1, True, -1),       # a depth gradient coming in
                                  (4, 4000, 128, 23, 2.5, -1, False, -1),      # G2 saved
                                  (4, 8000, 256, 21, 3.0, -1, False, -1),      # 21 taps on 256-wide rows: G2 saved
                                  (8, 2000, 64, 3, 0.5, -1, False, 1),
                                  (4, 3000, 96, 7, 1.0, -1, False, -1)])
def test_gpu_both_walks_agree_bit_for_bit(gpu_lib, poison_mode, case):
    B, N, D, K, sigma, Dz, depth, cs = case
    both_walks_agree_bit_for_bit(gpu_lib, "cuda", B, N, D, K, sigma, Dz=Dz, with_depth=depth, cs=cs)
