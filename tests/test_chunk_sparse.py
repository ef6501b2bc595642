"""Chunk-sparse grids (csrc/k_prelude.inc, host_launch.inc chunk_sparse_on): the fused path stores / loads only the 128-byte
chunks of the saved and gradient grids that lie within the blur's reach of a point on their plane.  It is a layout decision,
not an approximation: with dpc_set_chunk_sparse(0 / 1) the two forms must agree BIT FOR BIT in the images and in the point
gradients (the per-view pose / scale sums, which run through float atomics, to their run-to-run noise),
the sparse form must leave most of the saved grid unwritten, and the reference-convention cases must hold in both."""
import ctypes

import numpy as np
import pytest
import torch

import dpc_amd
import parity_cases


@pytest.fixture
def lib(emu):
    lib = dpc_amd.get_library()
    yield lib
    lib.dpc_set_chunk_sparse(-1)


def _forward_raw(lib, D, K, N, B=2, radius=0.2, dev="cpu", pc=None, pose=None, sigma=0.9):
    """dpc_project_forward on caller buffers whose saved grid starts as NaN: what stays NaN was never written"""
    rng = np.random.default_rng(1)
    if pc is None:
        pc = torch.tensor((rng.normal(size=(B, N, 3)) * radius / 2).clip(-radius, radius).astype(np.float32), device=dev)
        pose = torch.tensor(rng.normal(size=(B, 4)).astype(np.float32), device=dev)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    taps = [k.reshape(-1).contiguous() for k in dpc_amd.smoothing_kernel(cfg, sigma, device=dev)]
    S = dpc_amd._capi.DpcShape(B, N, D, D, K, K, K)
    P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0)
    z = lambda *s, **kw: torch.zeros(*s, device=dev, **kw)
    tr_pc, cmask = z(B, N, 3), z(B, N, 4, dtype=torch.uint8)
    pindex = z(lib.dpc_point_index_ints(ctypes.byref(S)), dtype=torch.int32)
    grid = torch.full((B, D, D, D), float("nan"), device=dev)
    sums, proj, depth = z(B, D, D, 2, dtype=torch.float64), z(B, D, D), z(B, D, D)
    nws = lib.dpc_workspace_bytes(ctypes.byref(S), 0)
    ws = torch.empty(nws + 256, dtype=torch.uint8, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = None if dev == "cpu" else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.dpc_project_forward(stream, ctypes.byref(S), ctypes.byref(P), p(pc), p(pose), None, None, None, p(taps[0]), p(taps[1]),
                                 p(taps[2]), p(tr_pc), None, p(cmask), p(pindex), p(grid), p(sums), p(proj), p(depth),
                                 ctypes.c_void_p((ws.data_ptr() + 255) & ~255), nws)
    lib.check(rc, "dpc_project_forward")
    if dev != "cpu":
        torch.cuda.synchronize()
    return grid.cpu().numpy(), proj.cpu().numpy(), depth.cpu().numpy()


def sparse_form_writes_fewer_chunks_and_the_same_images(lib, D, K, N, dev="cpu"):
    """The saved grid, dense against sparse form.  Up to 19 taps (21 on rows up to 128 wide) it is the xy-blurred grid: what the
    sparse form stored must be the dense form's values, what it skipped must be zeros there.  Beyond it is G2, in BOTH forms stored through
    the mask 'within K/2 planes of a part of the xy grid that is there' (planes without mass nearby / chunks without a mark
    nearby): the stored part must agree, and the sparse form stores less."""
    assert lib.dpc_set_chunk_sparse(0) in (-1, 0, 1)
    gd, pd_, dd = _forward_raw(lib, D, K, N, dev=dev)
    assert lib.dpc_set_chunk_sparse(1) == 0
    gs, ps, ds = _forward_raw(lib, D, K, N, dev=dev)
    lib.dpc_set_chunk_sparse(-1)
    assert np.array_equal(pd_, ps) and np.array_equal(dd, ds)            # images bit for bit
    w = ~np.isnan(gs)
    assert np.array_equal(gd[w], gs[w])                                  # what it wrote is what the dense form writes there
    assert np.all((gd[~w] == 0) | np.isnan(gd[~w]))                      # ... and what it skipped are zeros (or unwritten planes)
    assert np.isnan(gs).mean() > np.isnan(gd).mean() + 0.2               # a good part of the grid is never stored


def both_forms_agree_bit_for_bit(lib, dev, B, N, D, K, sigma, seed=7, Dz=-1):
    """product API, forward + every gradient, chunk-sparse forced off and on"""
    inp = dpc_amd.synthetic.make_inputs(B, N, seed)
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    gt = torch.tensor(dpc_amd.synthetic.disk_gt(B, D), device=dev)
    res = []
    for mode in (0, 1):
        lib.dpc_set_chunk_sparse(mode)
        t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
        pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
        g = torch.autograd.grad(out["proj"], [pc, pose, scale], ((out["proj"] - gt) / B).detach())
        res.append([out["proj"].detach().cpu().numpy(), out["proj_depth"].detach().cpu().numpy()] + [x.cpu().numpy() for x in g])
    lib.dpc_set_chunk_sparse(-1)
    for a, b, name in zip(res[0], res[1], ("proj", "depth", "dpc", "dpose", "dscale")):
        if name in ("dpose", "dscale"):
            # (the per-view pose sums of several work-groups go through float atomics: two runs of the SAME form already
            # differ in the last places, on the device and -- its threads are OS threads -- in the emulation; the quaternion
            # Jacobian then works on differences of those sums: 5e-6 of the largest component seen at 8000 points)
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(a).max(), 1e-30), (name, float(np.abs(a - b).max()))
        else:
            assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))


@pytest.mark.parametrize("D,K,N", [(64, 5, 300), (128, 11, 400), (64, 21, 300), (64, 23, 300)])
def test_emu_sparse_form_writes_fewer_chunks_and_the_same_images(lib, D, K, N):
    sparse_form_writes_fewer_chunks_and_the_same_images(lib, D, K, N)


@pytest.mark.parametrize("case", [(2, 300, 32, 5, 0.9), (2, 500, 64, 9, 1.4), (1, 400, 64, 21, 3.0)])
def test_emu_both_forms_agree_bit_for_bit(lib, poison_mode, case):
    both_forms_agree_bit_for_bit(lib, "cpu", *case)


def test_emu_both_forms_agree_on_a_256_wide_grid(lib):
    """256-wide rows: a wavefront of the z kernels covers HALF a row (chunks 0-3 or 4-7 of its eight); shallow (32 planes: one
    partly filled 64-plane block of the byte map) to keep the emulation quick"""
    both_forms_agree_bit_for_bit(lib, "cpu", 1, 600, 256, 5, 1.0, Dz=32)


@pytest.mark.parametrize("mode", [0, 1])
def test_emu_reference_conventions_hold_in_both_forms(lib, poison_mode, mode):
    """knife edges (corner cells of weight exactly 0 still carry a gradient: the chunk flags are geometry, not values) with the
    layout forced either way; forced sparse also dropout, the fused loss, a dense-gather plane (the dense form of those is what
    the other emulation tests run wherever the rule says so)"""
    lib.dpc_set_chunk_sparse(mode)
    parity_cases.knife_edge_inputs_match_reference_conventions("cpu", 32, 33)
    if mode == 1:
        parity_cases.fused_dropout_equals_explicit_subset("cpu", extras=False)
        parity_cases.fused_candidate_loss_equals_the_image_epilogue("cpu", N=100)
        parity_cases.fused_path_against_numpy_oracle("cpu", *parity_cases.DENSE_GATHER_CASE_EMU)


def test_bench_counts_the_chunks_the_kernels_mark(lib):
    """bench.py's byte model for the chunk-sparse layout counts marked chunks on the host (plane_occupancy): they must be the
    chunks the kernels really store (the 128-byte chunks of a NaN-initialised saved grid that came back written) -- where the saved
    grid is the xy-blurred one these are k_splat_xy's marks, where it is G2 those marks widened by K/2 planes"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for cid, D, K, N, sigma in ((97, 64, 5, 300, 0.9), (98, 32, 7, 200, 1.2), (95, 64, 23, 200, 0.9)):
        dpc_amd.synthetic.CONFIGS[cid] = dict(B=2, N=N, D=D, K=K, sigma=sigma)
        case = bench.build_case(cid, None, torch.device("cpu"))
        lib.dpc_set_chunk_sparse(1)
        live, nvalid, chunks, chunks_g2 = bench.plane_occupancy(case, K, True)
        grid, _, _ = _forward_raw(lib, D, K, N, pc=case["pc"].detach(), pose=case["pose"].detach(), sigma=sigma)
        written = ~np.isnan(grid.reshape(2, D, D, D // 32, 32))
        assert np.all(written.all(-1) == written.any(-1))              # chunks are stored whole
        expect = chunks if lib.saves_xy(2, N, D, K) else chunks_g2          # (xy grid: up to 19 taps, 21 on rows <= 128 wide)
        assert int(written.any(-1).sum()) == expect, (D, K, int(written.any(-1).sum()), chunks, chunks_g2)
        assert chunks <= chunks_g2 < 2 * D * D * D // 32 and live <= 2 * D and nvalid <= 2 * N


def test_the_rule_and_the_switch(lib):
    """dpc_set_chunk_sparse returns the previous mode; -1 is the per-shape rule (short filters against the grid width)"""
    assert lib.dpc_set_chunk_sparse(1) == -1 and lib.dpc_set_chunk_sparse(0) == 1 and lib.dpc_set_chunk_sparse(-1) == 0
    assert lib.dpc_set_chunk_sparse(-7) == -1 and lib.dpc_set_chunk_sparse(-1) == -1


# ---- the same on the device ------------------------------------------------------------------------------------------
@pytest.fixture
def gpu_lib():
    dpc_amd._capi.set_library(None)
    lib = dpc_amd.get_library()
    assert lib.path.endswith("libdpc_hip.so") and not lib.host_memory
    yield lib
    lib.dpc_set_chunk_sparse(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("D,K,N", [(64, 5, 300), (128, 11, 4000), (256, 11, 8000)])
def test_gpu_sparse_form_writes_fewer_chunks_and_the_same_images(gpu_lib, D, K, N):
    sparse_form_writes_fewer_chunks_and_the_same_images(gpu_lib, D, K, N, dev="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(32, 8000, 128, 11, 1.6),        # BASELINE configs[1]
                                  (40, 8000, 64, 21, 0.8),         # the training shape late in the sigma schedule (9 taps run)
                                  (40, 8000, 64, 21, 3.0),         # ... and early (21 taps: the rule keeps the dense form)
                                  (4, 16000, 256, 11, 2.0),        # configs[4]'s grid
                                  (8, 560, 64, 5, 0.9)])
def test_gpu_both_forms_agree_bit_for_bit(gpu_lib, poison_mode, case):
    both_forms_agree_bit_for_bit(gpu_lib, "cuda", *case)
