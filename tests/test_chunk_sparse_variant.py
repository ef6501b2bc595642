"""The chunk-sparse experiment (csrc built with -DDPC_CHUNK_SPARSE=1, default off): its emulation build against
the shipped one.  Opt-in (DPC_TEST_VARIANTS=1: it compiles a second emulation library, ~30 s); to run the WHOLE emulation tier on the
variant: `make -C tests/hipemu OUT=libdpc_emu_cs.so EXTRA=-DDPC_CHUNK_SPARSE=1` and `DPC_EMU_LIB=libdpc_emu_cs.so pytest tests -m "not gpu"`."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

import dpc_amd
import parity_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu")
pytestmark = pytest.mark.skipif(os.environ.get("DPC_TEST_VARIANTS") != "1", reason="opt-in: DPC_TEST_VARIANTS=1")


@pytest.fixture(scope="module")
def libs():
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU])
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU, "OUT=libdpc_emu_cs.so", "EXTRA=-DDPC_CHUNK_SPARSE=1"])
    mk = lambda n: dpc_amd._capi.DpcLibrary(os.path.join(EMU, n), host_memory=True)
    return mk("libdpc_emu.so"), mk("libdpc_emu_cs.so")


def _forward(lib, D, K, N, B=2, radius=0.2):
    """dpc_project_forward on caller buffers whose saved grid starts as NaN: what stays NaN was never written"""
    prev = dpc_amd._capi.set_library(lib)
    try:
        rng = np.random.default_rng(1)
        pc = torch.tensor((rng.normal(size=(B, N, 3)) * radius / 2).clip(-radius, radius).astype(np.float32))
        pose = torch.tensor(rng.normal(size=(B, 4)).astype(np.float32))
        cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
        taps = [k.reshape(-1).contiguous() for k in dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu")]
        S = dpc_amd._capi.DpcShape(B, N, D, D, K, K, K)
        P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0)
        tr_pc, cmask = torch.zeros(B, N, 3), torch.zeros(B, N, 4, dtype=torch.uint8)
        pindex = torch.zeros(lib.dpc_point_index_ints(ctypes.byref(S)), dtype=torch.int32)
        grid = torch.full((B, D, D, D), float("nan"))
        sums, proj, depth = torch.zeros(B, D, D, 2, dtype=torch.float64), torch.zeros(B, D, D), torch.zeros(B, D, D)
        nws = lib.dpc_workspace_bytes(ctypes.byref(S), 0)
        ws = torch.empty(nws + 256, dtype=torch.uint8)
        p = lambda x: ctypes.c_void_p(x.data_ptr())
        rc = lib.dpc_project_forward(None, ctypes.byref(S), ctypes.byref(P), p(pc), p(pose), None, None, None, p(taps[0]), p(taps[1]),
                                     p(taps[2]), p(tr_pc), None, p(cmask), p(pindex), p(grid), p(sums), p(proj), p(depth),
                                     ctypes.c_void_p((ws.data_ptr() + 255) & ~255), nws)
        lib.check(rc, "dpc_project_forward")
        return grid.numpy(), proj.numpy(), depth.numpy()
    finally:
        dpc_amd._capi.set_library(prev)


@pytest.mark.parametrize("D,K,N", [(64, 5, 300), (128, 11, 400)])
def test_variant_writes_fewer_chunks_and_the_same_images(libs, D, K, N):
    dense, sparse = libs
    gd, pd_, dd = _forward(dense, D, K, N)
    gs, ps, ds = _forward(sparse, D, K, N)
    assert np.array_equal(pd_, ps) and np.array_equal(dd, ds)            # images bit for bit
    w = ~np.isnan(gs)
    assert np.array_equal(gd[w], gs[w])                                  # what it wrote is what the shipped build writes there
    assert np.all((gd[~w] == 0) | np.isnan(gd[~w]))                      # ... and what it skipped are zeros (or unwritten planes)
    assert np.isnan(gs).mean() > np.isnan(gd).mean() + 0.2               # a good part of the grid is never stored


def test_variant_parity_cases(libs):
    """knife edges (corner cells of weight exactly 0 still carry a gradient: the chunk flags are geometry, not values), dropout,
    the fused loss, a dense-gather plane -- on the variant"""
    prev = dpc_amd._capi.set_library(libs[1])
    try:
        parity_cases.knife_edge_inputs_match_reference_conventions("cpu", 32, 33)
        parity_cases.fused_dropout_equals_explicit_subset("cpu", extras=False)
        parity_cases.fused_candidate_loss_equals_the_image_epilogue("cpu", N=100)
        parity_cases.fused_path_against_numpy_oracle("cpu", *parity_cases.DENSE_GATHER_CASE_EMU)
    finally:
        dpc_amd._capi.set_library(prev)
