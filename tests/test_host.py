"""CPU tier for the host side: the C-ABI library loads and exports every symbol
include/dpc_hip.h declares, argument validation matches the reference's error
behaviour, the product path refuses to run without its HIP library or on CPU
tensors (no fallback), and the lazily materialised outputs behave like the
reference's dict."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import dpc_amd
from dpc_amd import _capi
from helpers import ROOT, load, maxabs


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dpc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert "dpc_project_forward" in decl and "dpc_project_backward" in decl
    assert sorted(_capi.SIGNATURES) == decl


def test_hip_library_loads_and_exports_every_declared_symbol():
    """No compute calls (there is no GPU here): load + symbol presence only."""
    assert os.path.exists(_capi.LIB_PATH), "run `python __graft_entry__.py` (build) first"
    dll = ctypes.CDLL(_capi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(dll, name), name
    lib = _capi.DpcLibrary(_capi.LIB_PATH)
    assert "gfx950" in lib.version() and not lib.host_memory


def test_struct_mirrors_match_the_library_layout():
    """The ctypes mirrors of DpcShape / DpcParams have the sizes the library was built with (the loader
    refuses a mismatch), and the nullable device pointer added in round 2 sits where C puts it."""
    lib = _capi.get_library()
    assert lib.dpc_abi_struct_bytes(0) == ctypes.sizeof(_capi.DpcShape) == 28
    assert lib.dpc_abi_struct_bytes(1) == ctypes.sizeof(_capi.DpcParams) == 120
    assert lib.dpc_abi_struct_bytes(2) == 0
    assert _capi.DpcParams.dropout_state.offset == 40
    assert (_capi.DpcParams.l2_target.offset, _capi.DpcParams.l2_grad.offset, _capi.DpcParams.l2_weight.offset) == (48, 56, 64)


def test_hip_library_contains_gfx950_code_object():
    blob = open(_capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_zfwd" in blob and b"k_blur_plane" in blob


def test_c_abi_argument_validation_without_gpu():
    """Status codes are produced before any launch, so they can be checked here."""
    lib = _capi.DpcLibrary(_capi.LIB_PATH)
    S = _capi.DpcShape(2, 10, 8, 8, 4, 5, 5)           # even Kx
    P = _capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0)
    assert lib.dpc_workspace_bytes(ctypes.byref(S), 0) == 0
    S = _capi.DpcShape(2, 10, 8, 8, 5, 5, 5)
    g = 2 * 8 * 8 * 8 * 4
    assert lib.dpc_workspace_bytes(ctypes.byref(S), 0) >= (g + 255) // 256 * 256
    assert lib.dpc_workspace_bytes(ctypes.byref(S), 0) % 256 == 0
    assert lib.dpc_saved_layout(ctypes.byref(S), ctypes.byref(P)) == 1      # D=8: generic path, dense grid_raw
    S64 = _capi.DpcShape(2, 10, 64, 64, 11, 11, 11)
    # fused path: clip_mask + point_index; xy grid saved; 11 taps on 64-wide rows: chunk-sparse by the rule (64-wide rows: K <= D/8 + 5)
    assert lib.dpc_saved_layout(ctypes.byref(S64), ctypes.byref(P)) == 6 | 8 | 16
    assert lib.dpc_set_chunk_sparse(0) == -1 and lib.dpc_saved_layout(ctypes.byref(S64), ctypes.byref(P)) == 6 | 8
    assert lib.dpc_set_chunk_sparse(-1) == 0
    # records + inverse map + bucket starts + plane-occupancy words, and (rows of whole 32-ray words) the chunk maps:
    # a byte per (view, plane, row), once by plane and once by row
    assert lib.dpc_point_index_ints(ctypes.byref(S64)) == 5 * 2 * 10 + 2 * 66 + 2 * 8 + 2 * (2 * 64 * 64 // 4)
    S48 = _capi.DpcShape(2, 10, 48, 48, 5, 5, 5)                               # 48 on the 64-wide geometry: no chunk maps
    assert lib.dpc_point_index_ints(ctypes.byref(S48)) == 5 * 2 * 10 + 2 * 50 + 2 * 8
    S21 = _capi.DpcShape(2, 10, 64, 64, 21, 21, 21)
    assert lib.dpc_saved_layout(ctypes.byref(S21), ctypes.byref(P)) == 6 | 8   # 21 taps on rows up to 128 wide: still the xy grid
    S23 = _capi.DpcShape(2, 10, 64, 64, 23, 23, 23)
    assert lib.dpc_saved_layout(ctypes.byref(S23), ctypes.byref(P)) == 6       # beyond: G2 is what is saved (64-wide: dense layout)
    W21 = _capi.DpcShape(1, 10, 32, 256, 21, 21, 21)
    assert lib.dpc_saved_layout(ctypes.byref(W21), ctypes.byref(P)) == 6 | 16  # 21 taps on 256-wide rows: G2, chunk-sparse
    Sdeep = _capi.DpcShape(1, 10, 320, 32, 5, 5, 5)                            # Dz > 256: beyond the plane-occupancy words
    assert lib.dpc_saved_layout(ctypes.byref(Sdeep), ctypes.byref(P)) == 1
    assert lib.dpc_workspace_bytes(ctypes.byref(S), 1) >= 2 * g
    null = None
    rc = lib.dpc_project_forward(null, ctypes.byref(S), ctypes.byref(P), *([null] * 16), null, 0)
    assert rc == -1                                      # DPC_E_NULL
    rc = lib.dpc_voxelize_fwd(null, ctypes.byref(_capi.DpcShape(0, 1, 8, 8, 0, 0, 0)), null, null)
    assert rc == -2                                      # DPC_E_SHAPE
    rc = lib.dpc_blur3d(null, ctypes.byref(_capi.DpcShape(1, 1, 8, 8, 0, 0, 65)), null, null, null, null, null, null, 0)
    assert rc == -3                                      # DPC_E_TAPS
    assert lib.dpc_silhouette_loss_fwd(null, 6, 4, 8, 8, *([null] * 7)) == -2     # B not a multiple of C
    assert lib.dpc_silhouette_loss_fwd(null, 8, 4, 8, 4, *([null] * 7)) == -2     # GT smaller than the prediction
    assert lib.dpc_silhouette_loss_fwd(null, 8, 4, 8, 8, *([null] * 7)) == -1
    assert lib.dpc_silhouette_loss_bwd(null, 8, 4, 8, 8, *([null] * 5)) == -1
    # the fused dropout is refused (not silently ignored) off the fused path; point_index must be 16-byte aligned
    one = ctypes.c_void_p(256)
    Pd = _capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 3, 7)
    rc = lib.dpc_project_forward(null, ctypes.byref(_capi.DpcShape(1, 10, 18, 18, 5, 5, 5)), ctypes.byref(Pd), one, one, null,
                                 null, null, one, one, one, one, one, null, null, one, one, one, null, one, 10 ** 9)
    assert rc == -5                                      # DPC_E_MODE
    rc = lib.dpc_project_forward(null, ctypes.byref(_capi.DpcShape(1, 10, 32, 32, 5, 5, 5)), ctypes.byref(P), one, one, null,
                                 null, null, one, one, one, one, null, one, ctypes.c_void_p(260), one, one, one, null, one,
                                 10 ** 9)
    assert rc == -4                                      # DPC_E_WORKSPACE (misaligned point_index)
    # fused L2 epilogue: DRC collapse only, and it needs somewhere to write
    Pl = _capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 1, 0, 0, 0, None, 256, 256, 1.0)      # max collapse + l2_target
    fwd = lambda Pp: lib.dpc_project_forward(null, ctypes.byref(_capi.DpcShape(1, 10, 32, 32, 5, 5, 5)), ctypes.byref(Pp),
                                             one, one, null, null, null, one, one, one, one, null, one, one, one, one,
                                             one, null, one, 10 ** 9)
    assert fwd(Pl) == -5                                 # DPC_E_MODE
    Pl = _capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0, None, 256, None, 1.0)     # l2_target without l2_grad
    assert fwd(Pl) == -1                                 # DPC_E_NULL
    with pytest.raises(_capi.DpcError):
        lib.check(-4, "x")


def test_product_path_has_no_cpu_fallback():
    prev = _capi.set_library(None)
    try:
        cfg = dpc_amd.default_config(vox_size=8)
        pc = torch.zeros(1, 4, 3)
        q = torch.tensor([[1.0, 0, 0, 0]])
        with pytest.raises(ValueError, match="ROCm device"):
            dpc_amd.pointcloud_project_fast(cfg, pc, q, None, None)
        with pytest.raises(ValueError, match="ROCm device"):
            dpc_amd.pointcloud2voxels3d_fast(cfg, pc, None)
    finally:
        _capi.set_library(prev)


def test_missing_library_fails_loudly(monkeypatch):
    prev = _capi.set_library(None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libdpc_hip.so")
    try:
        with pytest.raises(_capi.DpcError, match="not built"):
            _capi.get_library()
    finally:
        _capi.set_library(prev)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "differentiable-point-clouds_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                for line in src.splitlines():
                    ls = line.strip()
                    if ls.startswith(("import ", "from ")):
                        assert "oracle" not in ls and "hipemu" not in ls and "tf_shim" not in ls, (f, ls)


# ---- argument checking mirrors the reference's graph-build-time errors ---------------
def test_shape_and_type_errors(emu):
    cfg = dpc_amd.default_config(vox_size=8, pc_gauss_kernel_size=3)
    pc = torch.zeros(2, 5, 3)
    with pytest.raises(ValueError, match="last dimension must be 4"):       # quaternion.py:22-29
        dpc_amd.pointcloud_project_fast(cfg, pc, torch.zeros(2, 3), None, None)
    with pytest.raises(ValueError):
        dpc_amd.pointcloud_project_fast(cfg, torch.zeros(2, 5, 2), torch.ones(2, 4), None, None)
    with pytest.raises(TypeError):
        dpc_amd.pointcloud_project_fast(cfg, pc.double(), torch.ones(2, 4).double(), None, None)
    with pytest.raises(ValueError):                                            # rgb must match the cloud
        dpc_amd.pointcloud_project_fast(cfg, pc, torch.ones(2, 4), None, torch.zeros(2, 4, 3))
    cfg_m = dpc_amd.default_config(vox_size=8, pose_quaternion=False)
    with pytest.raises(ValueError, match="quaternion pose"):                  # point_cloud.py:211-213
        dpc_amd.pointcloud_project_fast(cfg_m, pc, torch.eye(4).repeat(2, 1, 1), torch.zeros(2, 3), None)
    with pytest.raises(ValueError, match="even"):
        dpc_amd.gauss_kernel_1d(4, 1.0, device="cpu")
    cfg_bad = dpc_amd.default_config(vox_size=8, pc_separable_gauss_filter=False)
    with pytest.raises(NotImplementedError):                                   # dense 3-D blur kernel
        dpc_amd.pointcloud_project_fast(cfg_bad, pc, torch.ones(2, 4), None, None, torch.ones(5, 5, 5, 1, 1))
    with pytest.raises(KeyError):
        dpc_amd.default_config(no_such_key=1)


def test_empty_cloud_and_all_outliers(emu):
    """Ragged/empty input: every point outside the cube -> background image,
    zero point gradients."""
    cfg = dpc_amd.default_config(vox_size=8, pc_gauss_kernel_size=3)
    pc = torch.full((1, 6, 3), 0.9, requires_grad=True)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], requires_grad=True)
    out = dpc_amd.pointcloud_project_fast(cfg, pc, q, None, None, dpc_amd.smoothing_kernel(cfg, 0.7, device="cpu"))
    bg = 1 - (1 - 1e-5) ** 8
    assert abs(float(out["proj"].max()) - bg) < 1e-7 and abs(float(out["proj"].min()) - bg) < 1e-7
    out["proj"].sum().backward()
    assert float(pc.grad.abs().max()) == 0.0


def test_lazy_outputs_dict_semantics(emu):
    g = load("tiny")
    from run_case import product_cfg
    cfg = product_cfg("tiny", g)
    t = lambda k: torch.tensor(g[k])
    kern = dpc_amd.smoothing_kernel(cfg, 0.8, device="cpu")
    out = dpc_amd.pointcloud_project_fast(cfg, t("pc"), t("pose"), t("trans"), None, kern, scaling_factor=t("scale"))
    assert set(out.keys()) == {"proj", "voxels", "tr_pc", "voxels_rgb", "proj_rgb", "drc_probs", "proj_depth"}
    assert dict.__getitem__(out, "voxels") is None                  # not computed yet
    v = out["voxels"]
    assert v.shape == (2, 16, 16, 16, 1) and out["voxels"] is v     # computed once, cached
    assert maxabs(v.numpy(), g["voxels_f64"]) < 2e-5
    assert out["drc_probs"].shape == (17, 2, 16, 16, 1)
    assert out["voxels_rgb"] is None and out["proj_rgb"] is None
    assert out.get("proj") is out["proj"] and out.get("nope", 7) == 7


def test_reference_filter_shapes_pick_the_axis(emu):
    """smoothen_voxels3d reads the blur axis off the filter shape, so a caller
    may pass the reference's [k_x,k_y,k_z] list in any order or a subset."""
    cfg = dpc_amd.default_config(vox_size=8, pc_gauss_kernel_size=3)
    vox = torch.rand(1, 8, 8, 8, 1)
    kx, ky, kz = dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu")
    assert kx.shape == (1, 1, 3, 1, 1) and ky.shape == (1, 3, 1, 1, 1) and kz.shape == (3, 1, 1, 1, 1)
    a = dpc_amd.smoothen_voxels3d(cfg, vox, [kx, ky, kz])
    b = dpc_amd.smoothen_voxels3d(cfg, vox, [kz, kx, ky])
    assert float((a - b).abs().max()) < 1e-6
    only_z = dpc_amd.smoothen_voxels3d(cfg, vox, [kz])
    ref = torch.nn.functional.conv3d(vox.permute(0, 4, 1, 2, 3), kz.reshape(1, 1, 3, 1, 1), padding=(1, 0, 0))
    assert float((only_z.permute(0, 4, 1, 2, 3) - ref).abs().max()) < 1e-6


def test_emulation_library_is_refused_without_the_test_switch(monkeypatch, emu_library):
    """A product process (no DPC_TEST_HOOKS) cannot install the host-memory emulation library."""
    monkeypatch.delenv("DPC_TEST_HOOKS", raising=False)
    with pytest.raises(_capi.DpcError, match="DPC_TEST_HOOKS"):
        _capi.set_library(emu_library)


def test_config_edit_counter_sees_every_mutator():
    """util.point_cloud._meta reuses what it derived from a Config until the config's edit counter moves: every way of
    writing into the mapping must move it (|= and popitem bypass dict.update / __setitem__ in CPython)."""
    import dpc_amd
    from dpc_amd.util.point_cloud import _meta
    cfg = dpc_amd.default_config(vox_size=32)
    assert _meta(cfg).D == 32
    cfg |= {"vox_size": 48}
    assert _meta(cfg).D == 48
    cfg.vox_size = 64
    assert _meta(cfg).D == 64
    n = cfg.__dict__["_edits"]
    cfg.popitem()
    assert cfg.__dict__["_edits"] == n + 1
    import gc, weakref
    ref = weakref.ref(cfg)
    del cfg
    gc.collect()
    assert ref() is None          # the cache holds no strong reference


def test_one_tap_filter_is_the_axis_no_other_filter_claims():
    """gauss_kernel.py:35-54 builds a ONE-tap z filter when round(K * vox_size_z / vox_size) is 1 (vox 112 x 32 deep,
    K = 5): its shape [1,1,1,1,1] names no axis; it is the z filter because x and y are taken -- and, being the
    normalised Gaussian [1.0], a pass-through."""
    import torch
    import dpc_amd
    from dpc_amd.util.point_cloud import _filter_axes, _flat_taps_uncached
    for D, Dz, K in ((112, 32, 5), (72, 24, 3)):
        cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K)
        kern = dpc_amd.smoothing_kernel(cfg, 1.0, device="cpu")
        assert [tuple(k.shape[:3]) for k in kern] == [(1, 1, K), (1, K, 1), (1, 1, 1)]
        assert _filter_axes(kern) == ["x", "y", "z"]
        tx, ty, tz = _flat_taps_uncached(kern, torch.device("cpu"))
        assert tx.numel() == K and ty.numel() == K and tz is None
    one = torch.ones(1, 1, 1, 1, 1)
    assert _filter_axes([one, one, one]) == ["x", "y", "z"]
    z3 = torch.ones(3, 1, 1, 1, 1) / 3
    assert _filter_axes([one, z3]) == ["x", "z"]
    import pytest
    with pytest.raises(NotImplementedError):
        _filter_axes([z3, z3, one, one, one])
