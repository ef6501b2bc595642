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
        for walk in (0, 2):          # (2: the walking instantiations wherever they are compiled -- the rule alone starts at 128-wide rows)
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
    (1, 200, 64, 21, 3.0, -1, False, 0),       # dense layout, 21 taps: no walking instantiation (both settings run the same kernels)
    (1, 200, 64, 23, 1.2, -1, False, 1),       # G2 saved through the widened mask: ditto
    (1, 400, 128, 11, 1.6, 32, False, -1),     # the headline's row width, shallow to keep the emulation quick
    (1, 300, 96, 7, 1.0, 24, False, -1),       # padded rows: no chunk maps, plane occupancy only
]


@pytest.mark.parametrize("case", WALK_CASES_EMU)
def test_emu_both_walks_agree_bit_for_bit(emu, poison_mode, case):
    B, N, D, K, sigma, Dz, depth, cs = case
    emu.dpc_emu_dead_groups_take()
    emu.dpc_emu_deals_take()
    both_walks_agree_bit_for_bit(emu, "cpu", B, N, D, K, sigma, Dz=Dz, with_depth=depth, cs=cs)
    dead = emu.dpc_emu_dead_groups_take()
    assert dead > 0 or K > 11, "no wavefront skipped a group: the case does not exercise the sparse walk"   # (the walk is compiled up to 11 taps)
    # 128-wide chunk-sparse rows run the 1024-thread form: its wavefronts re-deal the work-group's tiles by cost (zdeal_tiles)
    assert (emu.dpc_emu_deals_take() > 0) == (D == 128), "the tile deal ran where it should not, or not where it should"


def test_emu_sparse_walk_against_the_numpy_oracle(emu):
    """the sparse walk (default on) through the fused path against the float64 oracle, incl. depth upstream, translation, focal"""
    import parity_cases
    emu.dpc_emu_dead_groups_take()
    emu.dpc_set_sparse_walk(2)
    try:
        parity_cases.fused_path_against_numpy_oracle("cpu", 2, 300, 64, 64, 5, 0.9, True, True)
    finally:
        emu.dpc_set_sparse_walk(1)
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


# ---- the depth sort's three forms (one work-group per view / two launches / several work-groups per view in one launch) ----------
_SORT_SCRIPT = r'''
import ctypes, hashlib, json, os, sys
import numpy as np
import torch
root = sys.argv[1]; dev = sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import dpc_amd
if dev == "cpu":
    dpc_amd._capi.set_library(dpc_amd._capi.DpcLibrary(os.path.join(root, "tests", "hipemu", "libdpc_emu.so"), host_memory=True))
lib = dpc_amd.get_library()
B, N, D, K = 3, 2500, 32, 5
inp = dpc_amd.synthetic.make_inputs(B, N, 31)
pc, pose = torch.tensor(inp["pc"], device=dev), torch.tensor(inp["pose"], device=dev)
cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
taps = [k.reshape(-1).contiguous() for k in dpc_amd.smoothing_kernel(cfg, 0.9, device=dev)]
S = dpc_amd._capi.DpcShape(B, N, D, D, K, K, K)
P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0)
z = lambda *s, **kw: torch.zeros(*s, device=dev, **kw)
tr_pc, cmask = z(B, N, 3), z(B, N, 4, dtype=torch.uint8)
pindex = torch.full((lib.dpc_point_index_ints(ctypes.byref(S)),), -1, dtype=torch.int32, device=dev)
grid = z(B, D, D, D); sums, proj, depth = z(B, D, D, 2, dtype=torch.float64), z(B, D, D), z(B, D, D)
nws = lib.dpc_workspace_bytes(ctypes.byref(S), 0)
ws = torch.full((nws + 256,), 255, dtype=torch.uint8, device=dev)
p = lambda x: ctypes.c_void_p(x.data_ptr())
stream = None if dev == "cpu" else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = lib.dpc_project_forward(stream, ctypes.byref(S), ctypes.byref(P), p(pc), p(pose), None, None, None, p(taps[0]), p(taps[1]), p(taps[2]),
                             p(tr_pc), None, p(cmask), p(pindex), p(grid), p(sums), p(proj), p(depth), ctypes.c_void_p((ws.data_ptr() + 255) & ~255), nws)
lib.check(rc, "dpc_project_forward")
if dev != "cpu":
    torch.cuda.synchronize()
pi = pindex.cpu().numpy(); tr = tr_pc.cpu().numpy()
srec = pi[:4 * B * N].view(np.float32).reshape(B, N, 4); slot_of = pi[4 * B * N:5 * B * N].reshape(B, N)
zstart = pi[5 * B * N:5 * B * N + B * (D + 2)].reshape(B, D + 2)
ok = True
for b in range(B):
    n = srec[b, :, 3].view(np.int32)
    ok &= bool(np.array_equal(np.sort(n), np.arange(N)))                     # every point exactly once
    ok &= bool(np.array_equal(slot_of[b][n], np.arange(N)))                   # slot_of is the inverse map
    ok &= bool(np.array_equal(srec[b, :, :3], tr[b][n]))                      # the records carry the transformed points
    valid = np.all((tr[b] >= -0.5) & (tr[b] <= 0.5), axis=-1)
    cell = np.floor((tr[b][:, 0] + np.float32(0.5)) * np.float32(D - 1)).astype(np.int64).clip(0, D - 1)
    bucket = np.where(valid, cell, D)
    ok &= bool(zstart[b, 0] == 0 and zstart[b, D + 1] == N and np.all(np.diff(zstart[b]) >= 0))
    ok &= bool(np.all((slot_of[b] >= zstart[b][bucket]) & (slot_of[b] < zstart[b][bucket + 1])))   # every point inside its bucket
print(json.dumps({"ok": ok, "proj": hashlib.sha256(proj.cpu().numpy().tobytes()).hexdigest(),
                  "depth": hashlib.sha256(depth.cpu().numpy().tobytes()).hexdigest(), "zstart": zstart.tolist()}))
'''


def _sort_forms(dev):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for form in ("0", "1", "2"):
        env = dict(os.environ, DPC_ZSORT_SPLIT=form, DPC_TEST_HOOKS="1")
        env.pop("DPC_POISON_BUFFERS", None)
        out = subprocess.run([sys.executable, "-c", _SORT_SCRIPT, root, dev], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[form] = json.loads(out.stdout.strip().splitlines()[-1])
        assert res[form]["ok"], "form %s: the sorted records are not a bucket sort of the view's points" % form
    # the same buckets, and -- the splat adds integers -- bitwise the same images whatever the order inside a bucket
    assert res["0"]["zstart"] == res["1"]["zstart"] == res["2"]["zstart"]
    assert res["0"]["proj"] == res["1"]["proj"] == res["2"]["proj"] and res["0"]["depth"] == res["2"]["depth"]


def test_emu_depth_sort_three_forms_agree(emu):
    """DPC_ZSORT_SPLIT=0 / 1 / 2: k_zsort, k_zhist + k_zscatter, k_zsort_multi (round 6: G work-groups per view meeting through
    a zeroed row of counters, one launch) on a case with three work-groups per view (N = 2500), each in its own process
    (the switch is read once): a valid bucket sort each, identical bucket starts, bitwise identical images."""
    _sort_forms("cpu")


@pytest.mark.gpu
def test_gpu_depth_sort_three_forms_agree(gpu_lib):
    _sort_forms("cuda")


# ---- the generic path's plane blur beyond 21 taps (k_blur_plane; round 5 compiled the LDS-free stream form there: 1 KB of scratch) ----
def plane_blur_beyond_21_taps(dev, D, K):
    """smoothen_voxels3d (dpc/util/point_cloud.py:139-145) as its own stage on a power-of-two grid -- the shape family the LDS-free
    stream kernel serves up to 21 taps -- with 25 / 31 taps: forward and the adjoint (the gradient w.r.t. the input grid) against
    the float64 NumPy oracle."""
    B, sigma = 2, K / 6.0
    rng = np.random.default_rng(K)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    taps = onp.smoothing_taps(D, -1, K, sigma)
    vox_np = rng.uniform(0.0, 1.0, (B, D, D, D)).astype(np.float32)
    vox = torch.tensor(vox_np[..., None], device=dev, requires_grad=True)
    sm = dpc_amd.smoothen_voxels3d(cfg, vox, kern)
    ref = onp.blur3d(vox_np.astype(np.float64), taps)
    assert np.abs(sm.detach().cpu().numpy()[..., 0] - ref).max() <= 2e-6
    w = rng.standard_normal((B, D, D, D))
    g, = torch.autograd.grad(sm, [vox], torch.tensor(w[..., None], dtype=torch.float32, device=dev))
    gref = onp.blur3d(w, [t[::-1] for t in taps], order=("z", "y", "x"))
    assert np.abs(g.cpu().numpy()[..., 0] - gref).max() <= 1e-5 * np.abs(gref).max()


@pytest.mark.parametrize("D,K", [(32, 25), (16, 31)])
def test_emu_plane_blur_beyond_21_taps(emu, D, K):
    plane_blur_beyond_21_taps("cpu", D, K)


@pytest.mark.gpu
@pytest.mark.parametrize("D,K", [(32, 25), (64, 31), (128, 27)])
def test_gpu_plane_blur_beyond_21_taps(gpu_lib, D, K):
    plane_blur_beyond_21_taps("cuda", D, K)


def test_graph_replay_refuses_a_frozen_gt_switch():
    """pc_gauss_filter_gt_switch_off is a host decision on sigma; a record-once caller would freeze it into the
    graph.  enable_graph_replay(follow_tap_counts=False) raises for that configuration (before it looks at the device)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "chair_unsupervised"))
    cfg = dpc_amd.default_config(vox_size=32, pc_gauss_kernel_size=5, pc_gauss_filter_gt=True, pc_gauss_filter_gt_switch_off=True)
    m = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
    with pytest.raises(ValueError, match="pc_gauss_filter_gt_switch_off"):
        m.enable_graph_replay()


# ---- the tiles of a 1024-thread work-group re-dealt to its wavefronts by cost (zdeal_tiles) ------------------------------------------
_DEAL_SCRIPT = r'''
import hashlib, json, os, sys
root, dev = sys.argv[1], sys.argv[2]
sys.path[:0] = [root, os.path.join(root, "tests")]
import numpy as np, torch
import dpc_amd
from helpers import synth
if dev == "cpu":
    dpc_amd._capi.set_library(dpc_amd._capi.DpcLibrary(os.path.join(root, "tests", "hipemu", "libdpc_emu.so"), host_memory=True))
B, N, D, K, sigma, Dz = (1, 400, 128, 11, 1.6, 32) if dev == "cpu" else (4, 8000, 256, 11, 2.0, -1)
inp = synth.make_inputs(B, N, 33)
inp["pc"][:, :, 1] *= np.float32(0.5)                    # (a lopsided cloud: the tiles of a work-group cost differently)
cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K)
kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
w = torch.tensor(np.random.default_rng(6).standard_normal(tuple(out["proj"].shape)).astype(np.float32), device=dev)
g = torch.autograd.grad(out["proj"], [pc, pose, scale], w)
h = lambda x: hashlib.sha256(np.ascontiguousarray(x.detach().cpu().numpy()).tobytes()).hexdigest()
print(json.dumps({"proj": h(out["proj"]), "depth": h(out["proj_depth"]), "dpc": h(g[0]), "nonzero": bool(g[0].abs().max() > 0),
                  "dpose": g[1].cpu().numpy().astype(np.float64).ravel().tolist(), "dscale": g[2].cpu().numpy().astype(np.float64).ravel().tolist()}))
'''


def _deal_forms(dev):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for form in ("0", "1", "2"):
        env = dict(os.environ, DPC_ZDEAL=form, DPC_TEST_HOOKS="1")
        out = subprocess.run([sys.executable, "-c", _DEAL_SCRIPT, root, dev], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        res[form] = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["0"]["nonzero"]
    for form in ("1", "2"):
        for k in ("proj", "depth", "dpc"):          # which wavefront walks a tile changes nothing a ray computes
            assert res[form][k] == res["0"][k], (form, k)
        for k in ("dpose", "dscale"):               # per-view sums: the order of the work-group's partials / float atomics
            a, b = np.array(res["0"][k]), np.array(res[form][k])
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(a).max(), 1e-30), (form, k)


def test_emu_tile_deal_changes_no_ray():
    """DPC_ZDEAL=0 / 1 / 2 (read once: a process each): the Latin-square tiles as they come / k_zbwd's wavefronts re-deal their
    work-group's 16 tiles by cost (k_zfwd from 256-wide rows) / both kernels wherever they run the 1024-thread form.  Images, depth
    and point gradients bit for bit; the per-view pose / scale sums to their summation-order noise."""
    _deal_forms("cpu")


@pytest.mark.gpu
def test_gpu_tile_deal_changes_no_ray(gpu_lib):
    _deal_forms("cuda")


# ---- GradBuckets: where the division by the world size happens (average=...) ---------------------------------------------------
def test_grad_buckets_average_policy(monkeypatch):
    """average="auto" | "collective" | "divide" and the DPC_BUCKET_AVG override: without a process group nothing is averaged in a
    collective whatever is asked; a bad value is refused.  (Under RCCL: tests/test_gpu_parity.py, the one-rank group -- "auto" sends a
    one-rank average out as a SUM, "collective" as RCCL's AVG; gloo sums and divides: tests/test_distributed.py.)"""
    net = torch.nn.Linear(4, 3)
    with pytest.raises(ValueError, match="average"):
        dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=1, average="mean")
    for avg in ("auto", "collective", "divide"):
        net = torch.nn.Linear(4, 3)
        red = dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=1, gather="copy", average=avg)
        assert red.average == avg and red.in_collective_average is False and red._avg_op is False
        net(torch.ones(2, 4)).sum().backward()
        red.finish()
        assert torch.equal(net.weight.grad, torch.full((3, 4), 2.0))
    monkeypatch.setenv("DPC_BUCKET_AVG", "0")
    assert dpc_amd.distributed.GradBuckets(torch.nn.Linear(2, 2).parameters(), average="collective").average == "divide"
    monkeypatch.setenv("DPC_BUCKET_AVG", "1")
    assert dpc_amd.distributed.GradBuckets(torch.nn.Linear(2, 2).parameters()).average == "collective"
