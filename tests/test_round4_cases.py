"""Round-4 additions to the CPU tier: the sigma-aware effective tap count (host logic + the emulated kernels of every
compiled tap count), the dict conversions of the lazily built output dicts, the slow projector under point dropout,
and the GradBuckets invariants."""
import math

import numpy as np
import pytest
import torch

from helpers import synth


# ---------------------------------------------------------------------------------------------------------------
# effective tap count (dpc/models/model_pc.py:33-38,146-153: sigma is annealed 3.0 -> 0.2 while K stays 21)
# ---------------------------------------------------------------------------------------------------------------
def test_effective_half_width_matches_the_taps():
    """effective_half_width(sigma, h) is the largest offset whose tap is >= 1e-8 of the centre tap, computed from the
    taps themselves in float64 here."""
    from dpc_amd.util import gauss_kernel as gk
    for sigma in (3.0, 2.0, 1.65, 1.64, 1.2, 0.99, 0.98, 0.83, 0.82, 0.5, 0.33, 0.2):
        for half in (10, 5, 2):
            m = np.arange(0, half + 1, dtype=np.float64)
            rel = np.exp(-m * m / (2.0 * sigma * sigma))
            want = int(np.nonzero(rel >= gk.TAP_DROP_REL)[0].max())
            assert gk.effective_half_width(sigma, half) == want, (sigma, half)
    assert gk.effective_half_width(float("nan"), 10) == 10 and gk.effective_half_width(0.0, 10) == 10


def test_filters_carry_their_support_and_trim_to_compiled_counts():
    import dpc_amd
    from dpc_amd.util import point_cloud as pcu
    cfg = dpc_amd.default_config(vox_size=64, pc_gauss_kernel_size=21)
    expect = {3.0: 21, 1.7: 21, 1.6: 19, 1.4: 17, 1.2: 15, 1.0: 13, 0.9: 11, 0.7: 9, 0.5: 7, 0.4: 5, 0.2: 3}
    for sigma, k_eff in expect.items():
        kern = dpc_amd.smoothing_kernel(cfg, sigma, device="cpu")
        assert all(hasattr(k, "dpc_support") for k in kern)
        tx, ty, tz = pcu._flat_taps(cfg, kern, torch.device("cpu"))
        assert (tx.numel(), ty.numel(), tz.numel()) == (k_eff,) * 3, (sigma, tx.numel())
        assert pcu.effective_tap_counts(cfg, kern) == (k_eff,) * 3
        full = kern[0].reshape(-1)
        off = (21 - k_eff) // 2
        # the kept taps are the full filter's own values (a view into its buffer), not a renormalised filter
        assert tx.data_ptr() == full.data_ptr() + 4 * off
        assert torch.equal(tx, full[off:off + k_eff])
        dropped = float(full.sum() - tx.sum())
        assert dropped <= 21 * 1e-8, (sigma, dropped)
    cfg_off = dpc_amd.default_config(vox_size=64, pc_gauss_kernel_size=21, pc_trim_gauss_taps=False)
    kern = dpc_amd.smoothing_kernel(cfg_off, 0.5, device="cpu")
    assert pcu._flat_taps(cfg_off, kern, torch.device("cpu"))[0].numel() == 21
    # a filter that does not come from a host-side sigma is applied in full
    raw = [k.clone() for k in dpc_amd.smoothing_kernel(cfg, 0.5, device="cpu")]
    assert pcu._flat_taps(cfg, raw, torch.device("cpu"))[0].numel() == 21
    # vox_size_z: the z filter has its own size and sigma (gauss_kernel.py:38-50)
    cfgz = dpc_amd.default_config(vox_size=64, vox_size_z=32, pc_gauss_kernel_size=21)
    kz = dpc_amd.smoothing_kernel(cfgz, 1.2, device="cpu")
    assert pcu.effective_tap_counts(cfgz, kz) == (15, 15, 7)        # 11 z taps at sigma 0.6: offsets <= 3 matter


def test_model_tap_counts_follow_the_sigma_schedule():
    """ModelPointCloud.effective_tap_counts over the reference's schedule (sigma 3.0 -> 0.2, K = 21): non-increasing,
    21 for the first ~48 % of the run, <= 11 (the cheaper saved-state layout) for the last ~28 %."""
    import dpc_amd
    cfg = dpc_amd.default_config(vox_size=64, pc_gauss_kernel_size=21, pc_relative_sigma=3.0, pc_relative_sigma_end=0.2,
                                 max_number_of_steps=1000)
    m = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
    ks = []
    for step in range(0, 1001, 10):
        m.set_global_step(step)
        k = m.effective_tap_counts()
        assert k[0] == k[1] == k[2]
        ks.append(k[0])
    assert ks[0] == 21 and ks[-1] == 3 and all(a >= b for a, b in zip(ks, ks[1:]))
    assert abs(sum(k == 21 for k in ks) / len(ks) - 0.483) < 0.02
    assert abs(sum(k <= 11 for k in ks) / len(ks) - 0.282) < 0.02
    assert sorted(set(ks)) == [3, 5, 7, 9, 11, 13, 15, 17, 19, 21]


@pytest.mark.parametrize("sigma", [1.5, 1.2, 0.8, 0.6, 0.3])
def test_emu_trimmed_filter_equals_the_full_one(emu, sigma):
    """The projector with the trimmed filter (19 / 15 / 9 / 7 / 3 taps of a K = 21 configuration, each a different
    compiled kernel set and, at <= 11 taps, the xy-saving state layout) against the full 21 taps: images to 2e-7,
    gradients to 2e-6 of their largest entry."""
    import dpc_amd
    B, N, D = 2, 300, 32
    inp = synth.make_inputs(B, N, 11)
    res = {}
    for trim in (True, False):
        cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=21, pc_trim_gauss_taps=trim)
        kern = dpc_amd.smoothing_kernel(cfg, sigma, device="cpu")
        t = lambda a: torch.tensor(a, requires_grad=True)
        pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
        w = torch.tensor(np.random.default_rng(5).standard_normal((B, D, D, 1)).astype(np.float32))
        wd = torch.tensor(np.random.default_rng(6).standard_normal((B, D, D, 1)).astype(np.float32)) * 0.1
        g = torch.autograd.grad([out["proj"], out["proj_depth"]], [pc, pose, scale], [w, wd])
        res[trim] = (out["proj"].detach(), out["proj_depth"].detach(), g)
    a, b = res[True], res[False]
    assert float((a[0] - b[0]).abs().max()) <= 2e-7
    assert float((a[1] - b[1]).abs().max()) <= 2e-6
    for x, y in zip(a[2], b[2]):
        assert float((x - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max())), sigma


@pytest.mark.parametrize("K", [3, 7, 9, 13, 15, 17, 19, 23, 25, 27, 29, 31])
def test_emu_new_tap_counts_match_the_numpy_oracle(emu, K):
    """Every tap count that gained compiled kernels this round (fused front / back end and z kernels), forward and
    backward against the float64 NumPy oracle."""
    import dpc_amd
    from oracle import dpc_oracle_np as onp
    B, N, D, sigma = 2, 200, 32, 0.35 * K / 2
    inp = synth.make_inputs(B, N, 100 + K)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K, pc_trim_gauss_taps=False)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device="cpu")
    t = lambda a: torch.tensor(a, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    assert dpc_amd.ops.uses_fused_path(dpc_amd.get_library(), B, N, dpc_amd.util.point_cloud._meta(cfg), (K, K, K))
    gt = torch.tensor(synth.disk_gt(B, D))
    dproj = ((out["proj"] - gt) / B).detach()
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], dproj)
    f64 = lambda x: x.astype(np.float64)
    taps = onp.smoothing_taps(D, -1, K, sigma)
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, Dz=D, D=D)
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, fw,
                              dproj=f64(dproj.numpy()))
    assert np.abs(out["proj"].detach().numpy() - fw["proj"]).max() <= 2e-5
    for got, key in zip(g, ("dpc", "dpose", "dscale")):
        ref = bw[key].reshape(got.shape)
        assert np.abs(got.numpy() - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-12), (K, key)


# ---------------------------------------------------------------------------------------------------------------
# ADVICE (round 3)
# ---------------------------------------------------------------------------------------------------------------
def test_lazy_dicts_convert_with_their_real_entries():
    import dpc_amd
    base = {"points_1": torch.arange(12.0).reshape(2, 2, 3)}
    for conv in (dict, lambda d: {**d}, lambda d: d.copy(), lambda d: dict(d.items())):
        out = dpc_amd.model_pc.ReplicatedOutputs(base, 3)
        assert out.points_replication() is not None                      # nothing built yet
        got = conv(out)
        assert got["all_points"] is not None and got["all_points"].shape == (6, 2, 3)
    out = dpc_amd.model_pc.ReplicatedOutputs(base, 3)
    assert out.pop("all_points").shape == (6, 2, 3) and "all_points" not in out
    import pickle
    out = dpc_amd.model_pc.ReplicatedOutputs(base, 2)
    back = pickle.loads(pickle.dumps(dict(out)))
    assert back["all_points"].shape == (4, 2, 3)
    from dpc_amd.util.point_cloud import ProjectionOutputs
    po = ProjectionOutputs({"proj": 1}, lambda: "V", lambda: "P")
    assert dict(po) == {"proj": 1, "voxels": "V", "drc_probs": "P"}
    po = ProjectionOutputs({"proj": 1}, lambda: "V", lambda: "P")
    assert {**po}["voxels"] == "V" and po.copy()["drc_probs"] == "P"


def test_emu_slow_projector_sees_the_dropped_out_cloud(emu):
    """compute_projection with pc_fast=False and pc_point_dropout != 1 projects int(N * keep) points
    (dpc/models/model_pc.py:233-251), not the full cloud."""
    import dpc_amd
    cfg = dpc_amd.default_config(vox_size=16, pc_num_points=40, predict_pose=True, pose_predict_num_candidates=1,
                                 step_size=1, batch_size=2, pc_point_dropout=0.25, pc_fast=False)
    m = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
    inp = synth.make_inputs(2, 40, 3)
    seen = {}
    orig = dpc_amd.model_pc.pointcloud_project

    def spy(cfg_, pts, pose, sigma):
        seen["n"] = pts.shape[1]
        return orig(cfg_, pts, pose, sigma)

    dpc_amd.model_pc.pointcloud_project = spy
    try:
        outputs = {"points_1": torch.tensor(0.4 * inp["pc"]), "poses": torch.tensor(inp["pose"]),
                   "scaling_factor": torch.tensor(inp["scale"]), "rgb_1": None}
        outputs = m.replicate_outputs(outputs)
        m.compute_projection({}, outputs, is_training=True)
    finally:
        dpc_amd.model_pc.pointcloud_project = orig
    assert seen["n"] == int(40 * 0.25)
    assert outputs["projs"].shape == (2, 16, 16, 1)


def test_grad_buckets_refuse_broken_invariants():
    import dpc_amd
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    buckets = dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=1)
    x = torch.randn(5, 4)
    net(x).sum().backward()
    buckets.finish()
    buckets.zero_()
    # a second backward before finish(): the bucket's all-reduce has already gone out
    net(x).sum().backward()
    with pytest.raises(RuntimeError, match="second gradient"):
        net(x).sum().backward()
    buckets._arm()
    buckets.zero_()
    # optimizer.zero_grad() (set_to_none=True): the views are gone
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.zero_grad()
    net(x).sum().backward()
    with pytest.raises(RuntimeError, match="no longer its bucket view"):
        buckets.finish()


def test_grad_buckets_issue_in_bucket_order():
    """buckets complete in any order, the all-reduces go out in bucket order (what keeps ranks with different
    completion orders matched)"""
    import dpc_amd
    ps = [torch.nn.Parameter(torch.zeros(300000)) for _ in range(3)]          # 1.2 MB each: one bucket per parameter
    b = dpc_amd.distributed.GradBuckets(ps, bucket_mb=1)
    assert len(b.buckets) == 3
    order = [b._of[p] for p in ps]                 # reverse parameter order: [2, 1, 0]
    assert order == [2, 1, 0]
    b._hook(ps[0])                                 # bucket 2 completes first: must wait for buckets 0 and 1
    assert b._next == 0
    b._hook(ps[2])                                 # bucket 0
    assert b._next == 1
    b._hook(ps[1])                                 # bucket 1 -> 1 and 2 go out
    assert b._next == 3
    b.finish()
    assert b._next == 0


# ---------------------------------------------------------------------------------------------------------------
# the two bindings of the C ABI: ctypes (ops.ProjectFused) and the compiled one (csrc/dpc_torch.cpp)
# ---------------------------------------------------------------------------------------------------------------
def _project_both_ways(monkeypatch, **kw):
    import dpc_amd
    res = {}
    for binding in ("compiled", "ctypes"):
        monkeypatch.setenv("DPC_BINDING", "" if binding == "compiled" else "ctypes")
        dpc_amd._ext.reset()
        if binding == "compiled" and dpc_amd._ext.module() is None:
            pytest.skip("compiled binding not built (python __graft_entry__.py)")
        assert (dpc_amd._ext.module() is None) == (binding == "ctypes")
        res[binding] = kw["run"]()
    monkeypatch.delenv("DPC_BINDING")
    dpc_amd._ext.reset()
    return res["compiled"], res["ctypes"]


def test_emu_compiled_binding_equals_ctypes_binding(emu, monkeypatch):
    """Same inputs through csrc/dpc_torch.cpp and through ops.ProjectFused: every output and every gradient bit for bit
    (both only marshal pointers into the same library), for the plain call, for translation + focal length + L2 epilogue +
    fused dropout, and for in-kernel replication + candidate loss."""
    import dpc_amd
    B, N, D, K = 4, 200, 32, 5
    inp = synth.make_inputs(B, N, 9)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu")
    t = lambda a: torch.tensor(a, requires_grad=True)
    w = torch.tensor(np.random.default_rng(1).standard_normal((B, D, D, 1)).astype(np.float32))
    wd = 0.1 * torch.tensor(np.random.default_rng(2).standard_normal((B, D, D, 1)).astype(np.float32))

    def plain():
        pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
        g = torch.autograd.grad([out["proj"], out["proj_depth"]], [pc, pose, scale], [w, wd])
        return [out["proj"], out["proj_depth"], out["tr_pc"], *g]

    def rich():
        pc, pose, scale = t(0.5 * inp["pc"]), t(inp["pose"]), t(inp["scale"])
        trans = t(0.02 * np.ones((B, 3), np.float32))
        focal = t(np.full((B, 1), 1.9, np.float32))
        gt = torch.tensor(synth.disk_gt(B, D))
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, trans, None, kern, scaling_factor=scale, focal_length=focal,
                                              l2_target=(gt, 0.25), point_dropout=(150, 77))
        g = torch.autograd.grad(out["proj"], [pc, pose, trans, scale, focal], out["proj_l2_grad"])
        return [out["proj"], out["proj_l2_grad"], *g]

    def replicated():
        clouds, pose, scale = t(inp["pc"][:2]), t(inp["pose"]), t(inp["scale"])
        masks = torch.tensor(synth.disk_gt(2, 48))
        out = dpc_amd.pointcloud_project_fast(cfg, clouds, pose, None, None, kern, scaling_factor=scale, views_per_cloud=2,
                                              silhouette_target=(masks, 2, None))
        g = torch.autograd.grad(out["proj_loss"], [clouds, pose, scale])
        return [out["proj"], out["proj_loss"], out["winning_pose_candidates"], out["proj_inst_err"], *g]

    for run in (plain, rich, replicated):
        a, b = _project_both_ways(monkeypatch, run=run)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), run.__name__


def test_emu_compiled_binding_keeps_the_error_behaviour(emu, monkeypatch):
    """the exceptions a caller can see are those of the ctypes binding: ValueError for shapes / devices, TypeError for
    dtypes, DpcError for a status code of the library"""
    import dpc_amd
    if dpc_amd._ext.module() is None:
        pytest.skip("compiled binding not built")
    cfg = dpc_amd.default_config(vox_size=32, pc_gauss_kernel_size=5)
    kern = dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu")
    inp = synth.make_inputs(2, 50, 3)
    pc, pose = torch.tensor(inp["pc"]), torch.tensor(inp["pose"])
    with pytest.raises(ValueError):
        dpc_amd.pointcloud_project_fast(cfg, pc[..., :2], pose, None, None, kern)
    with pytest.raises(ValueError, match="quaternion"):
        dpc_amd.pointcloud_project_fast(cfg, pc, pose[:, :3], None, None, kern)
    with pytest.raises(TypeError):
        dpc_amd.pointcloud_project_fast(cfg, pc.double(), pose, None, None, kern)
    with pytest.raises(ValueError):
        dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=torch.ones(3, 1))
    cfg48 = dpc_amd.default_config(vox_size=50, pc_gauss_kernel_size=5)     # generic path (50 does not fill whole lanes): no fused dropout
    with pytest.raises(ValueError, match="fused point dropout"):
        dpc_amd.pointcloud_project_fast(cfg48, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg48, 0.9, device="cpu"),
                                        point_dropout=(10, 1))


# ---------------------------------------------------------------------------------------------------------------
# grids narrower than the fused kernels' power-of-two lane geometry (default_config.yaml:77 accepts any vox_size)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(2, 600, 48, 48, 5, 0.9, False, False), (2, 300, 24, 24, 5, 0.8, True, False),
                                  (1, 500, 40, 40, 7, 1.2, False, True), (1, 400, 96, 48, 5, 0.9, False, False)])
def test_emu_padded_grids_take_the_fused_path(emu, case):
    """vox_size 48 / 24 / 40 (on the 64- / 32- / 64-wide geometry, one strip per plane) and 96 (on the 128-wide one, two
    strips): the fused front / back end with the lanes and rows beyond the grid masked, forward and all gradients against
    the float64 NumPy oracle (the helper also asserts that the fused path -- not the generic one -- took the shape)."""
    import parity_cases
    parity_cases.fused_path_against_numpy_oracle("cpu", *case)


def test_fused_width_rule():
    """which vox_size values the fused path takes: every multiple of 4 in (16, 256] -- the rows are padded inside the kernels to the
    next power-of-two geometry (round 4: only widths that fill whole lanes of it; round 6: the padding is masked per 16-byte vector
    everywhere, so any multiple of 4 -- 100, 136, ... no longer fall to the generic path, 3 x slower)"""
    import ctypes
    import dpc_amd
    lib = dpc_amd.get_library()
    P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0)
    fused = lambda D: bool(lib.dpc_saved_layout(ctypes.byref(dpc_amd._capi.DpcShape(2, 100, 32, D, 5, 5, 5)), ctypes.byref(P)) & 2)
    assert all(fused(D) for D in (20, 24, 28, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 144, 160, 192, 240, 256, 36, 44, 52, 68, 100, 132, 136, 200, 252))
    assert not any(fused(D) for D in (16, 12, 18, 30, 33, 50, 66, 98, 130, 254, 257, 260))


def test_emu_view_walking_order_does_not_change_results(tmp_path):
    """The launcher walks the views last-to-first in some kernels when a grid exceeds the Infinity Cache
    (host_launch.inc: view_order): every order (DPC_VIEW_ORDER = 0 ... 15, read once per process) gives the same bits."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, hashlib
os.environ["DPC_TEST_HOOKS"] = "1"
import numpy as np, torch
sys.path.insert(0, %r)
import dpc_amd
emu = dpc_amd._capi.DpcLibrary(os.path.join(%r, "tests", "hipemu", "libdpc_emu.so"), host_memory=True)
dpc_amd._capi.set_library(emu)
rng = np.random.default_rng(4)
B, N, D, K = 5, 300, 32, 5
cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
pc = torch.tensor((rng.normal(size=(B, N, 3)) * 0.15).astype(np.float32), requires_grad=True)
pose = torch.tensor(rng.normal(size=(B, 4)).astype(np.float32), requires_grad=True)
scale = torch.tensor(rng.uniform(0.5, 1.0, (B, 1)).astype(np.float32), requires_grad=True)
out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu"), scaling_factor=scale)
w = torch.tensor(rng.standard_normal((B, D, D, 1)).astype(np.float32))
g = torch.autograd.grad((out["proj"] * w).sum(), [pc, scale])
h = hashlib.sha256()
for t in (out["proj"], out["proj_depth"], g[0]):
    h.update(t.detach().numpy().tobytes())
print("HASH", h.hexdigest(), float(g[1].sum()))
''' % (ROOT, ROOT)
    res = {}
    for order in ("0", "15"):
        env = dict(os.environ, DPC_VIEW_ORDER=order)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("HASH")][0].split()
        res[order] = (line[1], float(line[2]))
    assert res["0"][0] == res["15"][0], res
    assert abs(res["0"][1] - res["15"][1]) <= 1e-5 * max(1.0, abs(res["0"][1]))      # (dscale: per-work-group partials, same order)
