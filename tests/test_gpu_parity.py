"""`-m gpu` tier: the real libdpc_hip.so on an MI355X, through the product
Python API and the C ABI, against (a) the goldens generated from the reference
source, (b) the CPU oracles on the same seeded inputs, (c) size-independent
properties at BASELINE.json's full sizes.

Tolerances (fp32 path; north-star bar: silhouette max-abs <= 1e-4):
  proj 2e-5 vs fp64 truth, depth 2e-4, gradients 2e-4 of the tensor's largest
  reference gradient.  Float atomics make the scatter order-dependent (~1e-7
  relative), so nothing here demands bitwise equality."""
import numpy as np
import pytest
import torch

import dpc_amd
from helpers import (ALL_CASES, DRC_VARIANT_CASES, close_elementwise, close_elementwise_piecewise, load, maxabs, onp, rcpu,
                     relerr, synth)
from run_case import run_product
import parity_cases

pytestmark = pytest.mark.gpu

GRAD_KEYS = ("dpc", "dpose", "dtrans", "dscale", "dfocal")
TOL_PROJ = 2e-5
TOL_DEPTH = 2e-4
TOL_GRAD = 2e-4


@pytest.fixture(scope="module", autouse=True)
def _real_library():
    dpc_amd._capi.set_library(None)
    lib = dpc_amd.get_library()
    assert lib.path.endswith("libdpc_hip.so") and not lib.host_memory
    yield lib


@pytest.mark.parametrize("name", ALL_CASES + DRC_VARIANT_CASES)
def test_goldens(name):
    g = load(name)
    lazy = "voxels_f64" in g
    res, gr = run_product(name, g, "cuda", grads=True, touch_lazy=lazy)
    assert maxabs(res["tr_pc"], g["tr_pc_f64"]) < 2e-6
    assert maxabs(res["proj"], g["proj_f64"]) < TOL_PROJ
    assert maxabs(res["proj"], g["proj_f32"]) < 1e-4          # the north-star bar vs the fp32 reference run
    if "proj_depth_f64" in g:
        assert maxabs(res["proj_depth"], g["proj_depth_f64"]) < TOL_DEPTH
    if "voxels_f64" in g:
        assert maxabs(res["voxels"], g["voxels_f64"]) < TOL_PROJ
    if "drc_probs_f64" in g:
        assert maxabs(res["drc_probs"], g["drc_probs_f64"]) < TOL_PROJ
    for k in GRAD_KEYS:
        if k + "_f64" in g:
            assert relerr(gr[k], g[k + "_f64"]) < TOL_GRAD, k


def test_nan_points_dropped():
    g = load("tiny_nan")
    res, _ = run_product("tiny_nan", g, "cuda", grads=False)
    assert np.isfinite(res["proj"]).all()
    assert maxabs(res["proj"], g["proj_f64"]) < TOL_PROJ


def _cfg2_inputs(B, dev, grad=True):
    c = synth.config_inputs(2, B=B)
    cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    t = lambda a: torch.tensor(a, device=dev, requires_grad=grad)
    return c, cfg, t(c["pc"]), t(c["pose"]), t(c["scale"]), dpc_amd.smoothing_kernel(cfg, c["sigma"], device=dev)


def test_cfg2_shapes_against_cpu_oracle():
    """BASELINE configs[1] shapes (8000 pts, 128^3, K=11, sigma=1.6) at B=2:
    HIP vs the op-for-op torch-CPU restatement on identical inputs."""
    c, cfg, pc, pose, scale, kern = _cfg2_inputs(2, "cuda")
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    gt = torch.tensor(synth.disk_gt(2, c["D"]), device="cuda")
    dproj = ((out["proj"] - gt) / 2).detach()
    gpc, gpose, gscale = torch.autograd.grad(out["proj"], [pc, pose, scale], dproj)

    rc = rcpu.Cfg(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    cpc = torch.tensor(c["pc"], dtype=torch.float64, requires_grad=True)
    cpose = torch.tensor(c["pose"], dtype=torch.float64, requires_grad=True)
    cscale = torch.tensor(c["scale"], dtype=torch.float64, requires_grad=True)
    ref = rcpu.pointcloud_project_fast(rc, cpc, cpose, None, None, rcpu.smoothing_kernel(rc, c["sigma"], torch.float64),
                                       scaling_factor=cscale)
    rg = torch.autograd.grad(ref["proj"], [cpc, cpose, cscale], dproj.cpu().double())
    assert maxabs(out["proj"].detach().cpu().numpy(), ref["proj"].detach().numpy()) < TOL_PROJ
    assert maxabs(out["proj_depth"].detach().cpu().numpy(), ref["proj_depth"].detach().numpy()) < TOL_DEPTH
    assert relerr(gpc.cpu().numpy(), rg[0].numpy()) < TOL_GRAD
    assert relerr(gpose.cpu().numpy(), rg[1].numpy()) < TOL_GRAD
    assert relerr(gscale.cpu().numpy(), rg[2].numpy()) < TOL_GRAD


def test_full_batch_properties_cfg2():
    """bs=32 at full size: properties that need no CPU reference."""
    c, cfg, pc, pose, scale, kern = _cfg2_inputs(32, "cuda")
    D = c["D"]
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    proj = out["proj"]
    assert proj.shape == (32, D, D, 1) and torch.isfinite(proj).all()
    # empty rays: 1-(1-eps)^Dz (A.2), 1.279e-3 at Dz=128
    assert abs(float(proj.min()) - 1.279e-3) < 3e-6
    assert float(proj.max()) <= 1.0 + 1e-5
    # mass conservation of the trilinear scatter: sum(G0) = number of valid points
    tr = out["tr_pc"].detach()
    valid = ((tr >= -0.5) & (tr <= 0.5)).all(-1).sum(1).double()
    raw, _ = dpc_amd.pointcloud2voxels3d_fast(cfg, tr, None)
    assert torch.allclose(raw.double().sum((1, 2, 3)), valid, rtol=1e-5)
    # instances are independent: the first two views equal a B=2 run
    out2 = dpc_amd.pointcloud_project_fast(cfg, pc[:2].detach(), pose[:2].detach(), None, None, kern,
                                           scaling_factor=scale[:2].detach())
    assert float((out2["proj"] - proj[:2]).abs().max()) < 1e-5
    # backward runs at full size; outliers (1 % of points, outside the cube) get exactly zero gradient
    gt = torch.tensor(synth.disk_gt(32, D), device="cuda")
    gpc, gpose, gscale = torch.autograd.grad(proj, [pc, pose, scale], ((proj - gt) / 32).detach())
    assert torch.isfinite(gpc).all() and torch.isfinite(gpose).all() and torch.isfinite(gscale).all()
    invalid = ~((tr >= -0.5) & (tr <= 0.5)).all(-1)
    assert invalid.any() and float(gpc[invalid].abs().max()) == 0.0
    assert float(gpc.abs().max()) > 0


def test_point_permutation_invariance_and_rerun_stability():
    c, cfg, pc, pose, scale, kern = _cfg2_inputs(2, "cuda", grad=False)
    run = lambda p: dpc_amd.pointcloud_project_fast(cfg, p, pose, None, None, kern, scaling_factor=scale)["proj"]
    a = run(pc)
    b = run(pc)
    perm = torch.randperm(pc.shape[1], device="cuda")
    cperm = run(pc[:, perm])
    assert float((a - b).abs().max()) < 1e-5          # atomics ordering noise only
    assert float((a - cperm).abs().max()) < 1e-5


def test_stage_level_api_matches_cpu_oracle():
    parity_cases.stage_level_api_matches_cpu_oracle("cuda")


@pytest.mark.parametrize("D,K", parity_cases.ODD_CASES)
def test_odd_sizes_and_generic_tap_counts(D, K):
    parity_cases.odd_sizes_and_generic_tap_counts("cuda", D, K)


@pytest.mark.parametrize("case", parity_cases.FUSED_CASES_GPU)
def test_fused_path_against_numpy_oracle(case):
    parity_cases.fused_path_against_numpy_oracle("cuda", *case)


@pytest.mark.parametrize("name", ["tiny_rgb", "tiny_rgb_div"])
def test_rgb_channels(name):
    parity_cases.rgb_case_matches_goldens("cuda", name)


@pytest.mark.parametrize("name", parity_cases.LOSS_CASES)
def test_silhouette_loss(name):
    parity_cases.silhouette_loss_matches_reference("cuda", name)


@pytest.mark.parametrize("name,tag", parity_cases.NN_CASES)
def test_nn_distance(name, tag):
    parity_cases.nn_distance_matches_reference("cuda", name, tag)


def test_nn_distance_gradient_and_brute_force_at_scale():
    parity_cases.nn_distance_gradient("cuda")
    from dpc_amd.util.point_cloud_distance import point_cloud_distance
    gen = torch.Generator().manual_seed(9)
    vs = torch.rand(8000, 3, generator=gen, dtype=torch.float64).cuda()
    vt = torch.rand(30000, 3, generator=gen, dtype=torch.float64).cuda()
    proj, dist, idx = point_cloud_distance(vs, vt)
    ref = torch.cdist(vs, vt).min(dim=1)                     # evaluation-scale cross-check (different rounding)
    assert float((dist - ref.values).abs().max()) < 1e-9
    assert float((idx.to(torch.int64) != ref.indices).float().mean()) < 1e-3
    assert torch.equal(proj, vt[idx.to(torch.int64)])


@pytest.mark.parametrize("name", parity_cases.SLOW_VOX_CASES)
def test_gauss_voxeliser(name):
    parity_cases.gauss_voxeliser_matches_reference("cuda", name)


def test_slow_projector():
    parity_cases.slow_projector_matches_reference("cuda")


@pytest.mark.parametrize("mode", [None, "sum", "analytical"])
def test_gauss_voxeliser_multitile(mode):
    parity_cases.gauss_voxeliser_multitile_against_numpy_oracle("cuda", B=2, N=333, G=70, sigma=0.06, mode=mode)


def test_hip_graph_replay_matches_eager():
    """The library only enqueues on the stream it is given: a whole fwd+bwd step captures into a
    hipGraph (torch.cuda.CUDAGraph) and a replay on refreshed static inputs equals the eager run."""
    c = dpc_amd.synthetic.config_inputs(1, B=4)
    cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device="cuda")
    pc = torch.tensor(c["pc"], device="cuda", requires_grad=True)
    pose = torch.tensor(c["pose"], device="cuda", requires_grad=True)
    scale = torch.tensor(c["scale"], device="cuda", requires_grad=True)
    w = torch.rand(4, c["D"], c["D"], 1, device="cuda")

    def step():
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
        return (out["proj"],) + torch.autograd.grad(out["proj"], [pc, pose, scale], w)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static = step()
    c2 = dpc_amd.synthetic.config_inputs(1, B=4, seed_offset=7)            # new data into the static inputs
    with torch.no_grad():
        pc.copy_(torch.tensor(c2["pc"]))
        pose.copy_(torch.tensor(c2["pose"]))
    graph.replay()
    torch.cuda.synchronize()
    replayed = [t.clone() for t in static]
    eager = step()
    assert float((replayed[0] - eager[0]).abs().max()) < 2e-6              # float atomics: order-dependent rounding only
    for a, b in zip(replayed[1:], eager[1:]):
        assert relerr(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


def test_integration_md_ctypes_stub_runs_standalone():
    """The minimal ctypes binding printed in INTEGRATION.md (no dpc_amd import) drives the C ABI
    and reproduces the product path."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes, torch.*?)```", text, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)
    try:
        exec(compile(block, "INTEGRATION.md", "exec"), ns)
        c = dpc_amd.synthetic.config_inputs(1, B=2)
        cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
        kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device="cuda")
        pc, pose, scale = (torch.tensor(c[k], device="cuda") for k in ("pc", "pose", "scale"))
        taps = [k.reshape(-1).contiguous() for k in kern]
        proj, depth, saved = ns["project_forward"](cfg, pc, pose, scale, taps)
        torch.cuda.synchronize()
    finally:
        os.chdir(cwd)
    ref = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    assert float((proj - ref["proj"]).abs().max()) < 2e-6
    assert float((depth - ref["proj_depth"]).abs().max()) < 2e-5


def test_fused_edge_planes():
    parity_cases.fused_edge_planes_against_numpy_oracle("cuda")


def test_d256_fused_path_against_numpy_oracle():
    """256^3 grid (BASELINE configs[4] resolution) at B=1: k_splat_xy / k_gather_yx with 32-row
    strips and 16-byte lanes, against the float64 NumPy oracle."""
    parity_cases.fused_path_against_numpy_oracle("cuda", 1, 3000, 256, 256, 11, 2.0, False, False)


def test_cfg5_stress_shape_runs():
    """BASELINE configs[4]: 16000 pts -> 256^3, sigma 2.0 (reduced to B=2 here;
    bench.py --config 5 runs B=8)."""
    c = synth.config_inputs(5, B=2)
    cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None,
                                          dpc_amd.smoothing_kernel(cfg, c["sigma"], device="cuda"), scaling_factor=scale)
    proj = out["proj"]
    assert abs(float(proj.min()) - 2.557e-3) < 6e-6
    g = torch.autograd.grad(proj, [pc, pose, scale], torch.ones_like(proj) / proj.numel())
    assert all(torch.isfinite(x).all() for x in g)


# ---------------------------------------------------------------------------
# round 2
# ---------------------------------------------------------------------------
from helpers import close_elementwise  # noqa: E402


@pytest.mark.parametrize("name", ["tiny", "tiny_focal", "tiny_matrix", "k21", "cfg1", "mid"])
def test_point_gradients_elementwise(name):
    """dpc entry by entry against the fp64 goldens: |err| <= 2e-5 max|ref| + 1e-3 |ref| (a small entry that is
    100 % wrong fails; the max-abs / max-magnitude ratio of test_goldens would let it through)."""
    g = load(name)
    _, gr = run_product(name, g, "cuda", grads=True)
    ok, worst = close_elementwise(gr["dpc"], g["dpc_f64"])
    assert ok, worst
    # the per-view gradients (pose, translation, scale, focal length) component by component as well
    for k in ("dpose", "dtrans", "dscale", "dfocal"):
        if k + "_f64" in g:
            ok, worst = close_elementwise(gr[k], g[k + "_f64"], rtol=1e-3, atol_frac=1e-4)
            assert ok, (k, worst)


@pytest.mark.parametrize("D,K", parity_cases.ASYM_CASES + [(64, 21), (128, 11)])
def test_asymmetric_filters(D, K):
    parity_cases.asymmetric_filters_against_cpu_oracle("cuda", D, K)


def test_student_loss():
    parity_cases.student_loss_equals_the_quaternion_composite("cuda")
    parity_cases.student_loss_equals_the_quaternion_composite("cuda", n=80, C=4, seed=1)
    parity_cases.student_loss_equals_the_quaternion_composite("cuda", n=300, C=2, seed=9)


def test_fused_l2_epilogue():
    parity_cases.fused_l2_epilogue_equals_the_autograd_loss("cuda")
    parity_cases.fused_l2_epilogue_equals_the_autograd_loss("cuda", B=32, N=8000, D=128, K=11)    # cfg2 (what bench.py times)
    parity_cases.fused_l2_epilogue_equals_the_autograd_loss("cuda", B=8, N=2000, D=64, K=21)
    parity_cases.fused_l2_epilogue_equals_the_autograd_loss("cuda", B=2, N=500, D=20, K=5)        # generic path


def test_fused_dropout():
    parity_cases.fused_dropout_equals_explicit_subset("cuda")
    parity_cases.fused_dropout_equals_explicit_subset("cuda", B=5, N=8000, D=64, K=21, keep=560, seed=77)
    parity_cases.fused_dropout_equals_explicit_subset("cuda", B=2, N=9000, D=64, K=11, keep=4500, seed=3)   # two-kernel sort


@pytest.mark.parametrize("D,Dz", [(32, 33), (64, 65), (33, 33)])
def test_knife_edges_without_nudging(D, Dz):
    parity_cases.knife_edge_inputs_match_reference_conventions("cuda", D, Dz)


def test_degenerate_clouds():
    parity_cases.degenerate_clouds_against_numpy_oracle("cuda", heavy=True)


def test_deep_grid_takes_the_generic_path():
    parity_cases.deep_grid_takes_the_generic_path("cuda")


def _against_reference_cpu(c, B, dev="cuda", chunk=None):
    """HIP fwd+bwd on B views of a synthetic config against oracle/reference_cpu.py (fp64) on the SAME views.
    The function is only piecewise smooth (cell faces, the clip at G0 = 1): with ~10^5 points a handful sit
    within fp32 rounding of such an edge and the fp32 path may legitimately take the other piece than fp64, so
    those points are moved off the edges first (the conventions AT the edges are pinned, un-nudged, by
    test_knife_edges_without_nudging)."""
    c = dict(c)
    nudged = parity_cases._nudge_off_cell_faces({"pc": c["pc"], "pose": c["pose"]}, None, None, c["D"], c["D"])
    c["pc"] = nudged["pc"]
    # how many inputs the nudge touched: a coordinate lands within 3e-5 of a lattice plane with probability 6e-5 per
    # axis, so ~2e-4 of the points are expected (measured: 1.8e-4 at cfg2, 2.5e-4 at cfg5), plus the corners of cells
    # whose sum sits within 3e-5 of 1 -- common where 8000 points share a 64^3 grid (6.7e-4 at the training shape);
    # far more than that would mean the comparison no longer sees the workload
    npts = c["pc"].shape[0] * c["pc"].shape[1]
    print("moved off cell faces / G0 = 1 knife edges: %d of %d points (%.2e)" % (nudged["moved_points"], npts,
                                                                              nudged["moved_points"] / npts))
    assert nudged["moved_points"] <= 1e-3 * npts + 2, nudged["moved_points"]
    cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
    kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device=dev)
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    gt = torch.tensor(synth.disk_gt(B, c["D"]), device=dev)
    dproj = ((out["proj"] - gt) / B).detach()
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], dproj)
    rc = rcpu.Cfg(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    ckern = rcpu.smoothing_kernel(rc, c["sigma"], torch.float64)
    chunk = chunk or B
    worst = {}
    ref_grads = [[], [], []]
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        d = lambda a: torch.tensor(a[lo:hi], dtype=torch.float64, requires_grad=True)
        cpc, cpose, cscale = d(c["pc"]), d(c["pose"]), d(c["scale"])
        ref = rcpu.pointcloud_project_fast(rc, cpc, cpose, None, None, ckern, scaling_factor=cscale)
        rg = torch.autograd.grad(ref["proj"], [cpc, cpose, cscale], dproj[lo:hi].cpu().double())
        for acc, x in zip(ref_grads, rg):
            acc.append(x.numpy())
        e = {"proj": maxabs(out["proj"][lo:hi].detach().cpu().numpy(), ref["proj"].detach().numpy()),
             "depth": maxabs(out["proj_depth"][lo:hi].detach().cpu().numpy(), ref["proj_depth"].detach().numpy()),
             "dpc": relerr(g[0][lo:hi].cpu().numpy(), rg[0].numpy()),
             "dpose": relerr(g[1][lo:hi].cpu().numpy(), rg[1].numpy()),
             "dscale": relerr(g[2][lo:hi].cpu().numpy(), rg[2].numpy())}
        for k, v in e.items():
            worst[k] = max(worst.get(k, 0.0), v)
    assert worst["proj"] < TOL_PROJ and worst["depth"] < TOL_DEPTH, worst
    assert worst["dpc"] < TOL_GRAD and worst["dpose"] < TOL_GRAD and worst["dscale"] < TOL_GRAD, worst
    # ELEMENTWISE over the whole batch: every point's gradient, every pose / scale component on its own
    # (|err| <= atol_frac max|ref| + rtol |ref|): a small entry that is wrong cannot hide behind the largest one.
    # Point gradients: at most 1e-4 of the entries beyond the bound and none beyond 5 bounds (see the helper: fp32 and
    # fp64 pick different pieces of the eps-clip in a few voxels); the per-view sums strictly.
    ok, ratio, frac = close_elementwise_piecewise(g[0].cpu().numpy(), np.concatenate(ref_grads[0]))
    worst["dpc_elementwise"], worst["dpc_beyond_bound"] = ratio, frac
    assert ok, ("dpc", ratio, frac)
    for name, got, ref_parts in (("dpose", g[1], ref_grads[1]), ("dscale", g[2], ref_grads[2])):
        ok, ratio = close_elementwise(got.cpu().numpy(), np.concatenate(ref_parts), rtol=1e-3, atol_frac=1e-4)
        worst[name + "_elementwise"] = ratio
        assert ok, (name, ratio)
    print("worst errors:", {k: float("%.3g" % v) for k, v in worst.items()})
    return worst


def test_cfg2_full_batch_against_cpu_oracle():
    """BASELINE configs[1] at its FULL batch (32 views x 8000 pts, 128^3, K=11): every view against the
    torch-CPU restatement of the reference graph (fp64), forward and all gradients."""
    _against_reference_cpu(synth.config_inputs(2), 32, chunk=8)


def test_cfg5_full_batch_against_cpu_oracle():
    """BASELINE configs[4] at its FULL size (8 views x 16000 pts, 256^3, K=11, sigma 2.0)."""
    _against_reference_cpu(synth.config_inputs(5), 8, chunk=1)


TRAIN_SHAPE = dict(B=320, D=64, K=21, sigma=3.0)    # configs[2]: 16 models x 5 views x 4 pose candidates


@pytest.mark.parametrize("N", [560, 4000, 8000])      # the dropout schedule's point counts (keep 0.07 -> 1.0)
def test_training_shape_projector(N):
    """The projector at the shape the reference trains with (B=320, 64^3, K=21): properties at the full batch,
    the CPU oracle on a slice of 8 views."""
    synth.CONFIGS[3] = dict(TRAIN_SHAPE, N=N)
    c = synth.config_inputs(3)
    cfg = dpc_amd.default_config(vox_size=64, pc_gauss_kernel_size=21)
    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
    kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device="cuda")
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    proj = out["proj"]
    assert proj.shape == (320, 64, 64, 1) and torch.isfinite(proj).all()
    assert float(proj.min()) >= 6.39e-4 - 3e-6 and float(proj.max()) <= 1.0 + 1e-5      # 1-(1-eps)^64 = 6.40e-4
    gt = torch.tensor(synth.disk_gt(320, 64), device="cuda")
    g = torch.autograd.grad(proj, [pc, pose, scale], ((proj - gt) / 320).detach())
    assert all(torch.isfinite(x).all() for x in g)
    tr = out["tr_pc"].detach()
    invalid = ~((tr >= -0.5) & (tr <= 0.5)).all(-1)
    assert float(g[0][invalid].abs().max()) == 0.0
    # instances are independent and the splat is order-independent: views 8..15 alone reproduce the batch bit for bit
    sub = dpc_amd.pointcloud_project_fast(cfg, pc[8:16].detach(), pose[8:16].detach(), None, None, kern,
                                          scaling_factor=scale[8:16].detach())
    assert float((sub["proj"] - proj[8:16]).abs().max()) == 0.0
    # oracle on the first 8 views (their upstream gradient is that of the 8-view problem: recompute it there)
    c8 = {k: (v[:8] if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    _against_reference_cpu(c8, 8, chunk=4)


def test_training_step_runs_and_learns():
    """BASELINE configs[2]: the full chair_unsupervised step (stock PyTorch nets -> HIP projector with fused
    point dropout -> HIP silhouette-loss epilogue with the min over 4 pose candidates -> Adam) on 1 GPU, at a
    reduced model batch; the loss is finite, every parameter gets a gradient, and a few steps lower the loss."""
    import os
    import sys
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "chair_unsupervised")
    sys.path.insert(0, ex)
    import train_step as ts
    from nets import Im2PointCloud
    dev = torch.device("cuda")
    cfg = ts.make_cfg(batch_size=4, pc_point_dropout=0.5, pc_point_dropout_scheduled=False)
    torch.manual_seed(0)
    net = Im2PointCloud(cfg, 128).to(dev)
    projector = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    inputs = ts.synthetic_batch(cfg, dev, 128, seed=0)
    losses = [float(ts.train_step(net, projector, inputs, opt)) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert projector._fused_dropout_ok(torch.empty(80, 8000, 3, device=dev), None)      # the fused draw was used
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_graph_replay_follows_the_dropout_and_sigma_schedules():
    """The whole training step recorded ONCE into a HIP graph (ModelPointCloud.enable_graph_replay): on every replay
    the fused dropout draws a new subset from the {keep, seed} pair in device memory, `keep` and the blur taps follow
    their schedules through set_global_step, and the projection inside the graph equals, bit for bit, an eager
    call on the same points with that (keep, seed) by value."""
    import os
    import sys
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "chair_unsupervised")
    sys.path.insert(0, ex)
    import train_step as ts
    from nets import Im2PointCloud
    dev = torch.device("cuda")
    cfg = ts.make_cfg(batch_size=2, pc_point_dropout=0.07, pc_point_dropout_scheduled=True, max_number_of_steps=100)
    torch.manual_seed(0)
    net = Im2PointCloud(cfg, 128).to(dev)
    projector = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=dev)
    projector.enable_graph_replay()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True)
    inputs = ts.synthetic_batch(cfg, dev, 128, seed=0)
    static = {}

    def run():
        outputs = net(inputs["images"])
        outputs = projector.replicate_outputs(outputs)
        outputs = projector.compute_projection(inputs, outputs, is_training=True)
        loss = projector.add_proj_loss(inputs, outputs, cfg.proj_weight)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        static.update(outputs=outputs, loss=loss.detach())

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    seeds, losses = [], []
    for step in (0, 1, 2, 40, 80, 100):
        projector.set_global_step(step)
        graph.replay()
        keep, seed = (int(v) for v in projector._dropout_state.tolist())
        want_keep = int(cfg.pc_num_points * dpc_amd.model_pc.get_dropout_prob(cfg, step))
        assert keep == want_keep, (step, keep, want_keep)
        seeds.append(seed)
        losses.append(float(static["loss"]))
        o = static["outputs"]
        kern = dpc_amd.smoothing_kernel(cfg, dpc_amd.model_pc.get_smooth_sigma(cfg, step), device=dev)
        for a, b in zip(kern, projector.gauss_kernel()):
            assert torch.equal(a, b)                                      # the taps moved in place
        # (the recorded step replicates inside the kernels; `all_points` of the captured dict would be built once, from
        # the first replay's clouds, so the copies are made here from the clouds of THIS replay)
        reps = cfg.step_size * cfg.pose_predict_num_candidates
        all_points = torch.repeat_interleave(o["points_1"].detach(), reps, dim=0)
        # (a step recorded ONCE runs the full 21-tap filter at every sigma: enable_graph_replay's default)
        cfg_full = type(cfg)(**dict(cfg, pc_trim_gauss_taps=False))
        eager = dpc_amd.pointcloud_project_fast(cfg_full, all_points, o["poses"].detach(), None, None, kern,
                                                scaling_factor=o["all_scaling_factors"].detach(),
                                                point_dropout=(keep, seed))
        # bit for bit while the integer splat applies; a plane holding >= 4096 of the (untrained, clustered) points
        # takes the float-atomic splat, whose sum order varies from launch to launch
        err = float((eager["proj"] - o["projs"].detach()).abs().max())
        assert err == 0.0 if keep < 4096 else err < 2e-6, (step, keep, err)
    assert len(set(seeds)) == len(seeds)                                  # a fresh draw per replay
    assert all(np.isfinite(losses))


def test_bench_line_contract_on_the_gpu():
    """`python bench.py` as the driver runs it (here at a toy step count): one JSON line with the contract's keys,
    the roofline and timing objects, the step replayed as a HIP graph; `--no-graph` gives the eager figure, and the
    gradients a replay leaves equal the eager ones (the recorded step is the measured step)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("DPC_POISON_BUFFERS", "DPC_TEST_HOOKS", "DPC_BENCH_DRY_RUN", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    lines = {}
    for extra in ([], ["--no-graph"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--repeats", "2",
                            "--no-cpu-baseline"] + extra, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(out) == 1, r.stdout
        lines[bool(extra)] = json.loads(out[0])
    j = lines[False]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "timing"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["unit"] == "views/s" and j["dtype"] == "f32"
    assert j["config"]["hip_graph"] is True and lines[True]["config"]["hip_graph"] is False
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 32.0) < 1e-6 * 32         # value = views / time
    r = j["roofline"]
    assert r["bound"] in ("hbm", "valu-issue", "latency") and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    # the issue side: SQ counters quoted only for the build they were taken on (else null + a note), and `bound` explained
    assert "issue" in r and "bound_basis" in r and (r["issue"] is None or r["issue"]["valu_per_wave"] > 0)
    if r["issue"] is None:
        assert r["bound"] == "hbm" and "unverified" in r["bound_basis"]["note"] and r.get("issue_note")
    # physical fractions: bytes this implementation must move over HIP-event time, against the spec peak
    assert 0 < r["frac"] <= 1.0 and 0 < r["step_frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert not any("frac" in k and isinstance(v, float) and v > 1.0 for k, v in r.items())
    assert r["kernel_bytes"]["read"] + r["kernel_bytes"]["written"] > 0
    c = r["ceilings"]
    assert 2000 < c["read"] < 8000 and 2000 < c["write"] < 8000 and 2000 < c["copy"] < 8000
    # PMC traffic is quoted only when profiles/traffic.json was taken on this very build; otherwise null + a note
    assert (r["traffic"] and r["traffic_source"]) or (r["traffic"] is None and r["traffic_note"])
    assert j["config"]["taps_run"] == 11 and j["config"]["sigma"] == 1.6
    assert 0.5 < lines[True]["value"] / j["value"] < 2.0


# ---------------------------------------------------------------------------
# round 3
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("keep", [560, 4000])
def test_fused_dropout_statistics_on_device(keep):
    """The draw the KERNELS make (read back from point_index after dpc_project_forward) over 10 240 (seed, instance)
    keys at N = 8000: equal to oracle/dropout_ref.py key by key on a sample, every instance keeps exactly `keep`
    points, and inclusion / pair-inclusion frequencies are those of sampling without replacement (chi-squares
    within 5 sigma, no z-score beyond 6) -- the reference draws with np.random.choice (point_cloud.py:293-319)."""
    from oracle import dropout_ref
    B, N, D, K = 320, 8000, 32, 5
    rng = np.random.default_rng(12)
    pc = rng.uniform(-0.25, 0.25, (B, N, 3)).astype(np.float32)          # |p| <= 0.44: inside the cube under any rotation
    pose = rng.standard_normal((B, 4)).astype(np.float32)
    masks = []
    for seed in range(1000, 1032):
        m = parity_cases.fused_dropout_kept_mask_from_library("cuda", pc, pose, D, K, keep, seed)
        if seed < 1002:
            assert np.array_equal(m[:6], dropout_ref.kept_mask(6, N, keep, seed))
        masks.append(m)
    masks = np.concatenate(masks)
    assert masks.shape == (10240, N)
    st = parity_cases.dropout_statistics(masks, keep, parity_cases.dropout_pairs(N, np.random.default_rng(7)))
    print("dropout statistics on device:", st)
    parity_cases.assert_dropout_statistics(st)


def test_points_bwd_view_finalize_under_stress():
    """k_points_bwd_sorted lets the LAST work-group of a view finish the view (quaternion Jacobian, dscale) from
    the other work-groups' atomics, ordered by returning atomics + a ticket instead of a release fence (DESIGN.md
    5.1).  10 000 backward launches -- 64 views x 32 work-groups each, a second stream hammering HBM to make the
    load uneven -- must all give the pose / scale gradients of a fixed-order fp64 sum: a hand-off that ever read
    a partial sum would be off by ~1/32 of the value, ten thousand times the rounding of the atomics' order."""
    B, N, D, K, sigma = 64, 8000, 32, 5, 0.9
    inp = synth.make_inputs(B, N, 31)
    inp = parity_cases._nudge_off_cell_faces(inp, None, None, D, D)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device="cuda")
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    w = torch.tensor(np.random.default_rng(3).standard_normal(tuple(out["proj"].shape)).astype(np.float32), device="cuda")
    f64 = lambda a: a.astype(np.float64)
    taps = onp.smoothing_taps(D, -1, K, sigma)
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, Dz=D, D=D)
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, fw,
                              dproj=f64(w.cpu().numpy()))
    ref_pose = torch.tensor(bw["dpose"], device="cuda", dtype=torch.float32)
    ref_scale = torch.tensor(bw["dscale"].reshape(B, 1), device="cuda", dtype=torch.float32)
    tol_pose = 2e-4 * float(ref_pose.abs().max())
    tol_scale = 2e-4 * float(ref_scale.abs().max())
    worst = torch.zeros(2, device="cuda")
    # background traffic on another stream: copies of a buffer larger than the Infinity Cache
    noise_src = torch.empty(96 << 20, dtype=torch.float32, device="cuda").normal_()
    noise_dst = torch.empty_like(noise_src)
    side = torch.cuda.Stream()
    launches = 10000
    for i in range(launches):
        if i % 8 == 0:
            with torch.cuda.stream(side):
                noise_dst.copy_(noise_src, non_blocking=True)
        g = torch.autograd.grad(out["proj"], [pose, scale], w, retain_graph=True)
        worst = torch.maximum(worst, torch.stack([(g[0] - ref_pose).abs().max(), (g[1] - ref_scale).abs().max()]))
    torch.cuda.synchronize()
    worst = worst.cpu().numpy()
    print("finalize stress: worst |dpose - ref| %.3e (tol %.3e), worst |dscale - ref| %.3e (tol %.3e) over %d launches"
          % (worst[0], tol_pose, worst[1], tol_scale, launches))
    assert worst[0] <= tol_pose and worst[1] <= tol_scale, worst


def _bench_multi_gpu(argv, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "DPC_BENCH_DRY_RUN", "DPC_POISON_BUFFERS")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")


@needs_two_gpus
def test_bench_two_gpus_projector_under_rccl():
    """`python bench.py --gpus 2`: bench.py starts its two ranks (torch.distributed.run, 127.0.0.1), backend nccl = RCCL,
    per-rank HIP-graph capture beside the RCCL watchdog, barriers + MAX all-reduce of the step time; weak and strong."""
    j = _bench_multi_gpu(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["global_batch"] == 64 and j["data"] == "synthetic"
    assert j["value"] > 0 and j["config"]["hip_graph"] is True
    s = _bench_multi_gpu(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--scaling", "strong"])
    assert s["scaling"] == "strong" and s["config"]["global_batch"] == 32


@needs_two_gpus
def test_bench_two_gpus_training_step_ddp_and_recorded():
    """BASELINE configs[3] on two GPUs: the eager DDP step, and the recorded step (--graph: forward, backward, the
    bucketed RCCL all-reduce of the gradients issued from hooks, Adam -- one HIP graph per rank)."""
    e = _bench_multi_gpu(["--gpus", "2", "--config", "3", "--steps", "5", "--warmup", "3", "--batch", "4"])
    assert e["n_gpus"] == 2 and e["config"]["training_step"] and "DDP" in e["config"]["parallelism"] and e["value"] > 0
    g = _bench_multi_gpu(["--gpus", "2", "--config", "3", "--steps", "5", "--warmup", "3", "--batch", "4", "--graph"])
    assert g["n_gpus"] == 2 and "GradBuckets" in g["config"]["parallelism"] and g["value"] > 0
    # the capture must have worked (a fall-back to eager launches is reported, and is a failure of this test)
    assert g["config"]["hip_graph"] is True, g["config"].get("hip_graph_note")


@needs_two_gpus
def test_grad_buckets_equal_ddp_on_two_gpus():
    """GradBuckets (the recordable reducer) and DistributedDataParallel give the same averaged gradients under RCCL."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "examples", "chair_unsupervised"))
import dpc_amd, train_step as ts
from nets import Im2PointCloud
rank, world, dev = dpc_amd.distributed.init("nccl")
cfg = ts.make_cfg(batch_size=2, pc_point_dropout=1.0)
inputs = ts.synthetic_batch(cfg, dev, 128, seed=rank)
grads = []
for mode in ("ddp", "buckets"):
    torch.manual_seed(0)
    net = Im2PointCloud(cfg, 128).to(dev)
    model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index]) if mode == "ddp" else net
    red = dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=16) if mode == "buckets" else None
    proj = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=dev)
    out = proj.compute_projection(inputs, proj.replicate_outputs(model(inputs["images"])), is_training=False)
    proj.add_proj_loss(inputs, out, 1.0).backward()
    if red is not None:
        red.finish()
    grads.append([p.grad.clone() for p in net.parameters()])
worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(*grads))
if rank == 0:
    print("WORST", worst)
assert worst < 1e-5, worst
dpc_amd.distributed.finalize()
''' % (root, root)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DPC_POISON_BUFFERS")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # torch.distributed.run takes a script path: write the snippet next to the test outputs
    path = os.path.join(root, "gpurun_out", "_grad_buckets_rccl.py")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fh.write(code)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", path], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "WORST" in r.stdout


# ---- RCCL on the one GPU this tier has: a forced ONE-rank process group (backend nccl).  What it exercises is first
# contact -- communicator creation on the device, the watchdog thread beside a HIP-graph capture, ProcessGroupNCCL's
# stream fork / join as graph edges, RCCL's kernel for the in-collective average, DDP's reducer -- not bytes over xGMI.

def test_bench_one_rank_rccl_projector():
    """`bench.py --gpus 1 --force-dist`: barriers and the MAX all-reduce of the step time through RCCL, the projector's
    step recorded into a HIP graph while the RCCL watchdog thread is alive."""
    j = _bench_multi_gpu(["--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    par = j["config"]["parallelism"]
    assert j["n_gpus"] == 1 and j["config"]["global_batch"] == 32 and j["value"] > 0 and j["config"]["hip_graph"] is True
    assert "backend nccl, world 1, RCCL " in par and "forced one-rank group" in par, par


def test_bench_one_rank_rccl_training_step_ddp_and_recorded():
    """The training step under RCCL with one rank: eager under DistributedDataParallel (its reducer's bucket
    all-reduces), and --graph: GradBuckets' bucket all-reduces (ReduceOp.AVG -- an RCCL kernel even with one rank)
    recorded into the HIP graph beside the live watchdog; the capture must not fall back."""
    e = _bench_multi_gpu(["--gpus", "1", "--force-dist", "--config", "3", "--steps", "5", "--warmup", "3", "--batch", "4",
                          "--no-cpu-baseline"])
    assert e["config"]["training_step"] and "(DDP)" in e["config"]["parallelism"] and e["value"] > 0
    assert "backend nccl, world 1, RCCL " in e["config"]["parallelism"]
    g = _bench_multi_gpu(["--gpus", "1", "--force-dist", "--config", "3", "--steps", "5", "--warmup", "3", "--batch", "4",
                          "--graph", "--no-cpu-baseline"])
    assert "(GradBuckets)" in g["config"]["parallelism"] and "backend nccl, world 1, RCCL " in g["config"]["parallelism"]
    assert g["config"]["hip_graph"] is True, g["config"].get("hip_graph_note")
    assert g["value"] > 0


def test_recording_beside_the_rccl_watchdog_survives_many_recordings():
    """25 recordings of a step that holds RCCL collectives beside the live watchdog thread, an eager barrier before each
    (without distributed.drain_watchdog about one recording in several hundred is killed by a watchdog poll that overlaps
    the capture -- scripts/rccl_capture_stress.py --drain 0 shows it now and then; this is the regression test that
    recording and re-recording keep working with the pause in)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DPC_POISON_BUFFERS",
                                                            "DPC_WATCHDOG_DRAIN_S")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29643")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "rccl_capture_stress.py"), "--records", "25"], env=env,
                       cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "OK: 25 recordings" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])


def test_grad_buckets_equal_ddp_equal_plain_under_one_rank_rccl():
    """One rank, backend nccl: GradBuckets (AVG inside the collective), DistributedDataParallel and the plain
    single-process backward give the same gradients; the recorded GradBuckets step replays to the same values; and RCCL
    really ran (a collective on a fresh tensor changes it as the op says)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "examples", "chair_unsupervised"))
import dpc_amd, train_step as ts
from nets import Im2PointCloud
rank, world, dev = dpc_amd.distributed.init("nccl", force=True)
assert dpc_amd.distributed.active() and world == 1 and dist.get_backend() == "nccl"
print("LIB", dpc_amd.distributed.collective_library())
# RCCL itself: AVG = pre-multiplied sum (a kernel even with one rank), all_gather = a device copy, broadcast, barrier
t = torch.arange(1024, device=dev, dtype=torch.float32)
dist.all_reduce(t, op=dist.ReduceOp.AVG); dist.all_reduce(t, op=dist.ReduceOp.SUM)
out = torch.empty(1024, device=dev); dist.all_gather_into_tensor(out, t); dist.broadcast(out, 0); dist.barrier()
assert torch.equal(out, torch.arange(1024, device=dev, dtype=torch.float32))
cfg = ts.make_cfg(batch_size=2, pc_point_dropout=1.0)
inputs = ts.synthetic_batch(cfg, dev, 128, seed=0)
grads = {}
for mode in ("plain", "ddp", "buckets", "buckets_avg"):       # (buckets: with one rank the all-reduce goes out as a SUM; buckets_avg: RCCL's AVG)
    torch.manual_seed(0)
    net = Im2PointCloud(cfg, 128).to(dev)
    model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index]) if mode == "ddp" else net
    red = (dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=16, average="collective" if mode == "buckets_avg" else "auto")
           if mode.startswith("buckets") else None)
    assert red is None or (red.reduce and red.in_collective_average and len(red.buckets) > 1 and red._avg_op == (mode == "buckets_avg"))
    proj = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=dev)
    def run():
        o = proj.compute_projection(inputs, proj.replicate_outputs(model(inputs["images"])), is_training=False)
        proj.add_proj_loss(inputs, o, 1.0).backward()
        if red is not None:
            red.finish()
    run()
    grads[mode] = [p.grad.clone() for p in net.parameters()]
    if mode.startswith("buckets"):   # the same step recorded (collectives inside the graph) and replayed twice
        red.zero_()
        step = dpc_amd.graphs.RecordedStep(lambda: (red.zero_(), run())[1], world=1, device=dev, collectives=True)
        step(); step()
        torch.cuda.synchronize()
        grads["replayed_" + mode] = [p.grad.clone() for p in net.parameters()]
ref = grads["plain"]
for mode in ("ddp", "buckets", "buckets_avg", "replayed_buckets", "replayed_buckets_avg"):
    worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(grads[mode], ref))
    print("WORST", mode, worst)
    assert worst < 1e-5, (mode, worst)
dpc_amd.distributed.finalize()
print("DONE")
""" % (root, root)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DPC_POISON_BUFFERS")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "DONE" in r.stdout and "backend nccl, world 1, RCCL " in r.stdout, r.stdout


@pytest.mark.parametrize("kw", [dict(), dict(C=1, with_valid=False), dict(B=12, C=2, S=32, rep=3),
                                dict(B=320, C=4, N=8000, D=64, K=21, S=128, rep=20),          # the training step's shape
                                dict(B=32, C=4, N=4000, D=128, K=11, S=128)])
def test_fused_candidate_loss(kw):
    parity_cases.fused_candidate_loss_equals_the_image_epilogue("cuda", **kw)


def test_views_per_cloud_on_device():
    """in-kernel replication (tf_repeat_0 as an index) == explicit copies, at the training step's shape"""
    import dpc_amd as d
    Cc, R, N, D, K = 16, 20, 8000, 64, 21
    inp = synth.make_inputs(Cc * R, N, 99)
    cfg = d.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = d.smoothing_kernel(cfg, 3.0, device="cuda")
    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    w = torch.randn(Cc * R, D, D, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    clouds, pose, scale = t(inp["pc"][::R].copy()), t(inp["pose"]), t(inp["scale"])
    a = d.pointcloud_project_fast(cfg, clouds, pose, None, None, kern, scaling_factor=scale, views_per_cloud=R)
    ga = torch.autograd.grad(a["proj"], [clouds, pose, scale], w)
    clouds2, pose2, scale2 = t(inp["pc"][::R].copy()), t(inp["pose"]), t(inp["scale"])
    b = d.pointcloud_project_fast(cfg, torch.repeat_interleave(clouds2, R, dim=0), pose2, None, None, kern, scaling_factor=scale2)
    gb = torch.autograd.grad(b["proj"], [clouds2, pose2, scale2], w)
    assert float((a["proj"] - b["proj"]).abs().max()) == 0.0
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max())


def test_training_step_full_batch_fused_paths_equal_explicit_paths():
    """BASELINE configs[2] at its FULL model batch (16 models x 5 views x 4 pose candidates = 320 instances of 8000
    points, 64^3, K = 21): one step with the replication and the candidate loss inside the projector's kernels
    (the default) against the same step with the explicit [320,8000,3] copies and the loss epilogue on the image --
    same loss, same winning candidates, same parameter gradients."""
    import os
    import sys
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "chair_unsupervised")
    sys.path.insert(0, ex)
    import train_step as ts
    from nets import Im2PointCloud
    dev = torch.device("cuda")
    res = {}
    for fused in (True, False):
        cfg = ts.make_cfg(batch_size=16, pc_point_dropout=1.0, pc_replicate_in_kernel=fused, pc_fused_proj_loss=fused)
        torch.manual_seed(0)
        net = Im2PointCloud(cfg, 128).to(dev)
        projector = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=dev)
        inputs = ts.synthetic_batch(cfg, dev, 128, seed=0)
        outputs = projector.replicate_outputs(net(inputs["images"]))
        outputs = projector.compute_projection(inputs, outputs, is_training=True)
        assert outputs["projs"].shape == (320, 64, 64, 1)
        assert ("_fused_proj_loss" in outputs) == fused and (outputs.points_replication() is not None) == fused
        loss = projector.add_proj_loss(inputs, outputs, cfg.proj_weight)
        loss.backward()
        res[fused] = (float(loss), outputs["winning_pose_candidates"].cpu().numpy(),
                      [p.grad.detach().clone() for p in net.parameters()])
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * abs(res[False][0])
    assert np.array_equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2], res[False][2]):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-12)


# ---------------------------------------------------------------------------
# round 4
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("sigma,taps", [(3.0, 21), (1.5, 19), (0.8, 9), (0.3, 3)])
def test_training_shape_at_the_schedules_sigmas(sigma, taps):
    """The reference anneals the blur from sigma 3.0 to 0.2 with K = 21 fixed (dpc/models/model_pc.py:33-38,146-153,
    dpc/util/gauss_kernel.py:5-11); the library runs the tap count the sigma still needs (outer taps below 1e-8 of the
    centre tap are dropped: 21 -> 19 -> 9 -> 3 here, with the xy-saving state layout at <= 11).  8 views of the
    training shape per sigma, forward and all gradients, against oracle/reference_cpu.py running the FULL 21 taps in
    float64: silhouette <= 2e-5, gradients at the bounds of every other full-batch comparison."""
    synth.CONFIGS[3] = dict(TRAIN_SHAPE, N=8000, sigma=sigma)
    c = synth.config_inputs(3)
    cfg = dpc_amd.default_config(vox_size=64, pc_gauss_kernel_size=21)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device="cuda")
    assert dpc_amd.util.point_cloud.effective_tap_counts(cfg, kern) == (taps,) * 3
    c8 = {k: (v[:8] if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    worst = _against_reference_cpu(c8, 8, chunk=4)
    assert worst["proj"] <= 2e-5, worst
    # and the trimmed run against the SAME kernels' untrimmed run, whole batch of 320: the dropped taps are below fp32 rounding
    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    res = []
    for trim in (True, False):
        cfg_t = dpc_amd.default_config(vox_size=64, pc_gauss_kernel_size=21, pc_trim_gauss_taps=trim)
        pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
        out = dpc_amd.pointcloud_project_fast(cfg_t, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg_t, sigma, device="cuda"),
                                              scaling_factor=scale)
        gt = torch.tensor(synth.disk_gt(320, 64), device="cuda")
        g = torch.autograd.grad(out["proj"], [pc, pose, scale], ((out["proj"] - gt) / 320).detach())
        res.append((out["proj"].detach(), g))
    assert float((res[0][0] - res[1][0]).abs().max()) <= 3e-7
    for a, b in zip(res[0][1], res[1][1]):
        # (a few voxels sit within the dropped taps' 1e-9 of the eps-clip at 1e-5 and switch piece: the bound of every other gradient check)
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()), sigma


def test_recorded_step_follows_the_tap_counts():
    """dpc_amd.graphs.RecordedStep: the projector step of a ModelPointCloud whose sigma is annealed 3.0 -> 0.2, recorded
    into a HIP graph, re-recorded whenever the effective tap count moves; after every call the gradients equal an eager
    step at that sigma bit for bit (same kernels, integer splat)."""
    dev = torch.device("cuda")
    cfg = dpc_amd.default_config(vox_size=64, pc_gauss_kernel_size=21, pc_relative_sigma=3.0, pc_relative_sigma_end=0.2,
                                 max_number_of_steps=40)
    m = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=dev)
    m.enable_graph_replay(follow_tap_counts=True)
    inp = synth.make_inputs(6, 3000, 17)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    gt = torch.tensor(synth.disk_gt(6, 64), device=dev)

    def run(kernel=None):
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kernel or m.gauss_kernel(), scaling_factor=scale,
                                              l2_target=(gt, 1.0 / 6))
        return torch.autograd.grad(out["proj"], [pc, pose, scale], out["proj_l2_grad"])

    step = dpc_amd.graphs.RecordedStep(run, world=1, device=dev, key=m.effective_tap_counts)
    seen = []
    for gs in range(0, 41, 2):
        m.set_global_step(gs)
        got = [g.clone() for g in step()]
        seen.append(m.effective_tap_counts()[0])
        sigma = dpc_amd.model_pc.get_smooth_sigma(cfg, gs)
        want = run(dpc_amd.smoothing_kernel(cfg, sigma, device=dev))
        # point and scale gradients bit for bit (integer splat, fixed-order sums); the pose gradient is summed over
        # work-groups with float atomics, whose order varies from launch to launch
        assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]), (gs, sigma)
        assert float((got[1] - want[1]).abs().max()) <= 1e-5 * float(want[1].abs().max()), (gs, sigma)
    assert seen[0] == 21 and seen[-1] == 3 and all(a >= b for a, b in zip(seen, seen[1:]))
    assert step.records == len(set(seen))                 # one recording per tap count met


@pytest.mark.parametrize("args", [["--config", "3", "--projector-only", "--sigma", "0.8", "--batch", "40"],
                                  ["--k", "15", "--sigma", "2.5", "--batch", "4"], ["--vox", "48", "--batch", "4"]])
def test_bench_workload_switches(args):
    """bench.py --sigma / --k / --vox (the lines under profiles/r04 come from these): the line names the tap count that
    ran and keeps every fraction physical."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("DPC_POISON_BUFFERS", "DPC_TEST_HOOKS", "DPC_BENCH_DRY_RUN", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--repeats", "2",
                        "--no-cpu-baseline", "--burn-in", "0"] + args, env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    want = {"0.8": 9, "2.5": 15}.get(args[args.index("--sigma") + 1] if "--sigma" in args else "", 11)
    assert j["config"]["taps_run"] == want, j["config"]
    rf = j["roofline"]
    assert 0 < rf["frac"] <= 1.0 and 0 < rf["step_frac"] <= 1.0


def test_compiled_binding_equals_ctypes_binding_on_the_device(monkeypatch):
    """csrc/dpc_torch.cpp against ops.ProjectFused on the GPU: same library, same stream, same bits (point / scale
    gradients and images exactly; the pose gradient to the float-atomic order), eagerly and inside a HIP graph."""
    if dpc_amd._ext.module() is None:
        pytest.skip("compiled binding not built (python __graft_entry__.py)")
    dev = torch.device("cuda")
    c = synth.config_inputs(1)
    cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device=dev)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
    gt = torch.tensor(synth.disk_gt(c["B"], c["D"]), device=dev)

    def run():
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, l2_target=(gt, 0.25))
        return [out["proj"], out["proj_depth"]] + list(torch.autograd.grad(out["proj"], [pc, pose, scale], out["proj_l2_grad"]))

    res = {}
    for binding in ("compiled", "ctypes"):
        monkeypatch.setenv("DPC_BINDING", "" if binding == "compiled" else "ctypes")
        dpc_amd._ext.reset()
        assert (dpc_amd._ext.module() is None) == (binding == "ctypes")
        # (detached copies: a live autograd graph of an earlier EAGER call keeps the leaves' gradient accumulators bound to
        # the default stream, which torch then tries to synchronise with from inside the capture -- torch's own warning
        # "AccumulateGrad node's stream does not match ... may break CUDA graph capture"; on ROCm it segfaults at capture end)
        res[binding] = [x.detach().clone() for x in run()]
        if binding == "compiled":
            step = dpc_amd.graphs.RecordedStep(run, world=1, device=dev)
            res["recorded"] = [x.detach().clone() for x in step()]
    monkeypatch.delenv("DPC_BINDING")
    dpc_amd._ext.reset()
    for other in ("ctypes", "recorded"):
        for i, (a, b) in enumerate(zip(res["compiled"], res[other])):
            if i == 3:       # dpose: float atomics across work-groups
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), other
            else:
                assert torch.equal(a, b), (other, i)


def test_cfg2_full_batch_unnudged_against_the_fp32_reference():
    """BASELINE configs[1] at its full batch with NO input moved: the HIP path against oracle/reference_cpu.py run in
    float32 -- the precision the reference itself computes in -- on the very same 256 000 points.  Both sides then see the
    same rounded lattice coordinates, so the piecewise structure (cell faces, the clip at G0 = 1, the eps-clip of the ray
    collapse) is entered alike except where the two summation orders differ in the last bit; those entries are COUNTED
    and bounded instead of being nudged away: transformed points and cells must agree exactly, silhouettes to 2e-5, and at
    most 2e-3 of the point-gradient entries may sit beyond the elementwise bound (none beyond 50 bounds: a point whose
    cell sum crosses 1.0 the other way loses or gains a whole clip-gated term)."""
    c = synth.config_inputs(2)
    B, D = c["B"], c["D"]
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=c["K"])
    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    pc, pose, scale = t(c["pc"]), t(c["pose"]), t(c["scale"])
    kern = dpc_amd.smoothing_kernel(cfg, c["sigma"], device="cuda")
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    gt = torch.tensor(synth.disk_gt(B, D), device="cuda")
    dproj = ((out["proj"] - gt) / B).detach()
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], dproj)
    rc = rcpu.Cfg(vox_size=D, pc_gauss_kernel_size=c["K"])
    ckern = rcpu.smoothing_kernel(rc, c["sigma"], torch.float32)
    tr_diff = cell_diff = 0
    worst_proj = 0.0
    ref_dpc, ref_dpose, ref_dscale = [], [], []
    for lo in range(0, B, 8):
        hi = lo + 8
        d = lambda a: torch.tensor(a[lo:hi], dtype=torch.float32, requires_grad=True)
        cpc, cpose, cscale = d(c["pc"]), d(c["pose"]), d(c["scale"])
        ref = rcpu.pointcloud_project_fast(rc, cpc, cpose, None, None, ckern, scaling_factor=cscale)
        rg = torch.autograd.grad(ref["proj"], [cpc, cpose, cscale], dproj[lo:hi].cpu())
        a, b = out["tr_pc"][lo:hi].detach().cpu().numpy(), ref["tr_pc"].detach().numpy()
        tr_diff += int((a != b).any(-1).sum())
        cell = lambda x: np.floor((x + np.float32(0.5)) * np.float32(D - 1))
        cell_diff += int((cell(a) != cell(b)).any(-1).sum())
        worst_proj = max(worst_proj, maxabs(out["proj"][lo:hi].detach().cpu().numpy(), ref["proj"].detach().numpy()))
        ref_dpc.append(rg[0].numpy())
        ref_dpose.append(rg[1].numpy())
        ref_dscale.append(rg[2].numpy())
    npts = B * c["N"]
    ok, ratio, frac = close_elementwise_piecewise(g[0].cpu().numpy(), np.concatenate(ref_dpc), outliers=2e-3, cap=50.0)
    print("un-nudged fp32 comparison: %d of %d transformed points differ in a bit, %d land in another cell; proj max-abs %.2e; "
          "dpc entries beyond the bound: %.2e (worst %.1f bounds)" % (tr_diff, npts, cell_diff, worst_proj, frac, ratio))
    assert cell_diff == 0 and tr_diff <= 1e-3 * npts, (tr_diff, cell_diff)
    assert worst_proj < TOL_PROJ, worst_proj
    assert ok, (ratio, frac)
    assert relerr(g[1].cpu().numpy(), np.concatenate(ref_dpose)) < TOL_GRAD
    assert relerr(g[2].cpu().numpy(), np.concatenate(ref_dscale)) < TOL_GRAD
