"""Parity checks shared by the CPU emulation tier (tests/test_emu_kernels.py,
device "cpu" + emulation library) and the GPU tier (tests/test_gpu_parity.py,
device "cuda" + libdpc_hip.so)."""
import pytest
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


FUSED_CASES = [
    # (B, N, D, Dz, K, sigma, with_trans, with_focal)  -- shapes that take the fused LDS splat / gather path
    (3, 300, 64, 64, 11, 1.3, True, False),      # B not a multiple of 8, N not a multiple of 64
    (2, 500, 64, 32, 11, 1.5, False, True),      # vox_size_z != vox_size: Kz = 5 (gauss_kernel.py:38-50)
    (2, 400, 64, 64, 21, 3.0, False, False),     # the shipped experiments' kernel size
    (2, 40, 64, 64, 5, 0.8, False, False),       # fewer points than one wave
    (1, 8300, 32, 32, 5, 0.8, True, False),      # more than 8 x 1024 points: the two-kernel depth sort; > 512 points per plane
    # round 4: grids narrower than the power-of-two lane geometry (rows padded inside the kernels)
    (2, 600, 48, 48, 5, 0.9, False, False),      # 48 on the 64-wide geometry: one strip per plane
    (2, 300, 24, 24, 5, 0.8, True, False),       # 24 on the 32-wide one
    # round 6: widths that end inside a lane of that geometry (any multiple of 4: the padding is masked per 16-byte vector)
    (1, 400, 68, 68, 5, 1.0, True, True),        # 68 on the 128-wide geometry (8 floats per lane): two strips
    (2, 300, 36, 36, 7, 1.0, False, False),      # 36 on the 64-wide one
]


FUSED_CASES_GPU = FUSED_CASES + [
    (1, 9000, 64, 64, 11, 1.6, False, False),    # more than 8 x 1024 points: k_zsort's strided (unbatched) branch
    (9, 200, 128, 128, 5, 1.0, True, True),      # B = 9 (odd XCD split), K = 5 at 128^3
    (1, 2000, 128, 128, 21, 3.0, False, False),  # K = 21 at 128^3
    (2, 300, 32, 32, 11, 1.2, False, False),     # smallest fused lattice
    (1, 1500, 128, 64, 11, 1.6, False, False),   # vox_size_z = vox_size / 2 at 128: Kz = 5
    # round 3: one-strip planes with more than 128 points -> k_gather_yx's dense flow (x-blur adjoint in registers while staging)
    (2, 6000, 32, 32, 21, 3.0, False, False),    # D = 32 (2 floats per lane), 21 taps: halo from 5 lanes away
    (2, 8000, 64, 64, 11, 1.6, True, True),      # D = 64, 11 taps, with translation and focal length
    (1, 8000, 64, 64, 5, 0.9, False, False),     # D = 64, 5 taps
    # round 4: the tap counts that gained compiled kernels (what the annealed sigma trims a 21-tap filter to)
    (2, 600, 64, 64, 3, 0.6, False, False),
    (2, 500, 32, 32, 7, 1.2, True, False),
    (2, 700, 64, 64, 9, 1.5, False, True),
    (1, 3000, 128, 128, 13, 2.2, False, False),
    (2, 8000, 64, 64, 15, 2.5, False, False),    # dense gather flow at 15 taps
    (1, 4000, 64, 64, 17, 2.9, False, False),
    (1, 1500, 128, 128, 19, 3.2, False, False),
    (1, 1200, 128, 64, 19, 3.2, False, False),   # vox_size_z = 64 at 128: Kz = 9
    (2, 3000, 96, 96, 11, 1.6, False, False),    # 96 on the 128-wide geometry: two strips, the second half empty below row 96
    (2, 6000, 48, 48, 21, 3.0, True, True),      # 48 on 64 with 21 taps: dense gather flow on padded rows
    (1, 2000, 80, 80, 9, 1.4, False, False),     # 80 on 128 (10 of 16 lanes)
    (1, 3000, 160, 64, 7, 1.2, False, False),    # 160 on 256, shallow grid
    # round 5: the tap counts beyond 21 (any odd pc_gauss_kernel_size is a legal configuration, default_config.yaml:55)
    (1, 3000, 128, 128, 23, 4.0, False, False),
    (2, 6000, 64, 64, 31, 5.0, True, False),     # widest compiled filter, dense gather flow
    (1, 2000, 128, 64, 27, 4.5, False, True),    # vox_size_z = 64 at 128: Kz = 13
    (1, 1500, 32, 32, 25, 4.2, False, False),    # 25 taps on 32-wide rows: the halo comes from 12 columns away
    (1, 2500, 96, 96, 29, 4.8, False, False),    # padded rows
    # round 6: widths that end inside a lane (the reference accepts any vox_size, default_config.yaml:77)
    (4, 8000, 100, 100, 11, 1.6, True, False),   # bench.py --vox 100
    (2, 8000, 200, 200, 11, 2.0, False, True),   # 200 on the 256-wide geometry (16 floats per lane), seven strips
    (2, 6000, 52, 52, 21, 3.0, False, False),    # 52 on 64 with 21 taps: dense gather flow
    (1, 3000, 132, 132, 5, 1.0, False, False),   # 132 on 256: half the geometry empty
    (1, 2000, 100, 40, 9, 1.4, False, False),    # vox_size_z != vox_size on such a width
]
DENSE_GATHER_CASE_EMU = (1, 3000, 32, 32, 5, 0.8, False, False)    # ~150+ points per occupied plane at D = 32


def _nudge_off_cell_faces(inp, trans, focal, Dz, D, tol=3e-5):
    """Gradient checks against the fp64 oracle need inputs away from the two places where the
    function is only piecewise smooth and fp32 rounding can pick the other piece:
      * the trilinear gradient jumps across cell faces (lattice coordinate within rounding of an
        integer), and
      * clip_by_value(G0, 0, 1) switches the gradient off where piled-up points sum to 1 +- rounding.
    Points involved in either coincidence are moved a little (0.3 %), until none is left."""
    f64 = lambda a: None if a is None else a.astype(np.float64)
    pc = inp["pc"].copy()
    moved = np.zeros(pc.shape[:2], dtype=bool)
    size = np.array([Dz - 1, D - 1, D - 1], dtype=np.float64)
    for _ in range(12):
        tr = onp.transform_fwd(f64(pc), f64(inp["pose"]), f64(trans), f64(focal))
        g = (tr + 0.5) * size
        bad = (np.abs(g - np.rint(g)) < tol).any(-1)
        G0 = onp.voxelize_fwd(tr, Dz, D)
        knife = np.abs(G0 - 1.0) < tol
        if knife.any():
            lo = np.floor(g).astype(np.int64)
            inside = ((tr >= -0.5) & (tr <= 0.5)).all(-1)
            for b, z, y, x in zip(*np.nonzero(knife)):
                d = np.array([z, y, x]) - lo[b]
                bad[b] |= inside[b] & ((d >= 0) & (d <= 1)).all(-1)
        if not bad.any():
            break
        pc[bad] *= np.float32(1.003)
        moved |= bad
    inp["pc"] = pc
    inp["moved_points"] = int(moved.sum())          # callers assert / report how many inputs were touched
    return inp


def fused_path_against_numpy_oracle(dev, B, N, D, Dz, K, sigma, with_trans, with_focal):
    """pointcloud_project_fast on shapes that use k_zsort/k_splat_xy/k_gather_yx,
    forward and all gradients (incl. depth upstream) vs the float64 NumPy oracle."""
    rng = np.random.default_rng(1000 + B * 7 + N)
    inp = synth.make_inputs(B, N, 900 + N)
    trans = (0.04 * rng.standard_normal((B, 3))).astype(np.float32) if with_trans else None
    focal = rng.uniform(1.7, 2.1, (B, 1)).astype(np.float32) if with_focal else None
    inp = _nudge_off_cell_faces(inp, trans, focal, Dz, D)
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=(Dz if Dz != D else -1), pc_gauss_kernel_size=K)
    lib = dpc_amd.get_library()
    Kz = len(onp.smoothing_taps(D, Dz if Dz != D else -1, K, sigma)[2])
    S = dpc_amd._capi.DpcShape(B, N, Dz, D, K, K, Kz)
    P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0)
    import ctypes
    assert lib.dpc_saved_layout(ctypes.byref(S), ctypes.byref(P)) & 6 == 6, "expected the fused path for this shape"
    t = lambda a: None if a is None else torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale, ttrans, tfocal = t(inp["pc"]), t(inp["pose"]), t(inp["scale"]), t(trans), t(focal)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, ttrans, None, kern, scaling_factor=scale, focal_length=tfocal)
    w = rng.standard_normal(out["proj"].shape)
    wd = 0.1 * rng.standard_normal(out["proj"].shape)
    loss = (out["proj"] * torch.tensor(w, dtype=torch.float32, device=dev)).sum() + \
           (out["proj_depth"] * torch.tensor(wd, dtype=torch.float32, device=dev)).sum()
    leaves = [x for x in (pc, pose, scale, ttrans, tfocal) if x is not None]
    grads = torch.autograd.grad(loss, leaves)
    f64 = lambda a: None if a is None else a.astype(np.float64)
    taps = onp.smoothing_taps(D, Dz if Dz != D else -1, K, sigma)
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), f64(trans), f64(inp["scale"]), f64(focal), taps, Dz=Dz, D=D)
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), f64(trans), f64(inp["scale"]), f64(focal), taps, fw,
                              dproj=w, dproj_depth=wd)
    assert maxabs(out["proj"].detach().cpu().numpy(), fw["proj"]) < TOL_PROJ
    assert maxabs(out["proj_depth"].detach().cpu().numpy(), fw["proj_depth"]) < TOL_DEPTH
    names = ["dpc", "dpose", "dscale"] + (["dtrans"] if with_trans else []) + (["dfocal"] if with_focal else [])
    for name, g in zip(names, grads):
        assert relerr(g.cpu().numpy().reshape(bw[name].shape), bw[name]) < TOL_GRAD, name


def rgb_case_matches_goldens(dev, name):
    """pc_rgb branch (point_cloud.py:111-118,244-262,275-279; drc.py:126-136) against the
    goldens produced by the reference's own source."""
    from helpers import load
    from run_case import run_product
    g = load(name)
    res, gr = run_product(name, g, dev, grads=True)
    assert maxabs(res["proj"], g["proj_f64"]) < TOL_PROJ
    assert maxabs(res["proj_rgb"], g["proj_rgb_f64"]) < 5e-5
    assert maxabs(res["voxels_rgb"], g["voxels_rgb_f64"]) < 5e-5
    for k in ("dpc", "dpose", "dtrans", "dscale", "drgb"):
        assert relerr(gr[k], g[k + "_f64"]) < TOL_GRAD, k


# (the *_bicubic cases: cfg.bicubic_gt_downsampling -- the resize is a restatement of TF's ResizeBicubic on both sides
# of the comparison (util/losses.resize_images_bicubic_tf1, oracle/tf_shim); what pins BOTH to the op is
# tests/test_tf_shim.py::test_resize_images_bicubic_hand_worked_vectors, not this comparison)
LOSS_CASES = ["c1_same", "c1_resize", "c2_x2", "c3_ratio", "c4_valid", "c1_bicubic", "c2_bicubic_x2"]


def silhouette_loss_matches_reference(dev, name):
    """Loss epilogue (model_pc.py:308-337,383-423) against goldens produced by the reference's
    own add_proj_loss: loss value, winning candidates (exact), gradient wrt the projections."""
    import dpc_amd
    from dpc_amd import model_pc as M
    from helpers import load
    g = load("caller_loss")
    B, C, D, S, var = (int(v) for v in g[name + "_meta"][:5])
    cfg = dpc_amd.default_config(vox_size=D, pose_predict_num_candidates=C, variable_num_views=bool(var),
                                 pose_predictor_student=False, bicubic_gt_downsampling=len(g[name + "_meta"]) > 5)
    model = M.ModelPointCloud(cfg, global_step=0, device=dev)
    pred = torch.tensor(g[name + "_pred"], device=dev, requires_grad=True)
    inputs = {"masks": torch.tensor(g[name + "_gt"], device=dev),
              "valid_samples": torch.tensor(g[name + "_valid"], device=dev)}
    outputs = {"projs": pred}
    loss = model.add_proj_loss(inputs, outputs, 1.0)
    (3.0 * loss).backward()                                   # non-unit upstream gradient
    ref = float(g[name + "_loss_f64"])
    assert abs(float(loss) - ref) < 2e-6 * max(ref, 1.0), (float(loss), ref)
    assert relerr(pred.grad.cpu().numpy() / 3.0, g[name + "_dpred_f64"]) < 1e-5
    if C > 1:
        assert np.array_equal(outputs["winning_pose_candidates"].cpu().numpy(), g[name + "_winners"])
        assert outputs["winning_pose_candidates"].dtype == torch.int64


NN_CASES = [(n, t) for n in ("rand", "ties", "one_target", "small_src") for t in ("f64", "f32")]


def nn_distance_matches_reference(dev, name, tag):
    """point_cloud_distance (point_cloud_distance.py:26-39) against goldens from the reference's own
    source: indices and projections bit-exact (first minimum on ties); distances to 1 ulp (the
    summation order of reduce_sum over the 3 components is the backend's choice)."""
    from dpc_amd.util.point_cloud_distance import chamfer_distance, point_cloud_distance
    from helpers import load
    g = load("nn_distance")
    vs = torch.tensor(g["%s_vs_%s" % (name, tag)], device=dev)
    vt = torch.tensor(g["%s_vt_%s" % (name, tag)], device=dev)
    proj, dist, idx = point_cloud_distance(vs, vt)
    assert idx.dtype == torch.int32 and dist.dtype == vs.dtype
    assert np.array_equal(idx.cpu().numpy(), g["%s_idx_%s" % (name, tag)])
    assert np.array_equal(proj.cpu().numpy(), g["%s_proj_%s" % (name, tag)])
    ref = g["%s_dist_%s" % (name, tag)]
    assert np.all(np.abs(dist.cpu().numpy() - ref) <= np.spacing(ref))
    a, b = chamfer_distance(vs, vt)
    assert abs(float(a) - g["%s_dist_%s" % (name, tag)].astype(np.float64).mean()) < 1e-6


def nn_distance_gradient(dev):
    """Gradient of sum(minDist) + <w, proj> equals torch autograd through an explicit gather."""
    from dpc_amd.util.point_cloud_distance import point_cloud_distance
    gen = torch.Generator().manual_seed(5)
    vs = torch.rand(90, 3, generator=gen, dtype=torch.float64).to(dev).requires_grad_(True)
    vt = torch.rand(130, 3, generator=gen, dtype=torch.float64).to(dev).requires_grad_(True)
    w = torch.rand(90, 3, generator=gen, dtype=torch.float64).to(dev)
    proj, dist, idx = point_cloud_distance(vs, vt)
    (dist.sum() + (w * proj).sum()).backward()
    g_vs, g_vt = vs.grad.clone(), vt.grad.clone()
    vs.grad = vt.grad = None
    sel = vt[idx.to(torch.int64)]
    ((sel - vs).pow(2).sum(1).sqrt().sum() + (w * sel).sum()).backward()
    assert maxabs(g_vs.cpu().numpy(), vs.grad.cpu().numpy()) < 1e-12
    assert maxabs(g_vt.cpu().numpy(), vt.grad.cpu().numpy()) < 1e-12


SLOW_VOX_CASES = ["vox_analytical", "vox_none", "vox_sum"]


def _slow_cfg(G, nsum=False, nana=True):
    return dpc_amd.default_config(vox_size=G, pc_normalise_gauss=bool(nsum), pc_normalise_gauss_analytical=bool(nana),
                                  pc_fast=False)


def gauss_voxeliser_matches_reference(dev, name):
    """pointcloud2voxels (point_cloud.py:17-57) against goldens from the reference's own source:
    the three normalisation modes, clipped pile-ups, gradient of a random functional."""
    from dpc_amd.util.point_cloud import pointcloud2voxels
    from helpers import load
    g = load("slow_path")
    B, N, G, nsum, nana = (int(v) for v in g[name + "_meta"])
    pc = torch.tensor(g[name + "_pc"], device=dev, requires_grad=True)
    vox = pointcloud2voxels(_slow_cfg(G, nsum, nana), pc, float(g[name + "_sigma"]))
    assert vox.shape == (B, G, G, G, 1)
    assert maxabs(vox.detach().cpu().numpy(), g[name + "_vox_f64"]) < 1e-5
    (vox * torch.tensor(g[name + "_w"], device=dev)).sum().backward()
    assert relerr(pc.grad.cpu().numpy(), g[name + "_dpc_f64"]) < TOL_GRAD


def slow_projector_matches_reference(dev):
    """pointcloud_project (point_cloud.py:219-226, cfg.pc_fast:false) end to end."""
    from dpc_amd.util.point_cloud import pointcloud_project
    from helpers import load
    g = load("slow_path")
    B, N, G = (int(v) for v in g["proj_meta"])
    pc = torch.tensor(g["proj_pc"], device=dev, requires_grad=True)
    pose = torch.tensor(g["proj_pose"], device=dev, requires_grad=True)
    proj, vox = pointcloud_project(_slow_cfg(G), pc, pose, float(g["proj_sigma"]))
    assert proj.shape == (B, G, G, 1) and vox.shape == (B, G, G, G, 1)
    assert maxabs(vox.detach().cpu().numpy(), g["proj_vox_f64"]) < 1e-5
    assert maxabs(proj.detach().cpu().numpy(), g["proj_proj_f64"]) < TOL_PROJ
    (proj * torch.tensor(g["proj_w"], device=dev)).sum().backward()
    assert relerr(pc.grad.cpu().numpy(), g["proj_dpc_f64"]) < TOL_GRAD
    assert relerr(pose.grad.cpu().numpy(), g["proj_dpose_f64"]) < TOL_GRAD


def gauss_voxeliser_multitile_against_numpy_oracle(dev, B=1, N=100, G=70, sigma=0.05, mode="analytical"):
    """lattice wider than one 64-node tile, N not a multiple of the point chunk: against the
    NumPy fp64 restatement (itself pinned to the reference goldens in test_oracle.py)."""
    gen = np.random.default_rng(12)
    pc = gen.uniform(-0.95, 0.95, (B, N, 3)).astype(np.float32)
    w = gen.standard_normal((B, G, G, G)).astype(np.float32)
    ref, raw = onp.gauss_voxelize_fwd(pc, G, sigma, mode)
    dref = onp.gauss_voxelize_bwd(pc, G, sigma, raw, w.astype(np.float64), mode)
    t = torch.tensor(pc, device=dev, requires_grad=True)
    nmode = {None: 0, "sum": 1, "analytical": 2}[mode]
    vox = dpc_amd.ops.GaussVoxelize.apply(t, sigma, G, (1, 0, 2), nmode)
    assert maxabs(vox.detach().cpu().numpy(), ref) < 1e-5 * max(1.0, float(ref.max()))
    (vox * torch.tensor(w, device=dev)).sum().backward()
    assert relerr(t.grad.cpu().numpy(), dref) < TOL_GRAD


def fused_edge_planes_against_numpy_oracle(dev):
    """Plane-occupancy edge cases on the fused path (D = 32): a view whose points all fall outside the
    cube (every plane empty), and a view with mass only in the first and the last depth cells."""
    B, N, D, K, sigma = 2, 48, 32, 5, 1.0
    rng = np.random.default_rng(3)
    pc = np.zeros((B, N, 3), np.float32)
    pc[0] = rng.uniform(1.5, 2.0, (N, 3))                                   # all outliers
    pc[1, :, 1:] = rng.uniform(-0.2, 0.2, (N, 2))
    pc[1, : N // 2, 0] = -0.5 + rng.uniform(1e-3, 5e-3, N // 2)             # depth cell 0
    pc[1, N // 2:, 0] = 0.5 - rng.uniform(1e-3, 5e-3, N - N // 2)           # depth cell Dz-2 (planes Dz-2, Dz-1)
    pose = np.tile(np.array([[1.0, 0.0, 0.0, 0.0]], np.float32), (B, 1))    # identity: depth = first component
    scale = np.array([[0.9], [0.7]], np.float32)
    inp = _nudge_off_cell_faces(dict(pc=pc, pose=pose, scale=scale), None, None, D, D)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    tpc, tpose, tscale = t(inp["pc"]), t(pose), t(scale)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    out = dpc_amd.pointcloud_project_fast(cfg, tpc, tpose, None, None, kern, scaling_factor=tscale)
    w = rng.standard_normal(out["proj"].shape)
    wd = 0.1 * rng.standard_normal(out["proj"].shape)
    loss = (out["proj"] * torch.tensor(w, dtype=torch.float32, device=dev)).sum() + \
           (out["proj_depth"] * torch.tensor(wd, dtype=torch.float32, device=dev)).sum()
    grads = torch.autograd.grad(loss, [tpc, tpose, tscale])
    f64 = lambda a: a.astype(np.float64)
    taps = onp.smoothing_taps(D, -1, K, sigma)
    fw = onp.project_forward(f64(inp["pc"]), f64(pose), None, f64(scale), None, taps, D, D)
    bw = onp.project_backward(f64(inp["pc"]), f64(pose), None, f64(scale), None, taps, fw, dproj=w, dproj_depth=wd)
    assert maxabs(out["proj"].detach().cpu().numpy(), fw["proj"]) < TOL_PROJ
    assert maxabs(out["proj_depth"].detach().cpu().numpy(), fw["proj_depth"]) < TOL_DEPTH
    empty_ray = 1.0 - (1.0 - 1e-5) ** D                                       # every ray of view 0
    assert abs(float(out["proj"][0].max()) - empty_ray) < 1e-6 and abs(float(out["proj"][0].min()) - empty_ray) < 1e-6
    for name, g in zip(("dpc", "dpose", "dscale"), grads):
        assert relerr(g.cpu().numpy().reshape(bw[name].shape), bw[name]) < TOL_GRAD, name
    assert float(grads[0][0].abs().max()) == 0.0                              # outliers get no gradient


# ---------------------------------------------------------------------------
# round-2 cases
# ---------------------------------------------------------------------------
def _asym_kernel(K, seed, dtype, dev="cpu"):
    """Three DIFFERENT, ASYMMETRIC, normalised separable filters in the reference's conv3d filter layout."""
    rng = np.random.default_rng(seed)
    ks = []
    for shape in ((1, 1, K, 1, 1), (1, K, 1, 1, 1), (K, 1, 1, 1, 1)):
        t = rng.uniform(0.05, 1.0, K) * np.linspace(0.3, 1.7, K)      # skewed to one side
        t = (t / t.sum()).astype(np.float32)
        ks.append(torch.tensor(t.reshape(shape), dtype=dtype, device=dev))
    return ks


ASYM_CASES = [(32, 5), (24, 5), (24, 7)]       # fused path; generic plane path with fixed z taps; run-time taps


def asymmetric_filters_against_cpu_oracle(dev, D, K):
    """smoothen_voxels3d / pointcloud_project_fast accept ARBITRARY separable filters (conv3d is a
    cross-correlation): forward and every gradient with three different skewed filters against the
    torch-CPU restatement of the reference graph.  A backward pass that re-applied the forward taps
    unreversed (correct only for symmetric Gaussians) fails this by ~100 %."""
    from helpers import close_elementwise
    B, N = 2, 300
    inp = synth.make_inputs(B, N, 4100 + D + K)
    inp = _nudge_off_cell_faces(inp, None, None, D, D)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    rc = rcpu.Cfg(vox_size=D, pc_gauss_kernel_size=K)
    kern = _asym_kernel(K, 7 + D, torch.float32, dev)
    ckern = _asym_kernel(K, 7 + D, torch.float64)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    c = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    cpc, cpose, cscale = c(inp["pc"]), c(inp["pose"]), c(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    ref = rcpu.pointcloud_project_fast(rc, cpc, cpose, None, None, ckern, scaling_factor=cscale)
    assert maxabs(out["proj"].detach().cpu().numpy(), ref["proj"].detach().numpy()) < TOL_PROJ
    w = np.random.default_rng(D).standard_normal(tuple(out["proj"].shape))
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], torch.tensor(w, dtype=torch.float32, device=dev))
    rg = torch.autograd.grad(ref["proj"], [cpc, cpose, cscale], torch.tensor(w))
    for name, a, b in zip(("dpc", "dpose", "dscale"), g, rg):
        ok, worst = close_elementwise(a.cpu().numpy(), b.numpy(), rtol=2e-3, atol_frac=5e-5)
        assert ok, (name, worst)
    # stage-level blur and its adjoint with the same filters
    vox = torch.tensor(np.random.default_rng(1).uniform(0, 1, (B, D, D, D, 1)).astype(np.float32), device=dev,
                       requires_grad=True)
    cvox = vox.detach().cpu().double().requires_grad_(True)
    sm = dpc_amd.smoothen_voxels3d(cfg, vox, kern)
    csm = rcpu.smoothen_voxels3d(rc, cvox, ckern)
    assert maxabs(sm.detach().cpu().numpy(), csm.detach().numpy()) < 1e-5
    wv = np.random.default_rng(2).standard_normal(tuple(sm.shape))
    gv, = torch.autograd.grad(sm, [vox], torch.tensor(wv, dtype=torch.float32, device=dev))
    cgv, = torch.autograd.grad(csm, [cvox], torch.tensor(wv))
    assert maxabs(gv.cpu().numpy(), cgv.numpy()) < 2e-5 * float(np.abs(cgv.numpy()).max())


def fused_dropout_equals_explicit_subset(dev, B=3, N=420, D=32, K=5, keep=137, seed=20260927, extras=True):
    """pointcloud_project_fast(point_dropout=(keep, seed)) == the projector run on the explicitly gathered
    subset that oracle/dropout_ref.py predicts (per-instance, exactly `keep` points, without replacement):
    forward image, gradients of the kept points, exact zeros for the dropped ones."""
    from oracle import dropout_ref
    inp = synth.make_inputs(B, N, 5150)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, 0.9, device=dev)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale,
                                          point_dropout=(keep, seed))
    w = torch.tensor(np.random.default_rng(9).standard_normal(tuple(out["proj"].shape)).astype(np.float32), device=dev)
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], w)

    mask = dropout_ref.kept_mask(B, N, keep, seed)
    assert (mask.sum(1) == keep).all() and not (mask[0] == mask[1]).all()      # exact count, per-instance draws
    idx = np.stack([np.nonzero(m)[0] for m in mask])                            # [B, keep]
    sub = torch.tensor(np.take_along_axis(inp["pc"], idx[:, :, None], axis=1), device=dev, requires_grad=True)
    pose2, scale2 = t(inp["pose"]), t(inp["scale"])
    ref = dpc_amd.pointcloud_project_fast(cfg, sub, pose2, None, None, kern, scaling_factor=scale2)
    rg = torch.autograd.grad(ref["proj"], [sub, pose2, scale2], w)
    # integer (order-independent) splat + identical blur / collapse kernels: the images agree to rounding of
    # nothing at all; allow one ulp-level slack for the block-reduction order of the pose sums only
    assert float((out["proj"] - ref["proj"]).abs().max()) <= 1e-7
    gpc = g[0].cpu().numpy()
    assert np.all(gpc[~mask] == 0.0)                                            # dropped points: exactly zero
    assert maxabs(np.take_along_axis(gpc, idx[:, :, None], axis=1), rg[0].cpu().numpy()) <= 1e-7 * max(1.0, float(np.abs(gpc).max()))
    assert relerr(g[1].cpu().numpy(), rg[1].cpu().numpy()) < 1e-5
    assert relerr(g[2].cpu().numpy(), rg[2].cpu().numpy()) < 1e-5
    assert out["tr_pc"].shape == (B, N, 3)
    if not extras:      # (the argument checks and the device-resident draw below do not depend on the sizes)
        return
    # keep >= N means "no dropout"; keep == 0 (an empty cloud in the reference, "off" to the kernels) is refused
    with pytest.raises(ValueError):
        dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, point_dropout=(0, 1))
    full = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, point_dropout=(N, 1))
    plain = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    assert float((full["proj"] - plain["proj"]).abs().max()) == 0.0
    # {keep, seed} read from device memory at run time (what a recorded HIP graph replays): same draw, bit for bit
    state = torch.tensor([keep, seed & 0x7fffffff], dtype=torch.int32, device=dev)
    by_value = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale,
                                               point_dropout=(keep, seed & 0x7fffffff))
    by_state = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale,
                                               point_dropout=state)
    assert float((by_state["proj"] - by_value["proj"]).abs().max()) == 0.0
    g2 = torch.autograd.grad(by_state["proj"], [pc], w)[0].cpu().numpy()
    assert np.count_nonzero(np.abs(g2).sum(-1)) <= B * keep
    state[0] = N                                             # the kernels follow the tensor, not a copy of it
    again = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, point_dropout=state)
    assert float((again["proj"] - plain["proj"]).abs().max()) == 0.0


def fused_l2_epilogue_equals_the_autograd_loss(dev, B=3, N=300, D=32, K=5):
    """pointcloud_project_fast(l2_target=(gt, w)): 'proj_l2_grad' is w * (proj - gt) exactly, and feeding it back
    as the gradient of proj gives the same input gradients as autograd through 0.5 * w * sum((proj - gt)^2)
    (model_pc.py:414-415: tf.nn.l2_loss(gt - pred) / num_samples, w = 1 / num_samples)."""
    inp = synth.make_inputs(B, N, 77)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, 0.9, device=dev)
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    gt = torch.tensor(np.random.default_rng(3).uniform(0, 1, (B, D, D, 1)).astype(np.float32), device=dev)
    w = 1.0 / B
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, l2_target=(gt, w))
    assert not out["proj_l2_grad"].requires_grad
    assert float((out["proj_l2_grad"] - (out["proj"].detach() - gt) * w).abs().max()) == 0.0
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], out["proj_l2_grad"])
    plain = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    assert float((plain["proj"] - out["proj"]).abs().max()) <= 1e-6      # (the generic path's float atomics reorder)
    loss = 0.5 * w * ((plain["proj"] - gt) ** 2).sum()
    gr = torch.autograd.grad(loss, [pc, pose, scale])
    for a, b in zip(g, gr):
        assert relerr(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    # [B,D,D] target works as well; anything else is refused, and so is the max-projection collapse
    out3 = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale,
                                           l2_target=(gt.reshape(B, D, D), w))
    assert float((out3["proj_l2_grad"] - out["proj_l2_grad"]).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale,
                                        l2_target=(gt[:, : D // 2], w))
    cfg_max = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K, ptn_max_projection=True)
    with pytest.raises(ValueError):
        dpc_amd.pointcloud_project_fast(cfg_max, pc, pose, None, None, kern, scaling_factor=scale, l2_target=(gt, w))


def student_loss_equals_the_quaternion_composite(dev, n=37, C=4, seed=5):
    """ops.StudentLoss (one kernel) == the reference's composite of quaternion_multiply / conjugate / normalise
    (model_pc.py:338-381), value and gradient, with and without valid_samples weights; unnormalised inputs."""
    from dpc_amd.util import quaternion as Q
    rng = np.random.default_rng(seed)
    poses = torch.tensor(rng.standard_normal((n * C, 4)).astype(np.float32), device=dev)
    winners = torch.tensor(rng.integers(0, C, n), device=dev, dtype=torch.int64)
    for weights in (None, torch.tensor(rng.uniform(0, 1, n).astype(np.float32), device=dev)):
        s1 = torch.tensor(rng.standard_normal((n, 4)).astype(np.float32), device=dev, requires_grad=True)
        s2 = s1.detach().clone().requires_grad_(True)
        loss = dpc_amd.ops.StudentLoss.apply(s1, poses, winners, weights, C, 20.0)
        (loss * 0.7).backward()
        teachers = poses.reshape(n, C, 4)[torch.arange(n, device=dev), winners]
        a = Q.quaternion_normalise(Q.quaternion_multiply(teachers, Q.quaternion_conjugate(s2)))[:, 0]
        ref = ((1.0 - a ** 2) * (1.0 if weights is None else weights)).sum() / float(n) * 20.0
        (ref * 0.7).backward()
        assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
        assert relerr(s1.grad.cpu().numpy(), s2.grad.cpu().numpy()) < 2e-6
    with pytest.raises(ValueError):
        dpc_amd.ops.StudentLoss.apply(s1, poses[:-4], winners, None, C, 1.0)
    with pytest.raises(ValueError):
        dpc_amd.ops.StudentLoss.apply(s1, poses, winners.to(torch.int32), None, C, 1.0)


def knife_edge_inputs_match_reference_conventions(dev, D, Dz):
    """NO nudging: points that sit EXACTLY on lattice nodes / cell faces / the faces of the unit cube, a node
    that receives exactly 1.0 (clip_by_value(G0,0,1) must still pass its gradient: closed interval), a node
    that receives 2.0 (gradient blocked), coincident points.  Identity pose and depth -0.125 make the
    perspective divide exact in fp32 (Z = 1.875 = focal length), so the fp32 lattice coordinates are the
    same bits in the product and in the fp32 CPU restatement of the reference graph, and the piecewise
    choices (floor, the closed validity test -0.5 <= p <= 0.5, the closed clip interval) are compared one to one."""
    f = 1.0 / (D - 1)
    fz = 1.0 / (Dz - 1)
    nodes_xy = [-0.5, 0.5] + ([-0.5 + 8 * f, -0.5 + 9 * f] if (D - 1) & (D - 2) == 0 else [])   # exact iff D-1 = 2^k
    zs = [-0.125, -0.5, 0.5] + ([-0.5 + 5 * fz] if (Dz - 1) & (Dz - 2) == 0 else [])
    pts = []
    for y in nodes_xy + [0.1]:
        for x in nodes_xy + [-0.2]:
            pts.append((-0.125, y, x))
    for z in zs:
        pts.append((z, 0.25, -0.25))
    pts += [(-0.125, -0.5, -0.5)] * 1                # second point on the corner node: G0 = 2 there
    pts += [(-0.125, 0.5, -0.5), (-0.125, 0.5000001, 0.0), (0.50000006, 0.0, 0.0)]   # just outside: dropped
    pts += [(-0.125, 0.3, 0.3), (-0.125, 0.3, 0.3)]  # coincident interior points
    pc_np = np.array(pts, dtype=np.float32)[None]
    B, N = 1, pc_np.shape[1]
    pose_np = np.array([[1.0, 0.0, 0.0, 0.0]], dtype=np.float32)
    K = 5
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=(Dz if Dz != D else -1), pc_gauss_kernel_size=K)
    rc = rcpu.Cfg(vox_size=D, vox_size_z=(Dz if Dz != D else -1), pc_gauss_kernel_size=K)
    pc = torch.tensor(pc_np, device=dev, requires_grad=True)
    pose = torch.tensor(pose_np, device=dev, requires_grad=True)
    cpc = torch.tensor(pc_np, requires_grad=True)                      # fp32 on purpose: same knife-edge decisions
    cpose = torch.tensor(pose_np, requires_grad=True)
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg, 0.7, device=dev))
    ref = rcpu.pointcloud_project_fast(rc, cpc, cpose, None, None, rcpu.smoothing_kernel(rc, 0.7, torch.float32))
    assert maxabs(out["tr_pc"].detach().cpu().numpy(), ref["tr_pc"].detach().numpy()) == 0.0   # bit-identical coordinates
    assert maxabs(out["proj"].detach().cpu().numpy(), ref["proj"].detach().numpy()) < TOL_PROJ
    w = np.random.default_rng(D + Dz).standard_normal(tuple(out["proj"].shape)).astype(np.float32)
    g = torch.autograd.grad(out["proj"], [pc, pose], torch.tensor(w, device=dev))
    rg = torch.autograd.grad(ref["proj"], [cpc, cpose], torch.tensor(w))
    gpc, rgpc = g[0].cpu().numpy()[0], rg[0].numpy()[0]
    scale_ = float(np.abs(rgpc).max())
    assert np.abs(gpc - rgpc).max() < 2e-4 * scale_, np.abs(gpc - rgpc).max() / scale_
    # the point just outside the closed cube (v = 0.50000012) is dropped: exactly zero gradient in both; its
    # neighbour ON the face (v = +0.5 exactly) is not
    assert pts[-4] == (-0.125, 0.5000001, 0.0) and np.all(rgpc[-4] == 0.0) and np.all(gpc[-4] == 0.0)
    assert np.abs(rgpc[-5]).max() > 0 and np.abs(gpc[-5]).max() > 0
    # the single point on the far corner node (+0.5, +0.5) is INSIDE (closed test) and its node holds exactly 1.0:
    # its gradient is not blocked by the clip
    k = len(nodes_xy + [0.1]) * 1 + 1                                   # (y = +0.5, x = +0.5)
    assert pts[k] == (-0.125, 0.5, 0.5) and np.abs(rgpc[k]).max() > 0 and np.abs(gpc[k]).max() > 0


def deep_grid_takes_the_generic_path(dev, D=32, Dz=320, K=5, sigma=0.9):
    """vox_size_z > 256: the plane-occupancy words of the fused path cover 256 planes, deeper grids must take
    the generic path (round-1 review: the fused path silently dropped planes >= 256).  Forward and gradients
    against the NumPy oracle."""
    import ctypes
    B, N = 2, 300
    inp = synth.make_inputs(B, N, 777)
    inp = _nudge_off_cell_faces(inp, None, None, Dz, D)
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K)
    taps = onp.smoothing_taps(D, Dz, K, sigma)
    lib = dpc_amd.get_library()
    S = dpc_amd._capi.DpcShape(B, N, Dz, D, K, K, len(taps[2]))
    P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, 0, 0)
    assert lib.dpc_saved_layout(ctypes.byref(S), ctypes.byref(P)) == 1
    t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg, sigma, device=dev),
                                          scaling_factor=scale)
    w = np.random.default_rng(4).standard_normal(tuple(out["proj"].shape))
    g = torch.autograd.grad(out["proj"], [pc, pose, scale], torch.tensor(w, dtype=torch.float32, device=dev))
    f64 = lambda a: a.astype(np.float64)
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, Dz=Dz, D=D)
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, fw, dproj=w)
    assert maxabs(out["proj"].detach().cpu().numpy(), fw["proj"]) < TOL_PROJ
    assert relerr(g[0].cpu().numpy(), bw["dpc"]) < TOL_GRAD
    assert relerr(g[1].cpu().numpy(), bw["dpose"]) < TOL_GRAD
    assert relerr(g[2].cpu().numpy(), bw["dscale"].reshape(g[2].shape)) < TOL_GRAD


def degenerate_clouds_against_numpy_oracle(dev, heavy=True):
    """Clouds that stress the bucket logic of the fused path: a single point; every point outside the cube; every
    point in ONE depth plane (far more than 2 x 256 points in a plane: the re-read loops; with `heavy` more than
    4096: the float-atomic fallback of the splat); all points on one spot (a pile-up far above 1: clipped, gradient
    blocked at that cell)."""
    D, K, sigma = 64, 5, 0.9
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    taps = onp.smoothing_taps(D, -1, K, sigma)
    rng = np.random.default_rng(11)
    quat = np.array([[1.0, 0.0, 0.0, 0.0]], np.float32)
    nsheet = 6000 if heavy else 700
    sheet = np.stack([np.full(nsheet, 0.1003), rng.uniform(-0.3, 0.3, nsheet), rng.uniform(-0.3, 0.3, nsheet)], -1)
    clouds = {
        "single": np.array([[0.05, -0.11, 0.2]]),
        "all_outside": rng.uniform(0.8, 0.9, (50, 3)),
        "sheet": sheet,
        "pile": np.tile(np.array([[0.013, 0.021, -0.034]]), (300, 1)),
    }
    for name, pts in clouds.items():
        pc_np = pts.astype(np.float32)[None]
        inp = {"pc": pc_np, "pose": quat}
        if name in ("sheet", "single"):
            inp = _nudge_off_cell_faces(inp, None, None, D, D)
        pc = torch.tensor(inp["pc"], device=dev, requires_grad=True)
        pose = torch.tensor(quat, device=dev, requires_grad=True)
        out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern)
        w = rng.standard_normal(tuple(out["proj"].shape))
        g = torch.autograd.grad(out["proj"], [pc, pose], torch.tensor(w, dtype=torch.float32, device=dev))
        f64 = lambda a: a.astype(np.float64)
        fw = onp.project_forward(f64(inp["pc"]), f64(quat), None, None, None, taps, Dz=D, D=D)
        bw = onp.project_backward(f64(inp["pc"]), f64(quat), None, None, None, taps, fw, dproj=w)
        assert maxabs(out["proj"].detach().cpu().numpy(), fw["proj"]) < TOL_PROJ, name
        scale_ = max(float(np.abs(bw["dpc"]).max()), 1e-12)
        assert maxabs(g[0].cpu().numpy(), bw["dpc"]) <= TOL_GRAD * scale_ + 1e-9, name
        if name == "all_outside":
            assert float(g[0].abs().max()) == 0.0 and float(g[1].abs().max()) == 0.0
            assert abs(float(out["proj"].max()) - float(out["proj"].min())) == 0.0       # every ray is the empty ray
        if name == "pile":
            assert float(g[0].abs().max()) == 0.0       # G0 = 300 x weight >> 1 at all 8 corners: the clip blocks everything


# ---------------------------------------------------------------------------
# round-3 cases
# ---------------------------------------------------------------------------
def fused_dropout_kept_mask_from_library(dev, pc_np, pose_np, D, K, keep, seed, sigma=0.9):
    """White-box read of the fused dropout's draw: run dpc_project_forward through the C ABI on caller-allocated
    buffers and decode point_index -- a point was kept (and is inside the cube) iff its slot lies below the start
    of the "dropped" bucket.  Returns a [B, N] bool array."""
    import ctypes
    lib = dpc_amd.get_library()
    B, N = pc_np.shape[0], pc_np.shape[1]
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device=dev)
    taps = [k.reshape(-1).contiguous() for k in kern]
    S = dpc_amd._capi.DpcShape(B, N, D, D, K, K, K)
    P = dpc_amd._capi.DpcParams(2.0, 1.875, 1e-5, 10.0, 1, 0, 0, int(keep), int(seed) & 0xffffffff)
    assert lib.dpc_saved_layout(ctypes.byref(S), ctypes.byref(P)) & 6 == 6, "expected the fused path"
    t = lambda a: torch.tensor(a, device=dev)
    pc, pose = t(pc_np), t(pose_np)
    new = lambda *s, **k: torch.empty(*s, device=dev, dtype=k.get("dtype", torch.float32))
    tr_pc, cmask = new(B, N, 3), new(B, N, 4, dtype=torch.uint8)
    pindex = new(lib.dpc_point_index_ints(ctypes.byref(S)), dtype=torch.int32)
    blur, sums, proj, depth = new(B, D, D, D), new(B, D, D, 2, dtype=torch.float64), new(B, D, D), new(B, D, D)
    nws = lib.dpc_workspace_bytes(ctypes.byref(S), 0)
    ws = torch.empty(nws + 256, dtype=torch.uint8, device=dev)
    p = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
    stream = None if lib.host_memory else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.dpc_project_forward(stream, ctypes.byref(S), ctypes.byref(P), p(pc), p(pose), None, None, None,
                                 p(taps[0]), p(taps[1]), p(taps[2]), p(tr_pc), None, p(cmask), p(pindex), p(blur), p(sums),
                                 p(proj), p(depth), ctypes.c_void_p((ws.data_ptr() + 255) & ~255), nws)
    lib.check(rc, "dpc_project_forward")
    pi = pindex.cpu().numpy()
    slot_of = pi[4 * B * N:5 * B * N].reshape(B, N)
    zstart = pi[5 * B * N:5 * B * N + B * (D + 2)].reshape(B, D + 2)
    return slot_of < zstart[:, D:D + 1]


def dropout_statistics(masks, keep, pairs):
    """Inclusion and pair-inclusion statistics of a stack of draws masks [K, N] (bool; row = one (seed, instance)
    key, exactly `keep` true entries): z-scores against sampling without replacement, as chi-squares.
    Returns dict(chi_incl, df_incl, zmax_incl, chi_pair, df_pair, zmax_pair)."""
    Kk, N = masks.shape
    assert (masks.sum(1) == keep).all()
    p = keep / N
    cnt = masks.sum(0).astype(np.float64)
    z = (cnt - Kk * p) / np.sqrt(Kk * p * (1 - p))
    p2 = keep * (keep - 1) / (N * (N - 1))
    both = (masks[:, pairs[:, 0]] & masks[:, pairs[:, 1]]).sum(0).astype(np.float64)
    z2 = (both - Kk * p2) / np.sqrt(Kk * p2 * (1 - p2))
    return dict(chi_incl=float((z ** 2).sum()), df_incl=N - 1, zmax_incl=float(np.abs(z).max()),
                chi_pair=float((z2 ** 2).sum()), df_pair=len(pairs), zmax_pair=float(np.abs(z2).max()))


def dropout_pairs(N, rng, nrandom=4000):
    """index pairs whose joint inclusion is tested: random ones, neighbours (n, n+1), and a stride of 64 (one wave)"""
    r = np.stack([rng.integers(0, N, nrandom), rng.integers(0, N, nrandom)], 1)
    r = r[r[:, 0] != r[:, 1]]
    return np.concatenate([r, np.stack([np.arange(N - 1), np.arange(1, N)], 1),
                           np.stack([np.arange(N - 64), np.arange(64, N)], 1)])


def assert_dropout_statistics(st, sigmas=5.0, zmax=6.0):
    for tag in ("incl", "pair"):
        chi, df = st["chi_" + tag], st["df_" + tag]
        assert abs(chi - df) < sigmas * np.sqrt(2.0 * df), (tag, st)
        assert st["zmax_" + tag] < zmax, (tag, st)


def fused_candidate_loss_equals_the_image_epilogue(dev, B=8, C=4, N=300, D=32, K=5, S=48, with_valid=True, rep=1):
    """pointcloud_project_fast(silhouette_target=...) -- per-instance errors from k_zfwd's partials, arg-min / weights /
    loss in dpc_silhouette_select, the loss gradient formed inside k_zbwd -- against ops.SilhouetteLoss run on the
    projection image (dpc_silhouette_loss_fwd/bwd, itself pinned by the reference goldens caller_loss*.npz): loss,
    winners, per-instance errors and every input gradient."""
    rng = np.random.default_rng(100 + B + D)
    inp = synth.make_inputs(B, N, 8181)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, 0.9, device=dev)
    G = B // C
    gt = torch.tensor((rng.uniform(size=(G, S, S, 1)) > 0.6).astype(np.float32), device=dev)
    valid = torch.tensor(rng.uniform(0.5, 1.0, G).astype(np.float32), device=dev) if with_valid else None
    res = {}
    for fused in (True, False):
        t = lambda a: torch.tensor(a, device=dev, requires_grad=True)
        pc = t(inp["pc"][::rep].copy()) if rep > 1 else t(inp["pc"])
        pose, scale = t(inp["pose"]), t(inp["scale"])
        kw = dict(views_per_cloud=rep) if rep > 1 else {}
        if fused:
            out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale,
                                                  silhouette_target=(gt, C, valid), **kw)
            loss, win, err = out["proj_loss"], out["winning_pose_candidates"], out["proj_inst_err"]
        else:
            out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale, **kw)
            loss, win, err = dpc_amd.ops.SilhouetteLoss.apply(out["proj"], gt, valid, C)
        g = torch.autograd.grad(loss * 3.0, [pc, pose, scale])
        res[fused] = (float(loss), win.cpu().numpy(), err.cpu().numpy(), [x.cpu().numpy() for x in g])
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]) and np.array_equal(a[1], b[1])
    assert maxabs(a[2], b[2]) <= 1e-5 * float(np.abs(b[2]).max())
    for x, y in zip(a[3], b[3]):
        assert maxabs(x, y) <= 2e-5 * max(float(np.abs(y).max()), 1e-12)
