"""Caller rows C1/C2: the PyTorch counterpart of
ModelPointCloud.compute_projection / add_proj_loss against goldens produced by
the reference's own model_pc.py (tests/golden/make_caller_goldens.py): instance
ordering (model-major -> view -> candidate), kwargs, output keys, loss value,
gradients, and the sigma / dropout schedules."""
import numpy as np
import pytest
import torch

import dpc_amd
from dpc_amd import model_pc as M
from helpers import load, maxabs, relerr


def _cfg(g):
    return dpc_amd.default_config(vox_size=int(g["D"]), pc_gauss_kernel_size=int(g["K"]), pc_relative_sigma=3.0,
                                  pc_relative_sigma_end=0.2, predict_pose=True, pose_predict_num_candidates=int(g["C"]),
                                  step_size=int(g["V"]), batch_size=int(g["Bm"]), pose_predictor_student=False)


def _run(dev):
    g = load("caller_toy")
    cfg = _cfg(g)
    model = M.ModelPointCloud(cfg, global_step=int(g["global_step"]), device=dev)
    t = lambda k, grad=True: torch.tensor(g[k], device=dev, requires_grad=grad)
    pts, poses, scal = t("points_1"), t("poses"), t("scaling_factor")
    outputs = {"points_1": pts, "poses": poses, "scaling_factor": scal, "focal_length": None}
    outputs = model.replicate_outputs(outputs)
    inputs = {"masks": t("masks", False)}
    outputs = model.compute_projection(inputs, outputs, is_training=False)
    loss = model.add_proj_loss(inputs, outputs, cfg.proj_weight)
    loss.backward()
    return g, model, outputs, loss, pts, poses, scal


def _check(g, model, outputs, loss, pts, poses, scal):
    assert abs(model._sigma_rel - float(g["sigma_rel_f64"])) < 1e-6
    assert maxabs(outputs["all_points"].detach().cpu().numpy(), g["all_points_f32"]) == 0.0   # ordering
    assert maxabs(outputs["projs"].detach().cpu().numpy(), g["projs_f64"]) < 2e-5
    assert maxabs(outputs["projs_depth"].detach().cpu().numpy(), g["projs_depth_f64"]) < 2e-4
    assert maxabs(outputs["projs_1"].detach().cpu().numpy(), g["projs_1_f64"]) < 2e-5
    assert outputs["projs_rgb"] is None and outputs["drc_probs"] is None
    assert abs(float(loss) - float(g["loss_f64"])) < 1e-4 * float(g["loss_f64"])
    assert relerr(pts.grad.cpu().numpy(), g["dpoints_f64"]) < 2e-4
    assert relerr(poses.grad.cpu().numpy(), g["dposes_f64"]) < 2e-4
    assert relerr(scal.grad.cpu().numpy(), g["dscaling_f64"]) < 2e-4


def test_caller_toy_emulated(emu):
    _check(*_run("cpu"))


@pytest.mark.gpu
def test_caller_toy_gpu():
    dpc_amd._capi.set_library(None)
    _check(*_run("cuda"))


def test_schedules_match_reference():
    g = load("caller_toy")
    cfg = _cfg(g)
    for s, ref in zip(g["sched_steps"], g["sched_sigma"]):
        assert abs(M.get_smooth_sigma(cfg, int(s)) - ref) < 1e-6
    cfg_d = dpc_amd.default_config(pc_point_dropout=0.07)
    for s, ref in zip(g["sched_steps"], g["sched_drop_lin"]):
        assert abs(M.get_dropout_prob(cfg_d, int(s)) - ref) < 1e-6
    cfg_e = dpc_amd.default_config(pc_point_dropout=0.07, pc_point_dropout_exponential_schedule=True)
    for s, ref in zip(g["sched_steps"], g["sched_drop_exp"]):
        assert abs(M.get_dropout_prob(cfg_e, int(s)) - ref) < 1e-6


def test_tf_repeat_0_and_resize():
    x = torch.arange(6.0).reshape(3, 2)
    assert torch.equal(M.tf_repeat_0(x, 2), x[[0, 0, 1, 1, 2, 2]])
    img = torch.arange(64.0).reshape(1, 8, 8, 1)
    half = M.resize_images_bilinear_tf1(img, [4, 4])
    assert torch.equal(half, img[:, ::2, ::2])                  # exact 2x: top-left sample (TF1 legacy)
    third = M.resize_images_bilinear_tf1(img, [3, 3])
    assert abs(float(third[0, 1, 1, 0]) - (2.0 + 2.0 / 3) * 9) < 1e-4   # src = 8/3 on both axes -> 9*src


def test_pc_point_dropout_semantics():
    """point_cloud.py:293-319: int(N*keep) distinct points per instance, drawn per instance."""
    g = torch.Generator().manual_seed(0)
    pts = torch.arange(2 * 50 * 3, dtype=torch.float32).reshape(2, 50, 3).requires_grad_(True)
    out, rgb = dpc_amd.pc_point_dropout(pts, None, 0.07 * 4, generator=g)
    assert rgb is None and out.shape == (2, 14, 3)
    for b in range(2):
        rows = {tuple(r.tolist()) for r in out[b].detach()}
        assert len(rows) == 14                                         # without replacement
        assert rows <= {tuple(r.tolist()) for r in pts[b].detach()}    # a subset of the instance's own points
    assert not torch.equal(out[0].detach() - pts[0, 0, 0].detach(), out[1].detach() - pts[1, 0, 0].detach())
    out.sum().backward()
    assert int((pts.grad.abs().sum(-1) > 0).sum()) == 28              # gradient reaches exactly the kept points
    full, _ = dpc_amd.pc_point_dropout(pts.detach(), None, 1.0)
    assert full.shape == (2, 50, 3)


def test_dropout_in_compute_projection(emu):
    g = load("caller_toy")
    cfg = _cfg(g)
    cfg.pc_point_dropout = 0.5
    model = M.ModelPointCloud(cfg, global_step=0, device="cpu")
    outputs = {"points_1": torch.tensor(g["points_1"]), "poses": torch.tensor(g["poses"]),
               "scaling_factor": torch.tensor(g["scaling_factor"]), "focal_length": None}
    outputs = model.compute_projection({}, model.replicate_outputs(outputs), is_training=True)
    assert outputs["projs"].shape == (8, 16, 16, 1) and torch.isfinite(outputs["projs"]).all()
    assert outputs["proj_out"]["tr_pc"].shape[1] == int(96 * 0.5)


def test_student_loss_matches_reference(emu):
    """add_student_loss (model_pc.py:338-381) on top of the toy case, weight 20."""
    g = load("caller_toy")
    cfg = _cfg(g)
    cfg.pose_predictor_student = True
    cfg.pose_predictor_student_loss_weight = 20.0
    model = M.ModelPointCloud(cfg, global_step=int(g["global_step"]), device="cpu")
    t = lambda k, grad=True: torch.tensor(g[k], requires_grad=grad)
    pts, poses, scal, stud = t("points_1"), t("poses"), t("scaling_factor"), t("pose_student")
    outputs = model.replicate_outputs({"points_1": pts, "poses": poses, "scaling_factor": scal,
                                       "focal_length": None, "pose_student": stud})
    inputs = {"masks": t("masks", False)}
    outputs = model.compute_projection(inputs, outputs, is_training=False)
    loss = model.add_proj_loss(inputs, outputs, cfg.proj_weight)
    loss.backward()
    assert abs(float(loss) - float(g["student_loss_f64"])) < 1e-4 * float(g["student_loss_f64"])
    assert relerr(stud.grad.numpy(), g["student_dstudent_f64"]) < 1e-4
    assert relerr(poses.grad.numpy(), g["student_dposes_f64"]) < 2e-4


def test_compute_projection_slow_path(emu):
    """cfg.pc_fast:false branch of compute_projection (model_pc.py:250-253): exact Gaussian
    splat, no depth / RGB outputs; the loss still back-propagates to points and poses."""
    g = load("caller_toy")
    cfg = _cfg(g)
    cfg.pc_fast = False
    model = M.ModelPointCloud(cfg, global_step=int(g["global_step"]), device="cpu")
    pts = torch.tensor(g["points_1"], requires_grad=True)
    poses = torch.tensor(g["poses"], requires_grad=True)
    outputs = model.replicate_outputs({"points_1": pts, "poses": poses, "scaling_factor": torch.tensor(g["scaling_factor"]),
                                       "focal_length": None})
    inputs = {"masks": torch.tensor(g["masks"])}
    outputs = model.compute_projection(inputs, outputs, is_training=False)
    assert outputs["projs"].shape == (8, 16, 16, 1) and outputs["projs_depth"] is None and outputs["projs_rgb"] is None
    model.add_proj_loss(inputs, outputs, 1.0).backward()
    assert torch.isfinite(pts.grad).all() and float(pts.grad.abs().sum()) > 0 and float(poses.grad.abs().sum()) > 0


def _run_losses(dev, name):
    """get_loss (model_pc.py:425-445): silhouette + DRC + RGB + depth terms on the toy RGB case."""
    g = load("caller_losses_rgb")
    extra = dict(pc_gauss_filter_gt=True, pc_gauss_filter_gt_rgb=True) if name == "filtered" else {}
    cfg = dpc_amd.default_config(vox_size=int(g["D"]), pc_gauss_kernel_size=int(g["K"]), pc_relative_sigma=3.0,
                                 pc_relative_sigma_end=0.2, predict_pose=True, pose_predict_num_candidates=1,
                                 step_size=int(g["V"]), batch_size=int(g["Bm"]), pose_predictor_student=False,
                                 pc_rgb=True, drc_weight=0.3, proj_rgb_weight=0.7, proj_depth_weight=0.2,
                                 max_depth=12.0, max_dataset_depth=10.0, **extra)
    model = M.ModelPointCloud(cfg, global_step=int(g["global_step"]), device=dev)
    leaves = {k: torch.tensor(g[k], device=dev, requires_grad=True) for k in ("points_1", "rgb_1", "poses", "scaling_factor")}
    outputs = model.replicate_outputs({"points_1": leaves["points_1"], "rgb_1": leaves["rgb_1"], "poses": leaves["poses"],
                                       "scaling_factor": leaves["scaling_factor"], "focal_length": None})
    inputs = {k: torch.tensor(g[k], device=dev) for k in ("masks", "images", "depths")}
    outputs = model.compute_projection(inputs, outputs, is_training=False)
    loss = model.get_loss(inputs, outputs)
    loss.backward()
    ref = float(g[name + "_loss_f64"])
    assert abs(float(loss) - ref) < 2e-5 * ref, (float(loss), ref)
    for k, v in leaves.items():
        assert relerr(v.grad.cpu().numpy(), g["%s_d%s_f64" % (name, k)]) < 2e-4, k


@pytest.mark.parametrize("name", ["plain", "filtered"])
def test_get_loss_all_terms_emulated(emu, name):
    _run_losses("cpu", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["plain", "filtered"])
def test_get_loss_all_terms_gpu(name):
    dpc_amd._capi.set_library(None)
    _run_losses("cuda", name)


def test_student_align_loss_matches_reference(emu):
    """pose_student_align_loss (model_pc.py:362-368): teacher and student rotate a fixed reference
    cloud; the cloud the reference drew is part of the golden."""
    g = load("caller_toy")
    cfg = _cfg(g)
    cfg.pose_predictor_student = True
    cfg.pose_predictor_student_loss_weight = 20.0
    cfg.pose_student_align_loss = True
    model = M.ModelPointCloud(cfg, global_step=int(g["global_step"]), device="cpu")
    assert model._pc_for_alignloss.shape == (2000, 3) and float(model._pc_for_alignloss.abs().max()) <= 3.0
    model._pc_for_alignloss = torch.tensor(g["align_ref_cloud"])
    t = lambda k, grad=True: torch.tensor(g[k], requires_grad=grad)
    pts, poses, scal, stud = t("points_1"), t("poses"), t("scaling_factor"), t("pose_student")
    outputs = model.replicate_outputs({"points_1": pts, "poses": poses, "scaling_factor": scal,
                                       "focal_length": None, "pose_student": stud})
    inputs = {"masks": t("masks", False)}
    outputs = model.compute_projection(inputs, outputs, is_training=False)
    loss = model.add_proj_loss(inputs, outputs, cfg.proj_weight)
    loss.backward()
    assert abs(float(loss) - float(g["align_loss_f64"])) < 1e-4 * float(g["align_loss_f64"])
    assert relerr(stud.grad.numpy(), g["align_dstudent_f64"]) < 1e-4
    assert relerr(poses.grad.numpy(), g["align_dposes_f64"]) < 2e-4


def test_gt_filter_taps_from_the_projectors_x_filter():
    """model_pc.py:398-404 under graph replay: the GT blur takes its taps from the projector's x filter (the same
    gauss_kernel_1d(K, sigma), gauss_kernel.py:27-32) instead of a host sigma -- same bits; and recording_key() carries what a
    recorded step froze: the tap counts and whether the GT filter is still on (pc_gauss_filter_gt_switch_off: off below sigma 1)."""
    import dpc_amd
    from dpc_amd.util.gauss_kernel import gauss_smoothen_image
    cfg = dpc_amd.default_config(vox_size=32, pc_gauss_kernel_size=11, pc_relative_sigma=2.0, pc_relative_sigma_end=0.5,
                                 max_number_of_steps=10, pc_gauss_filter_gt=True, pc_gauss_filter_gt_switch_off=True)
    img = torch.rand(3, 32, 32, 1)
    for sigma in (2.0, 1.3, 0.7):
        a = gauss_smoothen_image(cfg, img, sigma)
        b = gauss_smoothen_image(cfg, img, None, kernel=dpc_amd.smoothing_kernel(cfg, sigma, device="cpu")[0])
        assert torch.equal(a, b)
    m = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
    keys = []
    for gs in range(0, 11):
        m.set_global_step(gs)
        keys.append(m.recording_key())
        assert keys[-1][1] == (dpc_amd.model_pc.get_smooth_sigma(cfg, gs) >= 1.0)
    assert keys[0][1] is True and keys[-1][1] is False
    cfg2 = dpc_amd.default_config(vox_size=32, pc_gauss_kernel_size=11, pc_gauss_filter_gt=True)
    m2 = dpc_amd.model_pc.ModelPointCloud(cfg2, global_step=0, device="cpu")
    assert m2.recording_key() == (m2.effective_tap_counts(), True)


@pytest.mark.gpu
def test_gt_filter_under_graph_replay_gpu():
    """cfg.pc_gauss_filter_gt (+ switch_off) with the step replayed as a HIP graph: the loss and the gradient of the recorded
    step equal the eager ones at every step of a sigma schedule that crosses 1.0 (round 4 raised NotImplementedError here)."""
    import dpc_amd
    dev = torch.device("cuda")
    dpc_amd._capi.set_library(None)
    kw = dict(vox_size=32, pc_gauss_kernel_size=11, pc_relative_sigma=2.0, pc_relative_sigma_end=0.5, max_number_of_steps=12,
              pc_gauss_filter_gt=True, pc_gauss_filter_gt_switch_off=True, pose_predict_num_candidates=1, predict_pose=True,
              step_size=1, batch_size=4, pc_num_points=500, pc_point_dropout=1.0, pc_point_dropout_scheduled=False,
              pose_predictor_student=False)
    cfg = dpc_amd.default_config(**kw)
    rng = np.random.default_rng(3)
    pts = torch.tensor((0.25 * rng.standard_normal((4, 500, 3))).clip(-0.45, 0.45).astype(np.float32), device=dev, requires_grad=True)
    poses = torch.tensor(rng.standard_normal((4, 4)).astype(np.float32), device=dev)
    masks = torch.tensor((rng.random((4, 64, 64, 1)) > 0.5).astype(np.float32), device=dev)
    inputs = {"masks": masks}

    def make():
        return dpc_amd.model_pc.ModelPointCloud(dpc_amd.default_config(**kw), global_step=0, device=dev)

    def step_of(m):
        def run():
            out = {"points_1": pts, "all_points": pts, "poses": poses, "all_scaling_factors": None, "scaling_factor": None,
                   "all_rgb": None, "all_focal_length": None, "predicted_translation": None}
            out = m.compute_projection(inputs, out, is_training=True)
            loss = m.add_proj_loss(inputs, out, 1.0)
            g, = torch.autograd.grad(loss, [pts])
            return loss.detach(), g
        return run
    eager = make()
    rec = make()
    rec.enable_graph_replay(follow_tap_counts=True)
    step = dpc_amd.graphs.RecordedStep(step_of(rec), world=1, device=dev, key=rec.recording_key)
    flips = set()
    for gs in range(0, 13):
        eager.set_global_step(gs)
        rec.set_global_step(gs)
        l0, g0 = step_of(eager)()
        l1, g1 = step()
        flips.add(rec.recording_key()[1])
        assert abs(float(l0) - float(l1)) <= 1e-6 * abs(float(l0)), (gs, float(l0), float(l1))
        assert float((g0 - g1).abs().max()) <= 1e-6 * float(g0.abs().max()), gs
    assert flips == {True, False} and step.records >= 2
