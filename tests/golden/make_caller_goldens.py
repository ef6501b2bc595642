#!/usr/bin/env python3
"""Goldens for the caller rows C1/C2 of SURVEY.md 8(a): the reference's own
ModelPointCloud (dpc/models/model_pc.py, imported unchanged under
oracle/tf_shim) runs replicate_for_multiview / tf_repeat_0 ->
compute_projection -> add_proj_loss on a toy case (2 models x 2 views x 2 pose
candidates), plus the sigma / dropout schedules.  Container only.

    python tests/golden/make_caller_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "tf_shim"))
sys.path.insert(0, "/root/reference/dpc")

import tensorflow as tf  # noqa: E402  (the shim)
from models import model_pc as M  # noqa: E402  (reference, unchanged)


class Cfg(dict):
    __getattr__ = dict.__getitem__


def make_cfg(**kw):
    c = Cfg(vox_size=16, vox_size_z=-1, camera_distance=2.0, focal_length=1.875, pose_quaternion=True,
            pc_gauss_kernel_size=5, pc_separable_gauss_filter=True, ptn_max_projection=False, drc_logsum=True,
            drc_logsum_clip_val=1e-5, drc_tf_cumulative=True, max_depth=10.0, pc_rgb=False,
            pc_rgb_stop_points_gradient=False, pc_rgb_clip_after_conv=False, pc_rgb_divide_by_occupancies=False,
            pc_rgb_divide_by_occupancies_epsilon=0.01, pc_relative_sigma=3.0, pc_relative_sigma_end=0.2,
            max_number_of_steps=600000, pc_fast=True, predict_pose=True, predict_translation=False,
            pc_point_dropout=1.0, pc_point_dropout_scheduled=True, pc_point_dropout_exponential_schedule=False,
            pc_point_dropout_start_step=0.0, pc_point_dropout_end_step=1.0, pc_learn_occupancy_scaling=True,
            pose_predict_num_candidates=2, step_size=2, batch_size=2, pose_predictor_student=False,
            pose_student_align_loss=False, align_to_canonical=False, variable_num_views=False,
            bicubic_gt_downsampling=False, pc_gauss_filter_gt=False, pc_gauss_filter_gt_switch_off=False,
            proj_weight=1.0, drc_weight=0.0, proj_depth_weight=0.0)
    c.update(kw)
    return c


def run(dtype, inp, cfg, global_step):
    tf.set_float_dtype(dtype)
    model = M.ModelPointCloud(cfg, global_step=global_step)
    pts = torch.tensor(inp["points_1"], dtype=dtype, requires_grad=True)
    poses = torch.tensor(inp["poses"], dtype=dtype, requires_grad=True)
    scal = torch.tensor(inp["scaling_factor"], dtype=dtype, requires_grad=True)
    masks = torch.tensor(inp["masks"], dtype=dtype)
    C = cfg.pose_predict_num_candidates
    # the reference's own replication code (model_pc.py:261-299)
    all_points = model.replicate_for_multiview(tf.convert_to_tensor(pts))
    all_points = M.tf_repeat_0(all_points, C)
    all_scal = M.tf_repeat_0(model.replicate_for_multiview(tf.convert_to_tensor(scal)), C)
    outputs = {"points_1": tf.convert_to_tensor(pts), "all_points": all_points, "all_rgb": None,
               "poses": tf.convert_to_tensor(poses), "all_scaling_factors": all_scal, "all_focal_length": None}
    student = None
    if cfg.pose_predictor_student:
        student = torch.tensor(inp["pose_student"], dtype=dtype, requires_grad=True)
        outputs["pose_student"] = tf.convert_to_tensor(student)
    inputs = {"masks": tf.convert_to_tensor(masks)}
    outputs = model.compute_projection(inputs, outputs, is_training=False)
    loss = model.add_proj_loss(inputs, outputs, cfg.proj_weight, add_summary=False)
    loss.backward()
    res = dict(sigma_rel=np.asarray(float(model._sigma_rel)), projs=outputs["projs"].detach().numpy(),
               projs_depth=outputs["projs_depth"].detach().numpy(), projs_1=outputs["projs_1"].detach().numpy(),
               all_points=all_points.detach().numpy(), loss=np.asarray(float(loss)),
               dpoints=pts.grad.numpy(), dposes=poses.grad.numpy(), dscaling=scal.grad.numpy())
    if student is not None:
        res["dstudent"] = student.grad.numpy()
    tf.set_float_dtype(torch.float32)
    return res


def loss_cases():
    """add_proj_loss alone (model_pc.py:383-423) on random predictions: candidate counts
    1/2/3, non-integer GT resize ratios, per-group valid_samples weights.  -> caller_loss.npz"""
    rng = np.random.default_rng(77)
    out = {}
    # (the bicubic cases come last: the earlier cases keep their draws; meta[5] = cfg.bicubic_gt_downsampling)
    cases = [("c1_same", 6, 1, 16, 16, False), ("c1_resize", 4, 1, 16, 24, False), ("c2_x2", 8, 2, 16, 32, False),
             ("c3_ratio", 12, 3, 12, 31, True), ("c4_valid", 16, 4, 8, 8, True),
             ("c1_bicubic", 4, 1, 16, 24, False, True), ("c2_bicubic_x2", 8, 2, 16, 32, False, True)]
    for name, B, C, D, S, var, *bic in cases:
        bic = bool(bic and bic[0])
        G = B // C
        pred = rng.uniform(0, 1, (B, D, D, 1)).astype(np.float32)
        gt = (rng.uniform(0, 1, (G, S, S, 1)) > 0.5).astype(np.float32)
        valid = rng.integers(0, 2, (G,)).astype(np.float32) if var else np.ones((G,), np.float32)
        if var:
            valid[0] = 1.0
        cfg = make_cfg(pose_predict_num_candidates=C, variable_num_views=var, vox_size=D, bicubic_gt_downsampling=bic)
        out[name + "_meta"] = np.array([B, C, D, S, int(var)] + ([1] if bic else []))
        out[name + "_pred"], out[name + "_gt"], out[name + "_valid"] = pred, gt, valid
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            tf.set_float_dtype(dt)
            model = M.ModelPointCloud(cfg, global_step=0)
            p = torch.tensor(pred, dtype=dt, requires_grad=True)
            inputs = {"masks": tf.convert_to_tensor(torch.tensor(gt, dtype=dt)),
                      "valid_samples": tf.convert_to_tensor(torch.tensor(valid, dtype=dt))}
            outputs = {"projs": tf.convert_to_tensor(p)}
            if C > 1:
                g_in = inputs["masks"]
                if S > D:
                    g_in = tf.image.resize_images(g_in, [D, D], tf.image.ResizeMethod.BICUBIC if bic
                                                  else tf.image.ResizeMethod.BILINEAR)
                _, win = model.proj_loss_pose_candidates(g_in, outputs["projs"], inputs)
                out[name + "_winners"] = np.asarray(win.detach().numpy(), dtype=np.int64)
            loss = model.add_proj_loss(inputs, outputs, 1.0, add_summary=False)
            loss.backward()
            out[name + "_loss_" + tag] = np.asarray(float(loss))
            out[name + "_dpred_" + tag] = p.grad.numpy()
        tf.set_float_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "caller_loss.npz"), names=np.array([c[0] for c in cases]), **out)
    print("wrote caller_loss.npz", {c[0]: float(out[c[0] + "_loss_f64"]) for c in cases})


def main():
    loss_cases()
    rng = np.random.default_rng(2024)
    Bm, V, C, N, D = 2, 2, 2, 96, 16
    cfg = make_cfg()
    inp = dict(points_1=(0.5 * np.tanh(rng.standard_normal((Bm, N, 3)) * 0.8)).astype(np.float32) * 0.8,
               poses=rng.standard_normal((Bm * V * C, 4)).astype(np.float32),
               scaling_factor=rng.uniform(0.5, 1.0, (Bm, 1)).astype(np.float32),
               masks=(rng.uniform(0, 1, (Bm * V, 32, 32, 1)) > 0.6).astype(np.float32))
    inp["pose_student"] = rng.standard_normal((Bm * V, 4)).astype(np.float32)
    gs = 150000
    out = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for k, v in run(dt, inp, cfg, gs).items():
            out[k + "_" + tag] = v
    # same case with the pose-student loss switched on (model_pc.py:338-381, weight 20)
    cfg_s = make_cfg(pose_predictor_student=True, pose_predictor_student_loss_weight=20.0)
    for k, v in run(torch.float64, inp, cfg_s, gs).items():
        if k in ("loss", "dposes", "dstudent", "dpoints"):
            out["student_" + k + "_f64"] = v
    # ... and with the alignment form of the student loss (model_pc.py:362-368); the reference draws its
    # 2000-point reference cloud with the global numpy RNG in setup_misc, so seed it and store the cloud
    cfg_a = make_cfg(pose_predictor_student=True, pose_predictor_student_loss_weight=20.0, pose_student_align_loss=True)
    np.random.seed(1234)
    tf.set_float_dtype(torch.float64)
    ref_cloud = M.ModelPointCloud(cfg_a, global_step=gs)._pc_for_alignloss.detach().numpy().astype(np.float32)
    np.random.seed(1234)
    for k, v in run(torch.float64, inp, cfg_a, gs).items():
        if k in ("loss", "dposes", "dstudent"):
            out["align_" + k + "_f64"] = v
    out["align_ref_cloud"] = ref_cloud
    # schedules (model_pc.py:35-64)
    steps = np.array([0, 1000, 150000, 300000, 599999], dtype=np.int64)
    sig = [float(M.get_smooth_sigma(cfg, int(s))) for s in steps]
    cfg_d = make_cfg(pc_point_dropout=0.07)
    drop_lin = [float(M.get_dropout_prob(cfg_d, int(s))) for s in steps]
    cfg_e = make_cfg(pc_point_dropout=0.07, pc_point_dropout_exponential_schedule=True)
    drop_exp = [float(M.get_dropout_prob(cfg_e, int(s))) for s in steps]
    np.savez_compressed(os.path.join(HERE, "caller_toy.npz"), global_step=gs, Bm=Bm, V=V, C=C, D=D, K=5, **inp, **out,
                        sched_steps=steps, sched_sigma=np.array(sig), sched_drop_lin=np.array(drop_lin),
                        sched_drop_exp=np.array(drop_exp))
    print("wrote caller_toy.npz; loss f32 %.6f f64 %.6f sigma %.4f" % (out["loss_f32"], out["loss_f64"], out["sigma_rel_f64"]))
    print("sigma", sig, "drop", drop_lin, drop_exp)


if __name__ == "__main__":
    main()
