#!/usr/bin/env python3
"""Goldens for the remaining loss terms: the reference's own
ModelPointCloud.get_loss (dpc/models/model_pc.py:425-445 -> add_proj_loss, add_drc_loss,
add_proj_rgb_loss, add_proj_depth_loss of dpc/util/losses.py, imported unchanged under
oracle/tf_shim) on a toy RGB case, with and without Gaussian-filtered GT.  Container only.

    python tests/golden/make_loss_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_caller_goldens as C  # noqa: E402  (sets up the shim + reference import paths)

tf, M = C.tf, C.M


def run(dtype, inp, cfg, gs):
    tf.set_float_dtype(dtype)
    model = M.ModelPointCloud(cfg, global_step=gs)
    leaves = {k: torch.tensor(inp[k], dtype=dtype, requires_grad=True) for k in ("points_1", "rgb_1", "poses", "scaling_factor")}
    all_points = model.replicate_for_multiview(tf.convert_to_tensor(leaves["points_1"]))
    all_rgb = model.replicate_for_multiview(tf.convert_to_tensor(leaves["rgb_1"]))
    all_scal = model.replicate_for_multiview(tf.convert_to_tensor(leaves["scaling_factor"]))
    outputs = {"points_1": tf.convert_to_tensor(leaves["points_1"]), "all_points": all_points, "all_rgb": all_rgb,
               "poses": tf.convert_to_tensor(leaves["poses"]), "all_scaling_factors": all_scal, "all_focal_length": None}
    inputs = {k: tf.convert_to_tensor(torch.tensor(inp[k], dtype=dtype)) for k in ("masks", "images", "depths")}
    outputs = model.compute_projection(inputs, outputs, is_training=False)
    loss = model.get_loss(inputs, outputs, add_summary=False)
    loss.backward()
    res = {"loss": np.asarray(float(loss))}
    for k, v in leaves.items():
        res["d" + k] = v.grad.numpy()
    tf.set_float_dtype(torch.float32)
    return res


def main():
    rng = np.random.default_rng(909)
    Bm, V, N, D = 2, 2, 96, 16
    base = dict(pose_predict_num_candidates=1, step_size=V, batch_size=Bm, pc_rgb=True, drc_weight=0.3,
                proj_rgb_weight=0.7, proj_depth_weight=0.2, max_depth=12.0, max_dataset_depth=10.0,
                pc_gauss_filter_gt_rgb=False, weight_decay=0.0)
    depths = rng.uniform(1.6, 2.4, (Bm * V, D, D, 1)).astype(np.float32)
    depths[rng.uniform(0, 1, depths.shape) > 0.6] = 10.0
    inp = dict(points_1=(0.5 * np.tanh(rng.standard_normal((Bm, N, 3)) * 0.8)).astype(np.float32) * 0.8,
               rgb_1=rng.uniform(0, 1, (Bm, N, 3)).astype(np.float32),
               poses=rng.standard_normal((Bm * V, 4)).astype(np.float32),
               scaling_factor=rng.uniform(0.5, 1.0, (Bm, 1)).astype(np.float32),
               masks=(rng.uniform(0, 1, (Bm * V, 32, 32, 1)) > 0.6).astype(np.float32),
               images=rng.uniform(0, 1, (Bm * V, 32, 32, 3)).astype(np.float32), depths=depths)
    gs = 150000
    out = {}
    for name, extra in (("plain", {}), ("filtered", dict(pc_gauss_filter_gt=True, pc_gauss_filter_gt_rgb=True))):
        cfg = C.make_cfg(**dict(base, **extra))
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            for k, v in run(dt, inp, cfg, gs).items():
                out["%s_%s_%s" % (name, k, tag)] = v
        print(name, "loss", float(out[name + "_loss_f64"]))
    np.savez_compressed(os.path.join(HERE, "caller_losses_rgb.npz"), global_step=gs, Bm=Bm, V=V, D=D, K=5, **inp, **out)


if __name__ == "__main__":
    main()
