#!/usr/bin/env python3
"""Goldens for the exact Gaussian voxeliser / slow projector: the
reference's own pointcloud2voxels and pointcloud_project (dpc/util/point_cloud.py:17-57,
:219-226, imported unchanged under oracle/tf_shim) on small seeded clouds, in fp32 and
fp64, with gradients of a random linear functional.  Container only.

    python tests/golden/make_slow_goldens.py
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
from util import point_cloud as ref_pc  # noqa: E402  (reference, unchanged)


class Cfg(dict):
    __getattr__ = dict.__getitem__


def make_cfg(**kw):
    c = Cfg(vox_size=16, vox_size_z=-1, camera_distance=2.0, focal_length=1.875, pose_quaternion=True,
            ptn_max_projection=False, drc_logsum=True, drc_logsum_clip_val=1e-5, drc_tf_cumulative=True,
            max_depth=10.0, pc_normalise_gauss=False, pc_normalise_gauss_analytical=True)
    c.update(kw)
    return c


def main():
    rng = np.random.default_rng(404)
    out = {}
    # name: (B, N, G, sigma, normalise flags)
    vox_cases = {"vox_analytical": (2, 150, 16, 0.11, (False, True)), "vox_none": (1, 90, 12, 0.2, (False, False)),
                 "vox_sum": (2, 70, 9, 0.25, (True, False))}
    for name, (B, N, G, sigma, (nsum, nana)) in vox_cases.items():
        cfg = make_cfg(vox_size=G, pc_normalise_gauss=nsum, pc_normalise_gauss_analytical=nana)
        pc = rng.uniform(-0.9, 0.9, (B, N, 3)).astype(np.float32)
        if name == "vox_none":
            pc[:, :30] = pc[:, :1] + 0.01 * rng.standard_normal((B, 30, 3)).astype(np.float32)   # pile-up -> clipped nodes
        w = rng.standard_normal((B, G, G, G, 1)).astype(np.float32)
        out[name + "_meta"] = np.array([B, N, G, int(nsum), int(nana)])
        out[name + "_sigma"] = np.float64(sigma)
        out[name + "_pc"], out[name + "_w"] = pc, w
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            tf.set_float_dtype(dt)
            p = torch.tensor(pc, dtype=dt, requires_grad=True)
            vox = ref_pc.pointcloud2voxels(cfg, tf.convert_to_tensor(p), sigma)
            (vox * torch.tensor(w, dtype=dt)).sum().backward()
            out[name + "_vox_" + tag] = vox.detach().numpy()
            out[name + "_dpc_" + tag] = p.grad.numpy()
        tf.set_float_dtype(torch.float32)
        v = out[name + "_vox_f64"]
        print(name, "max", v.max(), "clipped nodes", int((v >= 1.0).sum()), "of", v.size)
    # slow projector end to end
    B, N, G, sigma = 2, 200, 16, 2.0 / 16
    cfg = make_cfg(vox_size=G)
    pc = (0.4 * np.tanh(rng.standard_normal((B, N, 3)))).astype(np.float32)
    pose = rng.standard_normal((B, 4)).astype(np.float32)
    w = rng.standard_normal((B, G, G, 1)).astype(np.float32)
    out["proj_meta"] = np.array([B, N, G])
    out["proj_sigma"] = np.float64(sigma)
    out["proj_pc"], out["proj_pose"], out["proj_w"] = pc, pose, w
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        tf.set_float_dtype(dt)
        p = torch.tensor(pc, dtype=dt, requires_grad=True)
        q = torch.tensor(pose, dtype=dt, requires_grad=True)
        proj, vox = ref_pc.pointcloud_project(cfg, tf.convert_to_tensor(p), tf.convert_to_tensor(q), sigma)
        (proj * torch.tensor(w, dtype=dt)).sum().backward()
        out["proj_proj_" + tag], out["proj_vox_" + tag] = proj.detach().numpy(), vox.detach().numpy()
        out["proj_dpc_" + tag], out["proj_dpose_" + tag] = p.grad.numpy(), q.grad.numpy()
    tf.set_float_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "slow_path.npz"), names=np.array(list(vox_cases)), **out)
    print("wrote slow_path.npz; proj range", out["proj_proj_f64"].min(), out["proj_proj_f64"].max())


if __name__ == "__main__":
    main()
