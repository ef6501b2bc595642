"""Drive the product Python API (dpc_amd) on a golden case; shared by the
emulation tests (CPU tensors + emulation library) and the GPU tests."""
import numpy as np
import torch

import dpc_amd
from helpers import case_params


def product_cfg(name, g):
    cp = case_params(name, g)
    extra = {}
    if name == "tiny_rgb_div":
        extra = dict(pc_rgb_divide_by_occupancies=True, pc_rgb_clip_after_conv=True, pc_rgb_stop_points_gradient=True)
    if name.endswith("_nolog"):
        extra = dict(drc_logsum=False)
    if name == "tiny_loop":
        extra = dict(drc_tf_cumulative=False)
    return dpc_amd.default_config(**extra, vox_size=cp["D"], vox_size_z=(cp["Dz"] if cp["Dz"] != cp["D"] else -1),
                                  pc_gauss_kernel_size=(cp["K"] or 11),
                                  pose_quaternion=cp["pose_quaternion"],
                                  ptn_max_projection=cp["max_projection"])


def run_product(name, g, device, grads=True, touch_lazy=False):
    """Returns (outputs dict of numpy, grads dict of numpy)."""
    cp = case_params(name, g)
    cfg = product_cfg(name, g)
    leaves = {}
    for k in ("pc", "pose", "trans", "scale", "focal", "rgb"):
        if k in g:
            leaves[k] = torch.tensor(g[k], dtype=torch.float32, device=device, requires_grad=grads)
    kern = dpc_amd.smoothing_kernel(cfg, cp["sigma"], device=device) if cp["K"] is not None else None
    out = dpc_amd.pointcloud_project_fast(cfg, leaves["pc"], leaves["pose"], leaves.get("trans"), leaves.get("rgb"), kern,
                                          scaling_factor=leaves.get("scale"), focal_length=leaves.get("focal"))
    res = {"proj": out["proj"], "tr_pc": out["tr_pc"], "proj_depth": out["proj_depth"],
           "proj_rgb": out["proj_rgb"], "voxels_rgb": out["voxels_rgb"]}
    need_probs = "w_probs" in g
    if touch_lazy or need_probs:
        res["voxels"] = out["voxels"]
        res["drc_probs"] = out["drc_probs"]
    gr = {}
    if grads:
        loss = 0.0
        for wname, key in (("w_proj", "proj"), ("w_depth", "proj_depth"), ("w_probs", "drc_probs"),
                           ("w_projrgb", "proj_rgb")):
            if wname in g:
                loss = loss + (torch.tensor(g[wname], dtype=torch.float32, device=device) * res[key]).sum()
        loss.backward()
        gr = {"d" + k: t.grad.detach().cpu().numpy() for k, t in leaves.items() if t.grad is not None}
    res = {k: (None if v is None else v.detach().cpu().numpy()) for k, v in res.items()}
    return res, gr
