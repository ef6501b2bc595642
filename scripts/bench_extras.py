#!/usr/bin/env python3
"""Time the kernels outside the headline path on one GPU: silhouette loss epilogue,
nearest-neighbour distance (Chamfer), exact Gaussian voxeliser.  Prints one JSON object."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpc_amd  # noqa: E402
from dpc_amd import ops  # noqa: E402
from dpc_amd.util.point_cloud_distance import point_cloud_distance  # noqa: E402


def timed(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = "cuda"
    res = {}
    # loss epilogue at the chair_unsupervised shapes: 16 models x 4 views... B = 160 instances, C = 4
    proj = torch.rand(160, 64, 64, 1, device=dev, requires_grad=True)
    gt = (torch.rand(40, 128, 128, 1, device=dev) > 0.5).float()

    def loss_step():
        proj.grad = None
        l, _, _ = ops.SilhouetteLoss.apply(proj, gt, None, 4)
        l.backward()
    res["silhouette_loss_fwd_bwd_ms"] = timed(loss_step, 50, 5)
    # Chamfer: 8000 predicted vs 100k ground-truth points, fp64, both directions
    a = torch.rand(8000, 3, device=dev, dtype=torch.float64)
    b = torch.rand(100000, 3, device=dev, dtype=torch.float64)
    res["nn_8000_to_100k_f64_ms"] = timed(lambda: point_cloud_distance(a, b))
    res["nn_100k_to_8000_f64_ms"] = timed(lambda: point_cloud_distance(b, a))
    res["nn_pairs_per_s"] = 8000 * 100000 / (res["nn_8000_to_100k_f64_ms"] * 1e-3)
    # exact Gaussian voxeliser: 4 views x 8000 points, 64^3 lattice
    pc = (torch.rand(4, 8000, 3, device=dev) - 0.5).requires_grad_(True)
    w = torch.rand(4, 64, 64, 64, device=dev)
    sigma = 3.0 / 64

    def gv_fwd():
        return ops.GaussVoxelize.apply(pc, sigma, 64, (0, 1, 2), 2)

    def gv_step():
        pc.grad = None
        (gv_fwd() * w).sum().backward()
    res["gauss_voxelize_fwd_ms_per_view"] = timed(gv_fwd, 5, 1) / 4
    res["gauss_voxelize_fwd_bwd_ms_per_view"] = timed(gv_step, 5, 1) / 4
    res["gauss_voxelize_fwd_gflops"] = 2 * 8000 * 64 ** 3 / (res["gauss_voxelize_fwd_ms_per_view"] * 1e-3) / 1e9
    # the projector with a depth-image gradient coming in (proj_depth_weight > 0, dpc/util/losses.py:113-136): the collapse VJP
    # then runs its HAS_GD instantiation (gamma_j = g + gd psi_j per plane) -- per-kernel times next to the plain step
    import bench
    lib = dpc_amd.get_library()
    for name, cfg_id, B, sigma in (("cfg2", 2, None, None), ("training_shape", 3, 320, 3.0), ("training_shape_sigma0.8", 3, 320, 0.8)):
        case = bench.build_case(cfg_id, B, torch.device(dev), sigma=sigma)
        gd = torch.randn(case["B"], case["D"], case["D"], 1, device=dev) * 1e-3

        def depth_step():
            out = dpc_amd.pointcloud_project_fast(case["cfg"], case["pc"], case["pose"], None, None, case["kern"],
                                                  scaling_factor=case["scale"], l2_target=(case["gt"], 1.0 / case["B"]))
            return torch.autograd.grad([out["proj"], out["proj_depth"]], [case["pc"], case["pose"], case["scale"]],
                                       [out["proj_l2_grad"], gd])
        for label, fn in (("plain", lambda: bench.step(case)), ("with_depth_gradient", depth_step)):
            ms = timed(fn, 20, 5)
            lib.profile(True)
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            per = {}
            for lab, t in lib.profile_records():
                per[lab] = per.get(lab, 0.0) + t / 5
            lib.profile(False)
            res["projector_%s_%s" % (name, label)] = {"ms_per_step": ms, "kernel_ms": {k: round(v, 4) for k, v in sorted(per.items())}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
