#!/usr/bin/env python3
"""Goldens for the nearest-neighbour / Chamfer kernel: the
reference's own dpc/util/point_cloud_distance.py (imported unchanged under
oracle/tf_shim) on small seeded clouds in float64 and float32, including exact
duplicates (ties -> first index) and a single-target case.  Container only.

    python tests/golden/make_nn_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "tf_shim"))
sys.path.insert(0, "/root/reference/dpc")

import tensorflow as tf  # noqa: E402,F401  (the shim)
from util.point_cloud_distance import point_cloud_distance  # noqa: E402  (reference, unchanged)


def main():
    rng = np.random.default_rng(31)
    out = {}
    cases = {"rand": (700, 1500), "ties": (300, 257), "one_target": (65, 1), "small_src": (3, 2100)}
    for name, (ns, nt) in cases.items():
        vs = rng.uniform(-0.5, 0.5, (ns, 3))
        vt = rng.uniform(-0.5, 0.5, (nt, 3))
        if name == "ties":
            vt[100:200] = vt[0:100]                      # duplicated targets: the first copy must win
            vs[:50] = vt[20:70]                          # zero distances
            vt = np.round(vt * 8) / 8                    # coarse lattice: many exactly equal distances
            vs = np.round(vs * 16) / 16
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            a, b = torch.tensor(vs, dtype=dt), torch.tensor(vt, dtype=dt)
            proj, dist, idx = point_cloud_distance(a, b)
            out["%s_vs_%s" % (name, tag)], out["%s_vt_%s" % (name, tag)] = a.numpy(), b.numpy()
            out["%s_proj_%s" % (name, tag)], out["%s_dist_%s" % (name, tag)] = proj.numpy(), dist.numpy()
            out["%s_idx_%s" % (name, tag)] = idx.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "nn_distance.npz"), names=np.array(list(cases)), **out)
    print("wrote nn_distance.npz", {k: float(out[k + "_dist_f64"].mean()) for k in cases})


if __name__ == "__main__":
    main()
