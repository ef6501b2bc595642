#!/usr/bin/env python3
"""Goldens for the DRC switches of dpc/util/drc.py:47-102 other than the default
(log-space, tf.cumsum): drc_logsum=false (plain products, no clip, unity 1) and
drc_tf_cumulative=false (python loop instead of tf.cumsum), through the reference's own
pointcloud_project_fast under oracle/tf_shim.  Container only.

    python tests/golden/make_drc_variant_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as G  # noqa: E402  (sets up the shim + reference import paths)


def main():
    inp = G.tiny_inputs(seed=31)
    up = G.rand_upstream(32, 2, 16, 16, probs=True)
    cfg_nolog = G.make_cfg(vox_size=16, pc_gauss_kernel_size=5, drc_logsum=False)
    G.save("tiny_nolog", sigma=0.8, K=5, D=16, Dz=16, **inp, **up, **G.both(cfg_nolog, inp, 0.8, up))
    cfg_loop = G.make_cfg(vox_size=16, pc_gauss_kernel_size=5, drc_tf_cumulative=False)
    G.save("tiny_loop", sigma=0.8, K=5, D=16, Dz=16, **inp, **up, **G.both(cfg_loop, inp, 0.8, up))
    # the fused kernels (D = 32) with the non-log collapse
    cfg32 = G.make_cfg(vox_size=32, pc_gauss_kernel_size=5, drc_logsum=False)
    inp32 = G.synth.make_inputs(2, 300, 77)
    up32 = G.rand_upstream(33, 2, 32, 32)
    g32 = {k: v for k, v in G.both(cfg32, inp32, 1.0, up32).items() if not k.startswith(("voxels", "drc_probs"))}
    G.save("d32_nolog", sigma=1.0, K=5, D=32, Dz=32, **inp32, **up32, **g32)


if __name__ == "__main__":
    main()
