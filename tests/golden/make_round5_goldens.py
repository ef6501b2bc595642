#!/usr/bin/env python3
"""Round-5 fixtures, generated like the others by running the REFERENCE'S OWN SOURCE under oracle/tf_shim (see
make_goldens.py, whose helpers this script imports; container only -- needs /root/reference):

  k27           pc_gauss_kernel_size = 27 (sigma 4.5) at 32^3: one of the tap counts beyond 21 that got compiled kernels in
                round 5 (dpc/resources/default_config.yaml:55 accepts any size)
  voxz_onetap   vox_size 24, vox_size_z 8, K = 3: gauss_kernel.py:35-54 makes the z filter round(3 * 8 / 24) = 1 tap long

    python tests/golden/make_round5_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402


def main():
    cfg27 = mg.make_cfg(vox_size=32, pc_gauss_kernel_size=27)
    inp = mg.synth.make_inputs(2, 500, 127)
    gt = mg.synth.disk_gt(2, 32)
    r = mg.run_reference(cfg27, inp, 4.5, torch.float32, {}, want_grads=False)
    up = dict(w_proj=((r["proj"] - gt) / 2).astype(np.float32))
    g = {k: v for k, v in mg.both(cfg27, inp, 4.5, up).items() if not k.startswith(("voxels", "drc_probs"))}
    mg.save("k27", sigma=4.5, K=27, D=32, Dz=32, **inp, **up, **g)

    cfgz = mg.make_cfg(vox_size=24, vox_size_z=8, pc_gauss_kernel_size=3)
    inpz = mg.tiny_inputs(seed=31, N=200)
    upz = mg.rand_upstream(32, 2, 8, 24)
    gz = {k: v for k, v in mg.both(cfgz, inpz, 0.7, upz).items() if not k.startswith(("voxels", "drc_probs"))}
    assert gz["taps_z_f32"].shape == (1,), gz["taps_z_f32"].shape
    mg.save("voxz_onetap", sigma=0.7, K=3, D=24, Dz=8, **inpz, **upz, **gz)


if __name__ == "__main__":
    main()
