"""Round-5 cases on the CPU tiers (the GPU counterparts live in test_gpu_parity.py / test_chunk_sparse.py / test_caller.py)."""
import numpy as np
import pytest
import torch

import dpc_amd
from helpers import synth
from oracle import dpc_oracle_np as onp


@pytest.mark.parametrize("D,Dz,K,sigma", [(72, 24, 3, 0.8), (112, 32, 5, 1.0)])
def test_emu_one_tap_z_filter_projects_like_the_oracle(emu, D, Dz, K, sigma):
    """vox_size_z so much smaller than vox_size that round(K * vox_size_z / vox_size) is 1 (gauss_kernel.py:35-54): the z
    filter is the single tap [1.0]; round 4 classified it as a second x filter and raised.  Forward and point gradient
    against the float64 NumPy oracle."""
    B, N = 1, 300
    inp = synth.make_inputs(B, N, 5)
    cfg = dpc_amd.default_config(vox_size=D, vox_size_z=Dz, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sigma, device="cpu")
    assert tuple(kern[2].shape[:3]) == (1, 1, 1)
    t = lambda a: torch.tensor(a, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
    w = np.random.default_rng(1).standard_normal(tuple(out["proj"].shape))
    g, = torch.autograd.grad(out["proj"], [pc], torch.tensor(w, dtype=torch.float32))
    f64 = lambda a: a.astype(np.float64)
    taps = onp.smoothing_taps(D, Dz, K, sigma)
    assert [len(x) for x in taps] == [K, K, 1]
    fw = onp.project_forward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, Dz=Dz, D=D)
    bw = onp.project_backward(f64(inp["pc"]), f64(inp["pose"]), None, f64(inp["scale"]), None, taps, fw, dproj=w)
    assert np.abs(out["proj"].detach().numpy() - fw["proj"]).max() <= 2e-5
    assert np.abs(g.numpy() - bw["dpc"]).max() <= 2e-4 * np.abs(bw["dpc"]).max()


def test_grad_buckets_reduce_whenever_a_group_exists():
    """distributed.GradBuckets: without a process group plain accumulation; the collective path (and ReduceOp.AVG under the
    nccl backend only) is decided from the group, not from world > 1 -- a forced one-rank group issues every all-reduce."""
    import torch.distributed as dist
    net = torch.nn.Linear(4, 3)
    red = dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=1)
    assert not dist.is_initialized() and red.reduce is False and red.in_collective_average is False and red.world == 1
    net(torch.ones(2, 4)).sum().backward()
    red.finish()
    assert torch.equal(net.weight.grad, torch.full((3, 4), 2.0))
