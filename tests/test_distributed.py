"""N>1 path on CPU: world_size-2 gloo processes shard a view batch, run the
projector (kernel emulation library) on their slice with no data-path
collective, and the gathered result equals the single-process result exactly
(instances are independent).  Also the barrier / max-over-ranks timing
reduction bench.py uses."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import dpc_amd
    dd = dpc_amd.distributed
    emu = dpc_amd._capi.DpcLibrary(os.path.join(ROOT, "tests", "hipemu", "libdpc_emu.so"), host_memory=True)
    dpc_amd._capi.set_library(emu)
    r, w, dev = dd.init("gloo", device=torch.device("cpu"))
    assert (r, w) == (rank, world)
    B, N, D, K = 4, 120, 16, 5
    inp = dpc_amd.synthetic.make_inputs(B, N, 42)
    lo, hi = dd.shard_range(B, rank, world)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    t = lambda a: torch.tensor(a[lo:hi], requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu"),
                                          scaling_factor=scale)
    gt = torch.tensor(dpc_amd.synthetic.disk_gt(B, D)[lo:hi])
    g = torch.autograd.grad(out["proj"], [pc], ((out["proj"] - gt) / B).detach())[0]   # GLOBAL num_samples
    dd.barrier(dev)
    proj_all = dd.gather_views(out["proj"].detach())
    gpc_all = dd.gather_views(g)
    tmax = dd.max_over_ranks(1.0 + rank)
    if rank == 0:
        ret["proj"] = proj_all.numpy()
        ret["gpc"] = gpc_all.numpy()
        ret["tmax"] = tmax
    dd.finalize()


def test_two_rank_sharded_projection_equals_single_process(emu):
    import dpc_amd
    port = 29500 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["tmax"] == 2.0
    B, N, D, K = 4, 120, 16, 5
    inp = dpc_amd.synthetic.make_inputs(B, N, 42)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    t = lambda a: torch.tensor(a, requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu"),
                                          scaling_factor=scale)
    gt = torch.tensor(dpc_amd.synthetic.disk_gt(B, D))
    g = torch.autograd.grad(out["proj"], [pc], ((out["proj"] - gt) / B).detach())[0]
    assert np.abs(ret["proj"] - out["proj"].detach().numpy()).max() < 1e-6
    assert np.abs(ret["gpc"] - g.numpy()).max() < 1e-6 * max(1.0, float(g.abs().max()))


def test_shard_range_is_a_partition():
    import dpc_amd
    for total in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [dpc_amd.distributed.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
