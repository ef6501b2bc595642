"""BASELINE configs[2]/[3] plumbing at toy size: the full training step
(encoder -> decoder/pose -> projector -> loss -> Adam) runs, and 2-rank DDP
(gloo, CPU, kernel emulation library) produces the same parameter gradients as
one process on the concatenated batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples", "chair_unsupervised")


def _toy_cfg(ts, models):
    return ts.make_cfg(vox_size=16, pc_gauss_kernel_size=5, pc_num_points=64, pose_predict_num_candidates=2,
                       step_size=2, batch_size=models, pc_point_dropout=1.0)


def _grads(ts, nets, cfg, images, masks, world=1, ddp=False, buckets=False):
    import dpc_amd
    torch.manual_seed(0)
    net = nets.Im2PointCloud(cfg, image_size=32, f_dim=4, fc_dim=32, z_dim=32)
    model = torch.nn.parallel.DistributedDataParallel(net) if ddp else net
    red = (dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=0.05, gather=buckets if isinstance(buckets, str) else "accumulate")
           if buckets else None)   # several buckets
    proj = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
    outputs = proj.replicate_outputs(model(images))
    outputs = proj.compute_projection({"masks": masks}, outputs, is_training=False)
    loss = proj.add_proj_loss({"masks": masks}, outputs, 1.0)
    loss.backward()
    if red is not None:
        assert len(red.buckets) > 2
        red.finish()
    return {n: p.grad.clone() for n, p in net.named_parameters()}, float(loss)


def _data(models, seed=5):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(models * 2, 32, 32, 3, generator=g)
    masks = (torch.rand(models * 2, 32, 32, 1, generator=g) > 0.5).float()
    return images, masks


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (ROOT, EX):
        sys.path.insert(0, p)
    torch.set_num_threads(1)
    import dpc_amd
    import nets
    import train_step as ts
    emu = dpc_amd._capi.DpcLibrary(os.path.join(ROOT, "tests", "hipemu", "libdpc_emu.so"), host_memory=True)
    dpc_amd._capi.set_library(emu)
    dpc_amd.distributed.init("gloo", device=torch.device("cpu"))
    images, masks = _data(4)
    lo, hi = dpc_amd.distributed.shard_range(4, rank, world)        # shard over MODELS
    g, loss = _grads(ts, nets, _toy_cfg(ts, hi - lo), images[2 * lo:2 * hi], masks[2 * lo:2 * hi], world, ddp=True)
    # the recordable reducer (what bench.py --config 3 --graph --gpus N and train_step.py --graph use instead of DDP)
    gb, _ = _grads(ts, nets, _toy_cfg(ts, hi - lo), images[2 * lo:2 * hi], masks[2 * lo:2 * hi], world, buckets=True)
    # ... and its copy mode (round 6: autograd moves the gradients in, one multi-tensor copy packs each bucket)
    gc, _ = _grads(ts, nets, _toy_cfg(ts, hi - lo), images[2 * lo:2 * hi], masks[2 * lo:2 * hi], world, buckets="copy")
    if rank == 0:
        ret["grads"] = {k: v.numpy() for k, v in g.items()}
        ret["grads_buckets"] = {k: v.numpy() for k, v in gb.items()}
        ret["grads_buckets_copy"] = {k: v.numpy() for k, v in gc.items()}
    dpc_amd.distributed.finalize()


def test_training_step_runs_and_ddp_matches_single_process(emu):
    for p in (ROOT, EX):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nets
    import train_step as ts
    images, masks = _data(4)
    ref, loss = _grads(ts, nets, _toy_cfg(ts, 4), images, masks)
    assert np.isfinite(loss) and all(torch.isfinite(v).all() for v in ref.values())
    assert float(ref["decoder.pts.weight"].abs().max()) > 0 and float(ref["encoder.fc1.weight"].abs().max()) > 0
    port = 29500 + (os.getpid() % 2000) + 7
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    # each rank normalises by its local sample count (half), DDP averages the two ranks
    # => identical to the single-process gradient on the concatenated batch
    for k, v in ref.items():
        a, b = ret["grads"][k], v.numpy()
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max()), k
        for mode in ("grads_buckets", "grads_buckets_copy"):          # GradBuckets, both gather modes: same averages as DDP
            a = ret[mode][k]
            assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max()), (mode, k)


def test_grad_buckets_step_equals_plain_step(emu):
    """train_step(buckets=...) on one rank: gradients accumulate into the flat buckets in place, Adam steps on
    the views, the buckets are zeroed for the next step -- two steps give the same parameters as the plain path."""
    for p in (ROOT, EX):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dpc_amd
    import nets
    import train_step as ts
    cfg = _toy_cfg(ts, 2)
    images, masks = _data(2)
    finals = []
    for use_buckets in (False, "accumulate", "copy"):
        torch.manual_seed(0)
        net = nets.Im2PointCloud(cfg, image_size=32, f_dim=4, fc_dim=32, z_dim=32)
        proj = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        red = dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=0.05, gather=use_buckets) if use_buckets else None
        for _ in range(2):
            ts.train_step(net, proj, {"images": images, "masks": masks}, opt, is_training=False, buckets=red)
        finals.append({n: p.detach().clone() for n, p in net.named_parameters()})
    for k in finals[0]:
        assert torch.allclose(finals[0][k], finals[1][k], rtol=0, atol=1e-6), k
        assert torch.allclose(finals[0][k], finals[2][k], rtol=0, atol=1e-6), k


def test_optimizer_step_changes_parameters(emu):
    for p in (ROOT, EX):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dpc_amd
    import nets
    import train_step as ts
    cfg = _toy_cfg(ts, 2)
    torch.manual_seed(0)
    net = nets.Im2PointCloud(cfg, image_size=32, f_dim=4, fc_dim=32, z_dim=32)
    proj = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    images, masks = _data(2)
    before = net.decoder.pts.weight.detach().clone()
    l0 = ts.train_step(net, proj, {"images": images, "masks": masks}, opt, is_training=False)
    assert torch.isfinite(l0) and not torch.equal(before, net.decoder.pts.weight.detach())
