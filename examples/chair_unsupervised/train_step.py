#!/usr/bin/env python3
"""Full training step of BASELINE configs[2] (1 GPU) / configs[3] (DDP over N
GPUs): encoder -> decoder + pose candidates -> HIP projector -> min-over-
candidates silhouette loss (+ student loss) -> backward -> Adam.  Synthetic
images/masks, random-init weights.  Mirrors experiments/chair_unsupervised
(vox 64, K=21, sigma 3.0->0.2, 8000 points, 4 pose candidates, 5 views).

    python examples/chair_unsupervised/train_step.py --steps 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/chair_unsupervised/train_step.py --gpus 8

Data parallel over MODELS: each rank owns batch_size models and replicates
them over views / candidates locally, so the [B,N,3] replication never crosses
xGMI; the only exchange is DDP's bucketed all-reduce of the ~33 M parameter
gradients (backend "nccl" = RCCL).  Every rank normalises the loss by its
LOCAL number of samples; averaging the gradients over ranks (DDP, or the
recordable GradBuckets reducer under --graph) then reproduces the 1-GPU step
on the concatenated batch.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import dpc_amd  # noqa: E402
from nets import Im2PointCloud  # noqa: E402


def make_cfg(**kw):
    base = dict(vox_size=64, pc_gauss_kernel_size=21, pc_relative_sigma=3.0, pc_relative_sigma_end=0.2,
                pc_num_points=8000, predict_pose=True, pose_predict_num_candidates=4, step_size=5,
                batch_size=16, pose_predictor_student=True, pose_predictor_student_loss_weight=20.0,
                pc_point_dropout=0.07)
    base.update(kw)
    return dpc_amd.default_config(**base)


def synthetic_batch(cfg, device, image_size, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = cfg.batch_size * cfg.step_size
    images = torch.rand(n, image_size, image_size, 3, generator=g).to(device)
    yy, xx = torch.meshgrid(torch.arange(image_size), torch.arange(image_size), indexing="ij")
    c = (image_size - 1) / 2
    disk = (((yy - c) ** 2 + (xx - c) ** 2) <= (0.3 * image_size) ** 2).float()
    masks = disk.reshape(1, image_size, image_size, 1).expand(n, -1, -1, -1).contiguous().to(device)
    return {"images": images, "masks": masks}


def train_step(net, projector, inputs, optimizer, world=1, is_training=True, buckets=None):
    """One optimiser step.  `buckets` (dpc_amd.distributed.GradBuckets over the parameters of an UNWRAPPED net):
    the gradient all-reduce is issued per bucket from inside the backward pass and waited for before the
    optimiser -- the recordable alternative to DistributedDataParallel (whole step in one HIP graph per rank)."""
    cfg = projector.cfg()
    outputs = net(inputs["images"])
    outputs = projector.replicate_outputs(outputs)
    outputs = projector.compute_projection(inputs, outputs, is_training=is_training)
    # add_proj_loss normalises by the LOCAL sample count of this rank; the gradient AVERAGE over ranks (DDP, or
    # GradBuckets.finish) then equals the single-process gradient on the concatenated batch
    loss = projector.add_proj_loss(inputs, outputs, cfg.proj_weight)
    if buckets is None:
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        optimizer.step()
    else:
        loss.backward()                 # the buckets were zeroed by the previous step (or at construction)
        buckets.finish()
        optimizer.step()
        buckets.zero_()
    return loss.detach()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=16, help="models per GPU")
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--keep-prob", type=float, default=1.0, help="point dropout keep probability (1 = all 8000 points)")
    ap.add_argument("--scheduled", action="store_true",
                    help="anneal the dropout keep probability (from --keep-prob to 1) and the blur sigma over --max-steps")
    ap.add_argument("--max-steps", type=int, default=1000, help="length of the schedules (cfg.max_number_of_steps)")
    ap.add_argument("--graph", action="store_true",
                    help="record the whole step (nets, projector, loss, backward, Adam) into one HIP graph and replay it; "
                         "the schedules and the dropout draw keep moving (ModelPointCloud.enable_graph_replay).  With "
                         "--gpus N the gradient all-reduce is part of the recorded step (GradBuckets instead of DDP).")
    args = ap.parse_args()
    dd = dpc_amd.distributed
    rank, world, device = dd.init("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    cfg = make_cfg(batch_size=args.batch_size, pc_point_dropout=args.keep_prob,
                   pc_point_dropout_scheduled=args.scheduled, max_number_of_steps=args.max_steps,
                   **({} if args.scheduled else {"pc_relative_sigma_end": 3.0}))
    torch.manual_seed(0)
    net = Im2PointCloud(cfg, args.image_size).to(device)
    model, buckets = net, None
    dist_on = dd.active()          # several ranks, or a forced one-rank group (DPC_FORCE_DIST=1: the one-GPU rehearsal)
    if dist_on and args.graph:
        buckets = dd.GradBuckets(net.parameters(), bucket_mb=64, gather="copy")       # recordable bucketed all-reduce (no DDP wrapper)
    elif dist_on:
        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[device.index], bucket_cap_mb=64,
                                                          gradient_as_bucket_view=True)
    projector = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=device)
    torch.backends.cudnn.benchmark = True          # MIOpen find mode for the stock convolutions
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=args.graph, fused=True)
    inputs = synthetic_batch(cfg, device, args.image_size, seed=rank)
    run = lambda: train_step(model, projector, inputs, opt, world, buckets=buckets)
    if args.graph:
        # the shared recipe (dpc_amd/graphs.py): eager warm-up steps (11 with several ranks: RCCL sets its channels up
        # over the first collectives), barrier, thread-local capture (the RCCL watchdog polls events meanwhile), and a
        # new recording whenever the annealed blur moves on to a smaller tap count
        projector.enable_graph_replay(follow_tap_counts=True)
        run = dpc_amd.graphs.RecordedStep(run, world=world, device=device, collectives=dist_on,
                                             key=projector.recording_key)
    step = 0
    for _ in range(args.warmup):
        projector.set_global_step(step)          # sigma / dropout schedules (in place under --graph)
        run()
        step += 1
    dd.barrier(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        projector.set_global_step(step)
        loss = run()
        step += 1
    dd.barrier(device)
    dt = dd.max_over_ranks(time.perf_counter() - t0, device)
    if rank == 0:
        views = cfg.batch_size * cfg.step_size * cfg.pose_predict_num_candidates
        nparams = sum(p.numel() for p in net.parameters())
        print(json.dumps({"metric": "training steps/sec (chair_unsupervised, synthetic)", "value": args.steps / dt,
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
                          "projected_views_per_s": world * views * args.steps / dt, "scaling": "weak",
                          "config": {"models_per_gpu": cfg.batch_size, "views": cfg.step_size,
                                     "pose_candidates": cfg.pose_predict_num_candidates, "vox_size": cfg.vox_size,
                                     "K": cfg.pc_gauss_kernel_size, "points": int(cfg.pc_num_points * args.keep_prob),
                                     "params": nparams, "loss": float(loss), "hip_graph": bool(args.graph),
                                     "graph_recordings": getattr(run, "records", 0),
                                     "taps_at_end": list(projector.effective_tap_counts()),
                                     "collectives": dd.collective_library(),
                                     "scheduled": bool(args.scheduled)}}))
    dd.finalize()


if __name__ == "__main__":
    main()
