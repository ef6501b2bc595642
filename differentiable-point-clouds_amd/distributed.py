"""One process per GPU.  The projector shards over instances (model x view x
pose candidate): every instance is independent through forward and backward
(no cross-instance reduction anywhere on the path), so ranks take contiguous
slices of the view batch and NO collective sits on the data path.  Ranks meet
only at barriers and at the max-over-ranks reduction of the step time (and, in
a full training step, at the gradient all-reduce of the encoder/decoder
parameters, which is outside this path).

backend "nccl" is RCCL on ROCm; CPU tests use "gloo".
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment.  Returns
    (rank, world, device).  A single process needs no process group."""
    rank, local_rank, world = env_world()
    if device is None:
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
        else:
            device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if device.type == "cuda" else "gloo"
        kw = {"device_id": device} if device.type == "cuda" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def shard_range(total, rank, world):
    """Contiguous, balanced slice [lo, hi) of `total` instances for `rank`."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the step time of the slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if device is None:      # the RCCL ("nccl") backend only reduces device tensors
        on_gpu = dist.get_backend() == "nccl"
        device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_views(local, device=None):
    """all_gather of per-rank [b_i, ...] tensors (equal b_i) -> [sum b_i, ...];
    used by tests / evaluation only, never inside the timed path."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local.contiguous())
    return torch.cat(parts, dim=0)


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
