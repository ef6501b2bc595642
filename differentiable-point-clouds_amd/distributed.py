"""One process per GPU.  The projector shards over instances (model x view x
pose candidate): every instance is independent through forward and backward
(no cross-instance reduction anywhere on the path), so ranks take contiguous
slices of the view batch and NO collective sits on the data path.  Ranks meet
only at barriers and at the max-over-ranks reduction of the step time (and, in
a full training step, at the gradient all-reduce of the encoder/decoder
parameters, which is outside this path).

backend "nccl" is RCCL on ROCm; CPU tests use "gloo".
"""
import contextlib
import os
import sys
import time

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def active():
    """True iff a process group exists: the distributed machinery (barriers, reducers, the bench line's `parallelism`
    note) keys on THIS, not on world > 1 -- a forced one-rank group (below) runs the same code as eight ranks."""
    return dist.is_available() and dist.is_initialized()


def init(backend=None, device=None, force=None):
    """Initialise torch.distributed from the torchrun environment.  Returns
    (rank, world, device).  A single process needs no process group -- unless `force` (default: DPC_FORCE_DIST=1 in
    the environment; `bench.py --force-dist`) asks for one: a ONE-rank group under backend "nccl" creates a real RCCL
    communicator on the device, runs its watchdog thread, and sends every barrier / all-reduce / DDP bucket /
    GradBuckets collective through ProcessGroupNCCL on its own stream -- the whole multi-GPU code path on a one-GPU
    box (what it cannot show is bytes over xGMI)."""
    rank, local_rank, world = env_world()
    if force is None:
        force = os.environ.get("DPC_FORCE_DIST") == "1"
    if device is None:
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
        else:
            device = torch.device("cpu")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if device.type == "cuda" else "gloo"
        kw = {"device_id": device} if device.type == "cuda" else {}
        with _stdout_to_stderr():     # RCCL prints a version banner on STDOUT when its first communicator comes up;
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)      # stdout is for the ONE JSON line
            if backend == "nccl" and os.environ.get("DPC_INIT_BARRIER", "1") == "1":      # (0: dev, scripts/dev_r05)
                dist.barrier()        # the communicator is created here at the latest
                torch.cuda.synchronize(device)
    return rank, world, device


@contextlib.contextmanager
def _stdout_to_stderr():
    """file-descriptor-level redirect (the banner comes from C code): fd 1 points at fd 2 inside the block"""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def bind_to_gpu_numa(device):
    """Pin this process to the CPUs of the NUMA node its GPU hangs off (one process per GPU: the launch thread, the
    RCCL proxy thread and the host staging buffers then stay on the socket next to the device).  The node comes from
    sysfs (/sys/bus/pci/devices/<bdf>/local_cpulist); anything missing -- no such file, a container without the
    topology, a platform without sched_setaffinity -- leaves the affinity alone.  Returns a short description for the
    bench line, or None."""
    try:
        if device is None or torch.device(device).type != "cuda":
            return None
        pr = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/%s/" % bdf
        with open(base + "local_cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        if not cpus or cpus == allowed:
            return None
        # sched_setaffinity(0) moves the CALLING thread only; the pools that already exist (OpenMP / ATen, the HIP
        # runtime's and RCCL's helper threads) are moved one by one -- threads started later inherit the mask
        moved = 0
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            tids = [0]
        for tid in tids:
            try:
                os.sched_setaffinity(tid, cpus)
                moved += 1
            except OSError:              # a thread that exited meanwhile
                pass
        if moved == 0:
            os.sched_setaffinity(0, cpus)
        node = "?"
        try:
            with open(base + "numa_node") as f:
                node = f.read().strip()
        except OSError:
            pass
        return "GPU %s -> NUMA node %s (%d CPUs, %d threads bound)" % (bdf, node, len(cpus), max(moved, 1))
    except Exception:  # noqa: BLE001 -- placement only affects speed
        return None


def collective_library():
    """what the "nccl" backend really is here, for the bench line: RCCL's version and the world size it sees"""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    desc = "backend %s, world %d" % (dist.get_backend(), dist.get_world_size())
    if dist.get_backend() == "nccl":
        try:
            desc += ", RCCL %s" % ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            pass
    return desc


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


def drain_watchdog(seconds=None):
    """Give ProcessGroupNCCL's watchdog thread time to reap the works of the collectives issued so far; called after a
    synchronize and BEFORE a HIP-graph capture that will hold collectives.

    The very first run of this code under RCCL (round 5, one rank, `bench.py --config 3 --graph --force-dist`; log in
    profiles/r05/rccl_first_contact_abort.txt) was killed by the watchdog: its poll of a work's end event
    (WorkNCCL::isCompleted -> hipEventQuery) came back with hipErrorCapturedEvent ("operation not permitted on an event
    last recorded in a capturing stream"), which ProcessGroupNCCL turns into std::terminate.  What the hunt established
    (profiles/r05/rccl_abort_hunt.txt): without this pause 5 of 16 stress processes died, every one within the first
    ~130 ms after the burst of eager collectives of the warm-up -- the time the watchdog (100 ms cadence) still holds
    their works -- also when no eager collective at all is issued between the recordings, and none later in ~10 000
    recordings; with it, none in ~60 processes / ~900 recordings.  So the collision needs leftover eager works in the
    watchdog's list AND a capture that holds a collective AND a poll landing inside it; the pause removes the first.
    Made deterministic in scripts/rccl_capture_probe.py: `hooks_held` (20 eager steps whose collectives come from
    gradient hooks, i.e. from the autograd engine's thread, then at once a capture of the same step held open 0.3 s)
    dies 4 of 4; with 0.3 s between the eager steps and the capture it passes; the same ingredients issued from the
    CAPTURING thread (200 eager all-reduces, on the default or a side stream, then a held-open capture with a
    collective) pass every time.  So it takes collectives issued from another thread than the one that began the
    thread-local capture -- which is where every backward pass's hooks run (GradBuckets here, DDP's reducer alike).
    torch 2.10 no longer holds a capture back until the watchdog's list is empty (the pending-event-query
    counter of earlier releases is gone) and exposes no call that waits for it: everything issued so far has completed
    (the caller synchronised), the next poll removes it, and 2.5 poll periods are waited for here.  0.25 s per
    recording; DPC_WATCHDOG_DRAIN_S=0 switches it off (scripts/rccl_capture_stress.py --drain 0 shows the abort)."""
    # BEST EFFORT, by construction: a fixed pause sized for the watchdog's default 100 ms cadence.  It does not cover works of other
    # process groups created between the pause and the capture, nor a watchdog slowed down by TORCH_NCCL_* settings (raise
    # DPC_WATCHDOG_DRAIN_S accordingly); a collision that still happens ends in std::terminate inside ProcessGroupNCCL.
    if not active() or dist.get_backend() != "nccl":
        return
    if seconds is None:
        seconds = float(os.environ.get("DPC_WATCHDOG_DRAIN_S", "0.25"))
    if seconds > 0:
        time.sleep(seconds)


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


_DIAG = os.environ.get("DPC_DIAG_HOOK") == "1"     # dev: say from which thread / stream every bucket collective is issued


class GradBuckets(object):
    """Bucketed gradient all-reduce that can be RECORDED: the parameters' .grad tensors are views into a few flat
    buckets (filled in place by autograd), a post-accumulate hook per parameter counts its bucket down, and the
    bucket's all-reduce is issued -- asynchronously, on the process group's own stream -- as soon as its last
    gradient has landed, i.e. it overlaps with the rest of the backward pass exactly like DDP's reducer.  Unlike
    DDP there is no host-side bookkeeping between steps (no reducer rebuild, no unused-parameter search), so the
    whole step -- forward, backward, these collectives, the optimiser -- records into ONE HIP graph per rank
    (RCCL collectives are stream work; torch records the fork / join of the communication stream as graph edges).

        buckets = GradBuckets(net.parameters(), bucket_mb=64)      # once; every rank the same parameter order
        loss.backward(); buckets.finish()                          # per step: grads are now the rank AVERAGE
        optimizer.step(); buckets.zero_()                          # (never set_to_none: the views must survive)

    Averaging matches DDP: sum over ranks, divided by the world size.  Under RCCL the division rides inside the
    collective (ReduceOp.AVG: RCCL pre-multiplies by 1 / world as it reduces -- no second pass over the 133 MB of
    buckets); other backends (gloo has no AVG) sum and divide afterwards.  Works with any backend (the gloo tests
    check it against a single process on the concatenated batch); without a process group it degenerates to plain
    accumulation into the flat buffers.  A ONE-rank process group (distributed.init(force=True)) still issues every
    collective: that is the point of it -- the same calls, streams and graph edges as with eight ranks.

    Invariants, checked (a violation would otherwise let the ranks diverge silently):
    * every p.grad must still BE its bucket view when finish() runs -- optimizer.zero_grad() (set_to_none=True by
      default) or backward(create_graph=True) replace it, after which the hooks would reduce stale buffers:
      finish() raises;
    * ONE backward pass per finish(): a second one would drive a bucket's count negative (its all-reduce has
      already been issued with the first pass's gradients): the hook raises.  Accumulate into the buckets by
      calling finish() once per micro-step, or scale the loss instead;
    * the collectives are issued in BUCKET ORDER on every rank (a bucket whose gradients are complete waits for
      the buckets before it), so ranks whose autograd graphs finish buckets in different orders -- or leave
      different parameters without a gradient -- still issue matching all-reduces."""

    def __init__(self, params, bucket_mb=64, process_group=None, gather="accumulate", average="auto"):
        """gather="accumulate" (default): every p.grad IS its bucket view for the whole run; autograd accumulates into it in
        place (one add kernel per parameter and step -- ~100 launches of a few microseconds for the 33 M-parameter nets -- plus the
        zero-fill of the buckets by zero_()).  gather="copy" (round 6): the .grad tensors are left to autograd (None before the
        backward pass, so it MOVES each gradient in: no kernel); when the last gradient of a bucket has landed, ONE multi-tensor
        copy packs the bucket (torch._foreach_copy_), its all-reduce goes out as before, and every p.grad of the bucket is
        re-pointed at its bucket view -- the optimiser reads the reduced values there.  zero_() then only drops the .grad
        references.  Measured under a one-rank RCCL group (profiles/r06/rccl_world1.txt): the recorded training step 3.26 -> see
        there, against 2.54 ms without a reducer."""
        if gather not in ("accumulate", "copy"):
            raise ValueError("gather must be 'accumulate' or 'copy'")
        if average not in ("auto", "collective", "divide"):
            raise ValueError("average must be 'auto', 'collective' or 'divide'")
        self.gather = gather
        self.group = process_group
        self.reduce = active()              # a process group exists (possibly of ONE rank): the collectives are issued
        self.world = dist.get_world_size(process_group) if self.reduce else 1
        # Where the division by the world size happens.  "auto": inside the collective under RCCL (ReduceOp.AVG) when there is
        # more than one rank; with ONE rank the average IS the sum, so the all-reduce goes out as a plain SUM -- every call,
        # stream and graph edge is still there, but RCCL has nothing to do (its one-rank AVG is a pre-multiplied-sum kernel over
        # the bucket plus host-side setup: measured, the recorded training step under a one-rank group 2.93 -> 2.61 ms against
        # 2.53 without a reducer, profiles/r06/rccl_world1.txt); other backends sum and divide afterwards.  "collective": AVG
        # whenever the backend has it, also with one rank (what rounds 5 and 6 quoted as the world-1 overhead); "divide": always
        # sum, then divide.  DPC_BUCKET_AVG=1 / 0 in the environment forces "collective" / "divide" (bench.py's A/B switch).
        env = os.environ.get("DPC_BUCKET_AVG")
        if env in ("0", "1"):
            average = "collective" if env == "1" else "divide"
        self.average = average
        self.in_collective_average = self.reduce and dist.get_backend(process_group) == "nccl" and average != "divide"
        self._avg_op = self.in_collective_average and (self.world > 1 or average == "collective")
        params = [p for p in params if p.requires_grad]
        limit = int(bucket_mb * (1 << 20))
        # buckets in REVERSE parameter order: the backward pass produces the last layers' gradients first
        self.buckets, self._of = [], {}
        cur, cur_bytes = [], 0
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > limit or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._diag_left = 400
        self._works = []
        self._next = 0                 # first bucket whose all-reduce has not been issued in this step
        self._views = [(p, p.grad.data_ptr()) for _, plist in self.buckets for p in plist]
        self._view_of = {p: p.grad for _, plist in self.buckets for p in plist}
        if self.gather == "copy":
            for p in params:
                p.grad = None
        for p in params:
            p.register_post_accumulate_grad_hook(self._hook)
        self._arm()

    def _close(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            piece = flat[off:off + p.numel()]
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            # same strides as the parameter (channels-last filters stay channels-last): autograd's layout contract
            p.grad = piece.as_strided(p.shape, p.stride()) if dense else piece.view_as(p)
            self._of[p] = len(self.buckets)
            off += p.numel()
        self.buckets.append((flat, list(plist)))

    def _arm(self):
        for i, (_, plist) in enumerate(self.buckets):
            self._pending[i] = len(plist)
        self._next = 0

    def _issue_ready(self, force=False):
        """issue, in bucket order, the all-reduce of every leading bucket that is complete (force: of all that are left)"""
        while self._next < len(self.buckets) and (force or self._pending[self._next] == 0):
            if self.gather == "copy":
                self._pack(self._next)
            if self.reduce:
                if _DIAG and self._diag_left > 0:
                    import sys
                    import threading
                    self._diag_left -= 1
                    sys.stderr.write("[GradBuckets] bucket %d issued from thread %s, stream %s, capturing %s\n" % (
                        self._next, threading.current_thread().name, torch.cuda.current_stream(),
                        torch.cuda.is_current_stream_capturing()))
                op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
                self._works.append(dist.all_reduce(self.buckets[self._next][0], op=op, group=self.group, async_op=True))
            self._next += 1

    def _pack(self, i):
        """copy mode: the gradients autograd left in p.grad -> the bucket (one multi-tensor launch), p.grad -> its bucket view.
        A parameter without a gradient this step keeps p.grad = None (the optimiser skips it, as without a reducer); its
        part of the bucket is zeroed so that every rank reduces defined values."""
        plist = self.buckets[i][1]
        have = [p for p in plist if p.grad is not None and p.grad.data_ptr() != self._view_of[p].data_ptr()]
        if have:
            torch._foreach_copy_([self._view_of[p] for p in have], [p.grad for p in have])
        for p in plist:
            if p.grad is None:
                self._view_of[p].zero_()
            else:
                p.grad = self._view_of[p]

    def _hook(self, p):
        i = self._of[p]
        self._pending[i] -= 1
        if self._pending[i] < 0:
            raise RuntimeError("GradBuckets: a parameter received a second gradient before finish() -- one backward "
                               "pass per finish() (its bucket's all-reduce has already been issued)")
        if self._pending[i] == 0:
            self._issue_ready()

    def finish(self):
        """wait (stream-wise) for the bucket collectives and turn the sums into averages; re-arm for the next step"""
        if self.gather == "accumulate":
            for p, ptr in self._views:
                if p.grad is None or p.grad.data_ptr() != ptr:
                    raise RuntimeError("GradBuckets: the .grad of a parameter is no longer its bucket view (optimizer.zero_grad() "
                                       "sets it to None by default -- use buckets.zero_(); backward(create_graph=True) replaces "
                                       "it): the buckets hold stale gradients")
        self._issue_ready(force=True)        # buckets with parameters that received no gradient this step: reduce anyway
        if self.reduce:
            for w in self._works:
                w.wait()
            if not self.in_collective_average and self.world > 1:
                for flat, _ in self.buckets:
                    flat.div_(self.world)
        self._works = []
        self._arm()

    def zero_(self):
        if self.gather == "copy":            # nothing to fill: autograd moves the next step's gradients in
            for _, plist in self.buckets:
                for p in plist:
                    p.grad = None
            return
        for flat, _ in self.buckets:
            flat.zero_()
