"""Recording a step into a HIP graph and replaying it -- the ONE recipe bench.py and the training example share.

The library only enqueues work on the stream it is handed, so a whole step (projector forward with its loss
epilogue, backward; or encoder + decoder + projector + loss + backward + the bucketed RCCL all-reduce of
``distributed.GradBuckets`` + Adam) records into one hipGraph per rank; replaying it costs the host one launch.

What the recipe has to get right (each item was a failure on a real box or is a documented trap):

* eager warm-up steps on a side stream first -- allocator, MIOpen solver search, optimiser state, lazy kernel
  loading; with more than one rank 11 of them, because RCCL sets its channels up over the first collectives;
* ``torch.cuda.synchronize()`` + a barrier before the capture starts, so that no rank is still inside an eager
  collective while another one is already recording;
* ``capture_error_mode="thread_local"``: the RCCL watchdog thread queries events while this thread records, which
  the default (global) mode turns into a capture error;
* the watchdog's list of outstanding works is given time to empty before the capture begins
  (``distributed.drain_watchdog``): for ~100 ms after a burst of eager collectives the watchdog still holds their works,
  and a poll that lands inside a capture holding collectives can come back with hipErrorCapturedEvent -> std::terminate
  -- the first run under RCCL in round 5 died that way, 5 of 16 stress processes without the pause, none with it;
  deterministic reproducer: scripts/rccl_capture_probe.py hooks_held (it takes collectives issued from the autograd thread);
* a capture that fails half way leaves the rank's streams in capture mode and its peers waiting inside a collective:
  with more than one rank the error is raised, not swallowed;
* nothing may keep the autograd graph of an EARLIER eager call of the same leaves alive while the step is recorded (a
  list of outputs that still carry grad_fn, say): torch then synchronises the capture stream with the stream that graph
  was built on ("AccumulateGrad node's stream does not match ..."), which on ROCm ends in a segfault when the capture
  closes.  Keep detached copies;
* the recorded kernels are those of ONE set of launch decisions.  ``key`` (a callable) names the decisions that can
  move between replays -- for the projector the effective tap counts of the annealed blur
  (``ModelPointCloud.effective_tap_counts``).  When the key changes the step is run eagerly once (that IS the step of
  this call; it also loads the newly selected kernels) and recorded again for the calls that follow.  Every rank
  evaluates the key from the same global step, so all ranks re-record in the same call.
"""
import os
import sys
import time

import torch

from . import distributed as dd


def _trace(msg):
    if os.environ.get("DPC_BENCH_TRACE") == "1":
        sys.stderr.write("[graphs %.3f] %s\n" % (time.perf_counter(), msg))
        sys.stderr.flush()


class RecordedStep(object):
    """``step = RecordedStep(run, world=..., device=..., key=...)``; then ``out = step()`` per training / bench step.
    ``run`` takes no arguments and returns a tensor (or a tuple / None) that lives in the graph's memory pool --
    read it after the replay, before the next one."""

    def __init__(self, run, world=1, device=None, key=None, warmup=None, collectives=None):
        self._run, self._key = run, key
        self.world = int(world)
        # does the step hold collectives?  (default: with several ranks; a forced ONE-rank process group -- the
        # one-GPU rehearsal of the multi-GPU path -- says so explicitly and gets the same warm-up and barrier)
        self.collectives = (self.world > 1) if collectives is None else bool(collectives)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.records = 0                    # how many times the step has been recorded (1 + re-records)
        self.graph, self.out = None, None
        self._warm(11 if self.collectives else 3) if warmup is None else self._warm(int(warmup))
        self._record()

    def _warm(self, n):
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(n):
                self._run()
        torch.cuda.current_stream(self.device).wait_stream(side)

    def _record(self):
        torch.cuda.synchronize(self.device)          # nothing of the previous recording is in flight any more ...
        self.graph, self.out = None, None            # ... before its memory pool is released
        if self.collectives:
            dd.barrier(self.device)
            dd.drain_watchdog()                      # (precaution: the watchdog's list is empty when the capture begins)
        graph = torch.cuda.CUDAGraph()
        _trace("capture begins")
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = self._run()
        _trace("capture ended")
        self.graph, self.out = graph, out
        self.key_value = self._key() if self._key is not None else None
        self.records += 1

    def __call__(self):
        if self._key is not None and self._key() != self.key_value:
            out = self._run()                        # this call's step, eagerly, with the newly selected kernels
            self._record()                           # (recording executes nothing)
            return out
        self.graph.replay()
        return self.out

    replay = __call__
