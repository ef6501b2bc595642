#!/usr/bin/env python3
"""Record a step that holds RCCL collectives into a HIP graph again and again beside the live watchdog thread.

    python scripts/rccl_capture_stress.py --records 40              # with the drain (distributed.drain_watchdog)
    python scripts/rccl_capture_stress.py --records 40 --drain 0    # without: expected to die with hipErrorCapturedEvent

One rank, backend nccl (a forced one-rank group): a small network's backward with GradBuckets' bucket all-reduces,
recorded `--records` times; between recordings a few replays and an eager barrier (whose work the watchdog then holds).
Prints one line per run; exit code 0 iff every recording and replay went through."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=40)
    ap.add_argument("--drain", type=float, default=None, help="seconds (default: the library's 0.25); 0 = off")
    ap.add_argument("--step", default="mlp", choices=["mlp", "train"],
                    help="mlp: twelve linear layers (a capture of ~1 ms); train: the chair_unsupervised training step "
                         "(~250 launches, a capture of ~7 ms: the window the first RCCL run of round 5 died in)")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-barrier", action="store_true",
                    help="no eager collective before each recording (nothing of this script's is left in the watchdog's list)")
    ap.add_argument("--main-thread", action="store_true",
                    help="issue the bucket collectives from the capturing thread (in finish()) instead of from the "
                         "gradient hooks on the autograd thread")
    ap.add_argument("--pause-ms", type=float, default=0.0,
                    help="sleep this long between the eager barrier and the recording (moves the phase of the watchdog's "
                         "100 ms poll against the capture)")
    args = ap.parse_args()
    if args.drain is not None:
        os.environ["DPC_WATCHDOG_DRAIN_S"] = str(args.drain)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dd = dpc_amd.distributed
    rank, world, dev = dd.init("nccl", force=True)
    torch.manual_seed(0)
    if args.step == "mlp":
        net = torch.nn.Sequential(*[torch.nn.Linear(1024, 1024) for _ in range(12)]).to(dev)
        red = dd.GradBuckets(net.parameters(), bucket_mb=8)
        x = torch.randn(256, 1024, device=dev)

        def run():
            red.zero_()
            net(x).square().mean().backward()
            red.finish()
    else:
        sys.path.insert(0, os.path.join(ROOT, "examples", "chair_unsupervised"))
        import train_step as ts
        from nets import Im2PointCloud
        cfg = ts.make_cfg(batch_size=args.batch, pc_point_dropout=1.0, pc_point_dropout_scheduled=False)
        net = Im2PointCloud(cfg, 128).to(dev)
        red = dd.GradBuckets(net.parameters(), bucket_mb=64)
        projector = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=dev)
        torch.backends.cudnn.benchmark = True
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True, fused=True)
        inputs = ts.synthetic_batch(cfg, dev, 128, seed=0)
        projector.enable_graph_replay(follow_tap_counts=True)
        run = lambda: ts.train_step(net, projector, inputs, opt, world, buckets=red)
    if args.main_thread:
        # the hooks never see a complete bucket: finish() (capturing thread) issues every collective
        def arm():
            for i in range(len(red.buckets)):
                red._pending[i] = 1 << 30
            red._next = 0
        red._arm = arm
        arm()
    t0 = time.perf_counter()
    step = dpc_amd.graphs.RecordedStep(run, world=world, device=dev, collectives=True)
    for i in range(args.records - 1):
        for _ in range(5):
            step()
        if not args.no_barrier:
            dd.barrier(dev)              # an eager collective: its work sits in the watchdog's list for up to 100 ms
        if args.pause_ms > 0:
            time.sleep(args.pause_ms * 1e-3)
        sys.stderr.write("[stress] recording %d\n" % (i + 2))
        if args.no_barrier:              # (RecordedStep._record itself issues a barrier when the step holds collectives)
            step.collectives = False
        step._record()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print("OK: %d recordings of a step with %d bucket collectives, drain %s s%s%s, %.1f s [%s]" % (
        step.records, len(red.buckets), os.environ.get("DPC_WATCHDOG_DRAIN_S", "0.25"),
        ", no eager barrier" if args.no_barrier else "", ", collectives from the capturing thread" if args.main_thread else "",
        time.perf_counter() - t0, dd.collective_library()), flush=True)
    dd.finalize()


if __name__ == "__main__":
    main()
