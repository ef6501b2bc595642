#!/usr/bin/env python3
"""Headline benchmark: projected views/sec, forward + backward, of
pointcloud_project_fast on synthetic point clouds.

    python bench.py                                   # 1 GPU, BASELINE.json configs[1]
    python bench.py --gpus 8 --steps 20 --warmup 5    # spawns 8 ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...   # or under an external launcher

--config  1  configs[0]   1000 pts ->  64^3, K=11, 4 views            (launch-bound plumbing case)
          2  configs[1]   8000 pts -> 128^3, K=11, 32 views per GPU   (DEFAULT, the headline)
          3  configs[2]   full chair_unsupervised training step: encoder + decoder + pose candidates (stock
                          PyTorch) -> HIP projector (16 models x 5 views x 4 candidates = 320 views of
                          8000*keep pts -> 64^3, K=21) -> HIP silhouette-loss epilogue -> backward -> Adam.
                          With --gpus N > 1 this is configs[3]: DDP over models, RCCL all-reduce of the
                          parameter gradients.  --projector-only times just the projector at that shape.
          5  configs[4]   16000 pts -> 256^3, K=11, 8 views            (HBM stress)

One "step" = one fwd+bwd pass over one batch of B views per GPU (grads w.r.t. point_cloud, transform,
scaling_factor given dproj = (proj - gt)/B, the reference's L2 loss gradient, model_pc.py:414-415; config 3:
one optimiser step).  The projector shards over instances with no data-path collective (weak scaling, B views
per GPU); ranks only meet at the timing barriers (config 3: plus DDP's gradient all-reduce).

Rank 0 prints ONE JSON line.  `value` comes from EXACTLY --steps steps after --warmup untimed ones, bracketed
by barrier + torch.cuda.synchronize() on both sides, max over ranks (--burn-in seconds of further untimed steps come
BEFORE the warm-up steps, so that the clocks have settled: config.burn_in_s).  Besides the contract keys:
  timing       --repeats further blocks of --steps steps, each timed with HIP events: median / p10 / p90
  roofline     dominant projector kernel: the bytes THIS implementation must move in that launch (grids: occupied
               planes only, counted from this run's own point clouds; point records, images) / its mean launch
               duration (HIP events on the launch stream, dpc_profile_*) = `achieved`, over the 8 TB/s HBM3E peak =
               `frac` (a physical fraction, never above 1).  `ceilings`: a read-only, a write-only and a copy stream
               over 512 MiB measured in this run; `vs_ceiling` = achieved over the mix of those that matches the
               kernel's read / write split.  `traffic` = PMC-measured HBM bytes of that launch from
               profiles/traffic.json -- only when that file was taken on the very library being timed (sha256
               stamp), else null with a note.  step_* = the same for the projector's whole fwd+bwd.  `model` =
               SURVEY.md 8(d)'s stage model (8 V + P per view: what an UNFUSED pipeline has to move), kept for
               reference: its rates describe a pipeline this is not and may exceed the peak
  cpu_baseline oracle/reference_cpu.py (op-for-op torch-CPU restatement of the reference graph, kind "port")
               on a bounded sample on this host's cores (rank 0, N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import dpc_amd  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DRY_RUN = os.environ.get("DPC_BENCH_DRY_RUN") == "1"   # tests only: CPU emulation library + gloo, not a measurement

# projector shape of the training step (experiments/chair_unsupervised/config.yaml:7-14)
dpc_amd.synthetic.CONFIGS.setdefault(3, dict(B=320, N=8000, D=64, K=21, sigma=3.0))


_T0 = time.perf_counter()


def trace(msg):
    """progress marks on stderr (DPC_BENCH_TRACE=1): where a run was when something outside Python's reach ended it"""
    if os.environ.get("DPC_BENCH_TRACE") == "1":
        sys.stderr.write("[bench +%.2fs] %s\n" % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


def build_case(cfg_id, B, device, seed_offset=0, kind="shell", N=None, sigma=None, K=None, D=None):
    over = {k: v for k, v in (("N", N), ("sigma", sigma), ("K", K), ("D", D)) if v is not None}
    if over:
        dpc_amd.synthetic.CONFIGS[cfg_id] = dict(dpc_amd.synthetic.CONFIGS[cfg_id], **over)
    c = dpc_amd.synthetic.config_inputs(cfg_id, B=B, kind=kind, seed_offset=seed_offset)
    cfg = dpc_amd.default_config(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    t = lambda a: torch.tensor(a, device=device, requires_grad=True)
    case = dict(cfg=cfg, pc=t(c["pc"]), pose=t(c["pose"]), scale=t(c["scale"]),
                kern=dpc_amd.smoothing_kernel(cfg, c["sigma"], device=device),
                gt=torch.tensor(dpc_amd.synthetic.disk_gt(c["B"], c["D"]), device=device),
                B=c["B"], N=c["N"], D=c["D"], K=c["K"], sigma=c["sigma"])
    return case


def step(case):
    # forward with the L2 silhouette loss fused into the collapse kernel (model_pc.py:414-415 without pose
    # candidates): dproj = (proj - gt) / B comes out of k_zfwd's registers; then the backward pass
    out = dpc_amd.pointcloud_project_fast(case["cfg"], case["pc"], case["pose"], None, None, case["kern"],
                                          scaling_factor=case["scale"], l2_target=(case["gt"], 1.0 / case["B"]))
    return torch.autograd.grad(out["proj"], [case["pc"], case["pose"], case["scale"]], out["proj_l2_grad"])


def build_train_case(args, device, rank, world):
    """configs[2]/[3]: the full training step of examples/chair_unsupervised."""
    sys.path.insert(0, os.path.join(ROOT, "examples", "chair_unsupervised"))
    import train_step as ts
    from nets import Im2PointCloud
    models = args.batch or 16
    image, toy, net_kw = 128, {}, {}
    if DRY_RUN:    # toy sizes for the launch / DDP / reporting logic on the CPU emulation tier (never a measurement)
        models, image = 1, 32
        toy = dict(vox_size=32, pc_gauss_kernel_size=5, pc_relative_sigma=0.9, pc_num_points=150, step_size=2,
                   pose_predict_num_candidates=2)
        net_kw = dict(f_dim=4, fc_dim=32, z_dim=32)
    if args.sigma is not None:
        toy = dict(toy, pc_relative_sigma=args.sigma)
    cfg = ts.make_cfg(batch_size=models, pc_point_dropout=args.keep_prob, pc_point_dropout_scheduled=False, **toy)
    torch.manual_seed(0)
    net = Im2PointCloud(cfg, image, **net_kw).to(device)
    model, buckets = net, None
    dist_on = dpc_amd.distributed.active()     # several ranks, or one rank under --force-dist
    if dist_on and args.graph:
        # a recorded step cannot hold DDP's reducer (host-side bucket bookkeeping between steps): the same bucketed,
        # backward-overlapped all-reduce is issued by gradient hooks instead (dpc_amd.distributed.GradBuckets)
        # (gather="copy": autograd moves the gradients in and one multi-tensor copy packs each bucket -- no add kernel per parameter,
        # no zero-fill; DPC_BUCKET_GATHER=accumulate is the round-5 form, for A/B)
        buckets = dpc_amd.distributed.GradBuckets(net.parameters(), bucket_mb=float(os.environ.get("DPC_BUCKET_MB", "64")),
                                                   gather=os.environ.get("DPC_BUCKET_GATHER", "copy"))
    elif dist_on:
        ddp_kw = dict(device_ids=[device.index]) if device.type == "cuda" else {}
        model = torch.nn.parallel.DistributedDataParallel(net, bucket_cap_mb=64, gradient_as_bucket_view=True, **ddp_kw)
    projector = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device=device)
    # stock PyTorch settings for the layers around the projector: MIOpen's find mode for the convolutions, and the
    # single-kernel (fused) Adam -- the foreach default moves the 33 M parameters in ~15 launches at 1.8 TB/s
    fused = device.type == "cuda" and os.environ.get("DPC_ADAM_FUSED", "1") == "1"
    if device.type == "cuda" and "DPC_CUDNN_BENCHMARK" not in os.environ:
        torch.backends.cudnn.benchmark = True
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=bool(args.graph) and not DRY_RUN, **({"fused": True} if fused else {}))
    inputs = ts.synthetic_batch(cfg, device, image, seed=rank)
    views = cfg.batch_size * cfg.step_size * cfg.pose_predict_num_candidates
    if args.graph and not DRY_RUN:
        projector.enable_graph_replay(follow_tap_counts=True)   # blur taps and the dropout's {keep, seed} at fixed device addresses
    case = dict(B=views, N=int(cfg.pc_num_points * args.keep_prob), D=cfg.vox_size, K=cfg.pc_gauss_kernel_size,
                sigma=cfg.pc_relative_sigma, models=models, params=sum(p.numel() for p in net.parameters()),
                views_per_model=cfg.step_size, candidates=cfg.pose_predict_num_candidates,
                reducer="GradBuckets" if buckets is not None else ("DDP" if dist_on else None), projector=projector,
                run=lambda: ts.train_step(model, projector, inputs, opt, world, buckets=buckets))
    return case


def plane_occupancy(case, k_run=None, chunk_sparse=False):
    """Occupied planes per view (k_splat_xy writes, k_zfwd / k_zbwd / k_gather_yx read only planes that hold trilinear
    mass: a valid point in depth cell z-1 or z), counted on the host from one forward call's tr_pc -- so that the byte
    counts below are what the kernels have to move for THESE clouds, not a dense-grid model.  With the chunk-sparse
    layout (third entry of the result, else None) also the 128-byte chunks the kernels touch: chunk c of row y of plane
    z holds anything iff a point has a corner on plane z whose corner rows lie within K/2 rows of y and whose corner
    columns, widened by K/2, reach into columns 32 c .. 32 c + 31 (the marks k_splat_xy leaves: geometry, not values)."""
    with torch.no_grad():
        out = dpc_amd.pointcloud_project_fast(case["cfg"], case["pc"], case["pose"], None, None, case["kern"],
                                              scaling_factor=case["scale"])
        tr = out["tr_pc"].detach().cpu().numpy().astype(np.float32)
    D = case["D"]
    valid = np.all((tr >= -0.5) & (tr <= 0.5), axis=-1)
    cell = lambda i: np.floor((tr[..., i] + np.float32(0.5)) * np.float32(D - 1)).astype(np.int64).clip(0, D - 1)
    iz = cell(0)
    live = 0
    for b in range(tr.shape[0]):
        cells = np.zeros(D + 1, dtype=bool)
        cells[iz[b][valid[b]]] = True
        live += int((cells[:D] | np.concatenate([[False], cells[:D - 1]])).sum())
    chunks, chunks_g2 = None, 0
    if chunk_sparse:
        H = int(k_run if k_run is not None else case["K"]) // 2
        iy, ix = cell(1), cell(2)
        chunks = 0
        for b in range(tr.shape[0]):
            v = valid[b]
            z0, y0, x0 = iz[b][v], iy[b][v], ix[b][v]
            mark = np.zeros((D + 1, D, D // 32), dtype=bool)
            c_lo, c_hi = np.maximum(x0 - H, 0) >> 5, np.minimum(x0 + 1 + H, D - 1) >> 5
            for dy in range(-H, H + 2):
                y = y0 + dy
                ok = (y >= 0) & (y < D)
                for dz in (0, 1):
                    for c in (c_lo, c_hi):               # (2 H + 2 <= 32 columns: at most two chunks)
                        mark[z0[ok] + dz, y[ok], c[ok]] = True
            mark = mark[:D]
            chunks += int(mark.sum())
            # G2 (saved for the backward pass under this layout): the marks widened by K/2 planes either way
            wide = mark.copy()
            for d in range(1, H + 1):
                wide[d:] |= mark[:-d]
                wide[:-d] |= mark[d:]
            chunks_g2 += int(wide.sum())
    return live, int(valid.sum()), chunks, (chunks_g2 if chunk_sparse else None)


def kernel_bytes(label, case, save_xy, occ):
    """(read, written) HBM bytes THIS implementation must move in one launch of `label` over the batch (DESIGN.md
    'Kernels').  occ = (occupied planes summed over the views, valid points) or None (dense model: every plane)."""
    B, N, D = case["B"], case["N"], case["D"]
    plane = 4 * D * D
    L = occ[0] if occ else B * D                     # planes that exist
    img = B * plane                                  # one [B,D,D] image
    LG = B * D                                       # planes of G2 (saved when the xy grid is not): dense
    if occ and len(occ) > 2 and occ[2] is not None:
        # chunk-sparse layout: of the occupied planes only the marked 128-byte chunks move (L in units of planes), plus the
        # chunk marks themselves (a byte per row of an occupied plane, written twice by the splat, read once per consumer);
        # G2 is stored / read where a marked chunk lies within K/2 planes
        marks = L * D
        L = occ[2] * 128.0 / plane
        LG = occ[3] * 128.0 / plane
    else:
        marks = 0
    fused = {
        # points 12 B in; tr_pc 12 + record 16 + slot 4 out
        "zsort": (B * N * 12, B * N * 32), "zhist": (B * N * 12, B * N * 16), "zscatter": (B * N * 16, B * N * 20),
        # every point's record is read by the two planes it touches; 4 clip bytes per point; occupied planes out
        "splat_xy": (B * N * 32, L * plane + B * N * 4 + 2 * marks),
        # occupied planes in (+ all of G2 out at > 11 taps); proj, depth, loss gradient (+ its target in), fp64 sums
        "zfwd": (L * plane + img + marks, (0 if save_xy else LG * plane) + 3 * img + 4 * img),
        "zbwd": ((L if save_xy else LG) * plane + 4 * img + img + marks, L * plane),
        # occupied planes + records + clip bytes in; 12-byte partials out (2 or 4 per point)
        "gather_yx": (L * plane + B * N * 32 + B * N * 4 + marks, B * N * 12 * (2 if D <= 64 else 4)),
        "points_bwd": (B * N * (12 * (2 if D <= 64 else 4) + 12 + 12 + 4), B * N * 12),
    }
    if label in fused:
        return fused[label]
    V = D * plane
    generic = {"memset_grid": (0, B * V), "points_fwd": (B * N * 12, B * N * 12 + B * N * 8 * 8),
               "blur_plane": (B * V, B * V), "blur_xy": (B * V, B * V), "blur_z": (B * V, B * V)}
    return generic.get(label, (0, 0))


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def cpu_baseline(cfg_id, seconds_budget=20.0):
    """oracle/reference_cpu.py on a bounded sample (B = 2 views) of the same workload: 1 thread, every host
    CPU, and the best of a few thread counts (median of >= 3 runs each; `value` is the best one)."""
    from oracle import reference_cpu as rcpu
    c = dpc_amd.synthetic.config_inputs(cfg_id, B=2)
    cfg = rcpu.Cfg(vox_size=c["D"], pc_gauss_kernel_size=c["K"])
    kern = rcpu.smoothing_kernel(cfg, c["sigma"])
    gt = torch.tensor(dpc_amd.synthetic.disk_gt(2, c["D"]))
    pc = torch.tensor(c["pc"], requires_grad=True)
    pose = torch.tensor(c["pose"], requires_grad=True)
    scale = torch.tensor(c["scale"], requires_grad=True)

    def one():
        out = rcpu.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale)
        dproj = (out["proj"].detach() - gt) / 2
        torch.autograd.grad(out["proj"], [pc, pose, scale], dproj)

    def med(thr, runs, budget, cap=8.0):
        """median of >= `runs` runs (more while `budget` seconds last) after ONE untimed run at this thread count (the
        pool is resized by set_num_threads: the first run after a switch paid 30-40 % more in round 4 and made the
        ladder disagree with the final value); a thread count whose single run takes longer than `cap` seconds (all
        256 threads of the GPU box: ~20 s per run) is timed once, that very run"""
        torch.set_num_threads(thr)
        t0 = time.perf_counter()
        one()
        first = time.perf_counter() - t0
        if first > cap:
            return first, 1
        ts, t_start = [], time.perf_counter()
        while len(ts) < runs or (time.perf_counter() - t_start < budget and len(ts) < 200):
            t0 = time.perf_counter()
            one()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), len(ts)

    one()                                       # warm-up (allocator, lazy initialisation)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    # torch's default thread count (every logical CPU) oversubscribes badly on this workload (256-thread GPU
    # box: 128 threads 5 views/s, 16 threads 55 views/s): probe a ladder, >= 3 runs each, keep the best
    ladder = sorted({t for t in (1, 4, 8, 16, 32, 64, ncpu) if t <= ncpu})
    probe = {}
    for thr in ladder:
        probe[thr] = med(thr, 3, 0.0)[0]
    best_thr = min(probe, key=probe.get)
    best, nruns = med(best_thr, 5, seconds_budget)
    probe[best_thr] = best                      # the table's entry for that count comes from these runs ...
    best_thr = min(probe, key=probe.get)        # ... and `value` is the best entry of the table, whichever count holds it now
    best = probe[best_thr]
    torch.set_num_threads(default_threads)
    return {"value": 2.0 / best, "unit": "views/s", "cores": int(best_thr), "kind": "port",
            "sample": "B=2 views of the workload's shape (N=%d, %d^3, K=%d), fwd+bwd, median of %d runs at the best "
                      "of %s torch threads (>= 3 runs each), oracle/reference_cpu.py (torch-CPU op-for-op "
                      "restatement of the TF1 graph)" % (c["N"], c["D"], c["K"], nruns, ladder),
            "value_1_thread": 2.0 / probe[1], "value_all_cores": 2.0 / probe[ncpu],
            "views_per_s_by_threads": {str(k): round(2.0 / v, 2) for k, v in sorted(probe.items())},
            "host_cpus": ncpu, "cpu_model": _cpu_model()}


def hbm_ceilings(lib, device, mbytes=512, reps=20):
    """On-box HBM ceilings, each the best of a few variants over buffers far larger than the 256 MiB
    Infinity Cache, timed with HIP events on the launch stream: `read` (float4 loads summed per work-group, 4 or 8
    in flight per lane, default or nontemporal policy; torch's own sum()), `write` (float4 fill, default or
    nontemporal; torch's fill_()), `copy` (read + write: the library's float4 copies and torch's copy_)."""
    import ctypes
    n = mbytes * (1 << 20) // 4
    src = torch.empty(n, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    partials = torch.empty(8192, dtype=torch.float32, device=device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    ck = lib.check
    groups = {
        "read": (n * 4, {"float4 x4 in flight": lambda: ck(lib.dpc_debug_read(st, P(src), n, P(partials), 4), "read"),
                         "float4 x8 in flight": lambda: ck(lib.dpc_debug_read(st, P(src), n, P(partials), 8), "read"),
                         "float4 x4, nontemporal": lambda: ck(lib.dpc_debug_read(st, P(src), n, P(partials), 104), "read"),
                         "float4 x8, nontemporal": lambda: ck(lib.dpc_debug_read(st, P(src), n, P(partials), 108), "read"),
                         "torch sum()": lambda: src.sum()}),
        "write": (n * 4, {"float4": lambda: ck(lib.dpc_debug_fill(st, P(dst), n, 1.0, 0), "fill"),
                          "float4, nontemporal": lambda: ck(lib.dpc_debug_fill(st, P(dst), n, 1.0, 100), "fill"),
                          "torch fill_()": lambda: dst.fill_(1.0)}),
        "copy": (2 * n * 4, {"k_copy<4>": lambda: ck(lib.dpc_debug_copy(st, P(src), P(dst), n, 4), "copy"),
                             "float4 x4 in flight": lambda: ck(lib.dpc_debug_copy(st, P(src), P(dst), n, 44), "copy"),
                             "float4 x8 in flight": lambda: ck(lib.dpc_debug_copy(st, P(src), P(dst), n, 48), "copy"),
                             "float4 x4, nontemporal": lambda: ck(lib.dpc_debug_copy(st, P(src), P(dst), n, 144), "copy"),
                             "float4 x8, nontemporal": lambda: ck(lib.dpc_debug_copy(st, P(src), P(dst), n, 148), "copy"),
                             "torch copy_": lambda: dst.copy_(src)}),
    }
    out = {"unit": "GB/s", "method": "%d MiB buffers, mean of %d launches per variant, HIP events; the fastest variant of "
                                     "each kind is the ceiling" % (mbytes, reps)}
    for kind, (nbytes, variants) in groups.items():
        rates = {}
        for name, call in variants.items():
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            e1.synchronize()
            rates[name] = nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
        best = max(rates, key=rates.get)
        out[kind] = rates[best]
        out[kind + "_variant"] = best
        out[kind + "_all"] = {k: round(v, 1) for k, v in rates.items()}
    return out


def mixed_ceiling(ceil, rd, wr):
    """GB/s a stream of `rd` read and `wr` written bytes can reach if reads run at the read ceiling and writes at the
    write ceiling, one after the other (the copy ceiling is that same mix at rd == wr, measured)."""
    if rd + wr == 0:
        return None
    return (rd + wr) / (rd / ceil["read"] + wr / ceil["write"])


def library_sha256(lib):
    import hashlib
    h = hashlib.sha256()
    with open(lib.path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def source_sha256():
    """sha256 over the kernel sources and the C header: ties a measurement to the CODE even where the binary differs
    by an embedded build path"""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "differentiable-point-clouds_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc")) +
                    [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "dpc_hip.h")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_traffic(lib, args, case, fname="traffic.json", script="scripts/gpu_round6.sh (section pmc)"):
    """PMC-measured bytes per launch from profiles/traffic.json -- only if they were taken on THIS build of the library
    (the file carries the sha256 of the libdpc_hip.so it was measured on; scripts/gpu_round6.sh writes both) and on
    this workload.  Returns (entry | {}, source | None, note | None).  fname="issue.json": the SQ-counter figures of
    scripts/summarize_sq.py under the same rule."""
    tpath = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(tpath):
        return {}, None, "no profiles/" + fname
    try:
        doc = json.load(open(tpath))
    except (ValueError, OSError) as e:
        return {}, None, "profiles/%s unreadable (%s)" % (fname, e)
    default_sigma = {1: 1.0, 2: 1.6, 3: 3.0, 5: 2.0}[args.config]
    key = "config%d" % args.config + ("" if args.sigma in (None, default_sigma) else "_sigma%g" % args.sigma)
    if args.k is not None or args.vox is not None or args.num_points is not None or args.points != "shell":
        return {}, None, "no PMC passes for this workload variant"
    ent = doc.get(key)
    if ent is None:
        return {}, None, "profiles/%s has no entry %r" % (fname, key)
    # Only the hash of the BINARY counts (round 4 accepted the kernel sources' hash as well, and that stamp had been edited by
    # hand after source changes): the file is written -- numbers and stamp together -- by the GPU session script on the GPU box,
    # from the library that travelled there, which is the library the driver's bench run loads.
    if doc.get("lib_sha256") != library_sha256(lib):
        return {}, None, ("profiles/%s was measured on another build of libdpc_hip.so (library sha256 %s..., this one "
                          "%s...): re-run %s and copy the file unedited"
                          % (fname, str(doc.get("lib_sha256"))[:12], library_sha256(lib)[:12], script))
    if case["B"] != ent.get("B") or case["N"] != ent.get("N", case["N"]):
        return {}, None, "profiles/%s entry %r is for another batch / point count" % (fname, key)
    return ent, "profiles/%s[%s] (rocprofv3 --pmc passes on this build, see profiles/README.md)" % (fname, key), None


def binding_bound(hbm_frac, issue):
    """Which resource binds the dominant kernel: the largest of (a) the HBM fraction -- PMC-measured bytes where they exist for
    this build, else the bytes the kernel must move -- over the 8 TB/s peak, (b) `valu_busy`, the share of the chip's vector
    issue slots the kernel fills (SQ counters), provided one of them reaches 0.5; below that nothing is saturated and the
    kernel waits -- on memory / LDS latency it cannot cover at its occupancy (wait_any_frac) or on its own dependent chains
    (wait_inst_frac): "latency".  Without SQ counters for this build the decision cannot be made: "hbm" stays (the survey's classification) with a note saying so."""
    if not issue:
        return "hbm", {"note": "unverified: no SQ counters (profiles/issue.json) for this build of the library -- 'hbm' is the "
                               "survey's classification of the path (SURVEY.md 8(d)), not a measurement"}
    cand = {"hbm": hbm_frac, "valu-issue": issue["valu_busy"]}
    top = max(cand, key=cand.get)
    why = {"hbm_frac": hbm_frac, "valu_busy": issue["valu_busy"], "rule": "max(hbm_frac, valu_busy) if it reaches 0.5, else latency"}
    return (top if cand[top] >= 0.5 else "latency"), why


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks through torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def pctl(a, q):
    return float(np.percentile(np.asarray(a, dtype=np.float64), q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 5])
    ap.add_argument("--batch", type=int, default=None,
                    help="views per GPU (config 3: models per GPU; default: the config's)")
    ap.add_argument("--points", default="shell", choices=["shell", "ball"],
                    help="synthetic cloud: noisy sphere shell (surface-like, default) or uniform ball (SURVEY.md 8(d))")
    ap.add_argument("--num-points", type=int, default=None, help="override N (projector-only workloads)")
    ap.add_argument("--sigma", type=float, default=None,
                    help="override the blur's relative sigma (the reference anneals it 3.0 -> 0.2 over a run with K fixed, "
                         "model_pc.py:33-38; the library runs the tap count the sigma still needs)")
    ap.add_argument("--k", type=int, default=None, help="override the Gaussian kernel size (projector-only workloads)")
    ap.add_argument("--vox", type=int, default=None, help="override the grid size vox_size (projector-only workloads)")
    ap.add_argument("--projector-only", action="store_true", help="config 3: time the projector alone at the training shape")
    ap.add_argument("--keep-prob", type=float, default=1.0, help="config 3: point dropout keep probability (N = 8000 * keep)")
    ap.add_argument("--graph", action="store_true",
                    help="record one step into a HIP graph and replay it: the default for the projector workloads (the "
                         "library only enqueues on the stream it is handed; a 7-launch step is otherwise at the mercy "
                         "of the host's enqueue rate), opt-in for the training step (--config 3, single GPU)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step eagerly from Python")
    ap.add_argument("--repeats", type=int, default=None,
                    help="extra HIP-event-timed blocks of --steps steps for median/p10/p90 (default: >= 10, >= 1 s)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): the config's batch PER GPU; strong: the config's batch split over the GPUs "
                         "(SURVEY.md 8(e): secondary, latency-bound at 8 GPUs)")
    ap.add_argument("--burn-in", type=float, default=0.3,
                    help="seconds of untimed steps BEFORE the --warmup steps: the GPU's clocks take ~0.1 s of continuous load to "
                         "settle after the idle set-up phase (profiles/r03/startup_probe.txt); without it a 20-step timed block "
                         "runs 4-8 %% below the steady state the `timing` blocks see.  0 = off.  Reported in config.burn_in_s")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: create the process group anyway (backend nccl = RCCL, ONE rank) and take the "
                         "distributed code path -- RCCL communicator + watchdog, barriers and the MAX all-reduce through "
                         "ProcessGroupNCCL, --config 3: DDP's reducer, --config 3 --graph: GradBuckets' bucket all-reduces "
                         "recorded into the HIP graph.  The one-GPU rehearsal of the 2/4/8-GPU runs (no bytes cross xGMI)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    dd = dpc_amd.distributed
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.force_dist or os.environ.get("DPC_FORCE_DIST") == "1":
        args.force_dist = True
        os.environ["DPC_FORCE_DIST"] = "1"
        if "WORLD_SIZE" not in os.environ:            # a one-rank "launch": what torch.distributed.run would export
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    rank, _, world = dd.env_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if DRY_RUN:
        # launch / rendezvous / reporting logic on a GPU-less host (tests/test_bench_launch.py): the kernels'
        # CPU emulation build, gloo, toy sizes.  Refused unless the test hooks are on; never a measurement.
        emu = dpc_amd._capi.DpcLibrary(os.path.join(ROOT, "tests", "hipemu", "libdpc_emu.so"), host_memory=True)
        dpc_amd._capi.set_library(emu)
        torch.set_num_threads(1)
        rank, world, device = dd.init("gloo", device=torch.device("cpu"))
        dpc_amd.synthetic.CONFIGS[args.config] = dict(B=2, N=150, D=32, K=5, sigma=0.9)
        args.steps, args.warmup, args.repeats, args.no_cpu_baseline = min(args.steps, 2), min(args.warmup, 1), 0, True
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (the projector has no CPU fallback)")
        if os.environ.get("DPC_BENCH_SHARE_GPU") == "1":
            # dev hook: every rank on cuda:0 with gloo as the rendezvous -- exercises the launcher, the per-rank graph
            # capture, the barriers and the reporting on a ONE-GPU box (RCCL refuses two ranks on one device).
            # The number it prints is NOT a measurement (the ranks share the GPU).
            torch.cuda.set_device(0)
            rank, world, device = dd.init("gloo", device=torch.device("cuda", 0))
        else:
            rank, world, device = dd.init("nccl")

    lib = dpc_amd.get_library()
    train = args.config == 3 and not args.projector_only
    if args.graph and args.no_graph:
        raise SystemExit("--graph and --no-graph exclude each other")
    explicit_graph = bool(args.graph)
    if not train and not args.no_graph and not DRY_RUN:
        args.graph = True
    if os.environ.get("DPC_CUDNN_BENCHMARK"):     # dev switch: MIOpen find mode for the stock PyTorch layers
        torch.backends.cudnn.benchmark = os.environ["DPC_CUDNN_BENCHMARK"] == "1"
    if train:
        if args.scaling == "strong" and world > 1:
            total = args.batch or 16
            lo, hi = dd.shard_range(total, rank, world)
            if (hi - lo) * world != total:
                raise SystemExit("--scaling strong: %d models do not split evenly over %d ranks" % (total, world))
            args.batch = hi - lo
        case = build_train_case(args, device, rank, world)
        run = case["run"]
    else:
        batch = args.batch
        if args.scaling == "strong" and world > 1:
            total = batch or dpc_amd.synthetic.CONFIGS[args.config]["B"]
            lo, hi = dd.shard_range(total, rank, world)          # this rank's contiguous slice of the global batch
            if hi - lo == 0:
                raise SystemExit("--scaling strong: %d views do not split over %d ranks" % (total, world))
            batch, args.global_views = hi - lo, total
        case = build_case(args.config, batch, device, seed_offset=1000 * rank, kind=args.points, N=args.num_points,
                          sigma=args.sigma, K=args.k, D=args.vox)
        run = lambda: step(case)
    trace("case built")
    dist_on = dd.active()       # a process group exists (several ranks, or --force-dist): the distributed code path
    numa_note = dd.bind_to_gpu_numa(device) if (dist_on and not DRY_RUN) else None
    graph_note = None
    if args.graph and train and DRY_RUN:
        args.graph, graph_note = False, "dry run: the recordable reducer (GradBuckets) runs eagerly under gloo"
    if args.graph and train:
        # The training step is ~250 launches (stock PyTorch layers, fused optimiser, the library's kernels): eager,
        # the host sets the pace.  The whole step -- nets, projector, loss epilogue, backward, (N > 1: the bucketed RCCL
        # all-reduce of the gradients, issued from gradient hooks and overlapped with the backward pass,) Adam -- is
        # recorded into ONE hipGraph per rank and replayed (dpc_amd.graphs.RecordedStep: the shared recipe -- warm-up
        # count, barrier before the capture, thread-local capture mode, re-record when the blur's tap count moves).
        try:
            recorded = dpc_amd.graphs.RecordedStep(case["run"], world=world, device=device, collectives=dist_on,
                                                   key=case["projector"].recording_key)
            run = recorded
        except Exception as e:                       # noqa: BLE001
            # A capture that dies half way leaves the rank's streams in capture mode (seen with gloo on CUDA tensors,
            # whose collectives fork streams that never join: hipErrorStreamCaptureUnjoined) -- there is no clean way on
            # from here, and the other ranks are waiting in a collective: stop loudly instead of limping on.
            sys.stderr.write("[rank %d] recording the training step failed (%s: %s); run without --graph for the eager "
                             "DDP step\n" % (rank, type(e).__name__, e))
            raise
    if args.graph and not train:
        # the library only enqueues on the stream it is handed, so a whole step (forward with the loss
        # gradient, backward) records into one hipGraph; replay costs one launch on the host.  Capture is
        # per rank and thread-local (an RCCL watchdog thread may touch the runtime meanwhile); if it fails on
        # some runtime the run goes on eagerly and says so in the line.
        try:
            # (no collective inside this step: world = 1 as far as the recording is concerned)
            run = dpc_amd.graphs.RecordedStep(lambda: step(case), world=1, device=device)
        except Exception as e:                       # noqa: BLE001 -- report, do not lose the measurement
            if explicit_graph:
                raise
            torch.cuda.synchronize()
            args.graph, graph_note = False, "HIP graph capture failed (%s: %s); eager launches" % (type(e).__name__, e)
            sys.stderr.write("[rank %d] %s\n" % (rank, graph_note))
            run = lambda: step(case)
    trace("step recorded" if args.graph else "eager step")
    if args.burn_in > 0 and not DRY_RUN:       # clocks up (see --burn-in); untimed, before the contract's warm-up steps
        if train and dist_on:
            # every step holds collectives: all ranks must run the SAME number of steps -- a count, not a clock
            for _ in range(max(8, int(args.burn_in / 0.004))):
                run()
            torch.cuda.synchronize()
        else:
            t_burn = time.perf_counter()
            while time.perf_counter() - t_burn < args.burn_in:
                for _ in range(8):
                    run()
                torch.cuda.synchronize()
    if DRY_RUN and os.environ.get("DPC_BENCH_TEST_SLEEP_LAST_RANK") and rank == world - 1:
        # test hook (dry run only): the last rank is slower by that many seconds per step -- the reported time must be ITS
        nap, fast = float(os.environ["DPC_BENCH_TEST_SLEEP_LAST_RANK"]), run
        run = lambda: (time.sleep(nap), fast())[1]
    trace("burn-in done")
    for _ in range(args.warmup):
        run()
    dd.barrier(device)                      # barrier + torch.cuda.synchronize() on both sides
    trace("warm-up done, timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    dd.barrier(device)
    elapsed = dd.max_over_ranks(time.perf_counter() - t0, device)
    ms_step = elapsed / args.steps * 1e3
    trace("timed region done: %.3f ms/step" % ms_step)

    # ---- spread: further blocks of `steps` steps, HIP events on the launch stream, every rank in step ----
    repeats = args.repeats
    if repeats is None:
        repeats = int(min(200, max(10, np.ceil(1000.0 / max(ms_step * args.steps, 1e-3)))))
    blocks = []
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            run()
        e1.record()
        e1.synchronize()
        blocks.append(e0.elapsed_time(e1) / args.steps)
    if dist_on:
        dd.barrier(device)

    trace("spread blocks done")
    # ---- per-kernel durations (HIP events on the launch stream), rank 0 ----------
    roof = None
    psteps = max(3, min(args.steps, 20))
    if not DRY_RUN:                         # every rank steps (DDP's all-reduce needs them all); rank 0 records
        if rank == 0:
            lib.profile(True)
        for _ in range(psteps):                 # (eager also in --graph mode: the per-kernel events need real launches)
            case["run"]() if train else step(case)
        torch.cuda.synchronize()
    if rank == 0 and not DRY_RUN:
        recs = lib.profile_records()
        lib.profile(False)
        agg = {}
        for label, ms in recs:
            a = agg.setdefault(label, [0, 0.0])
            a[0] += 1
            a[1] += ms
        per_step = {k: v[1] / psteps for k, v in agg.items()}
        dom = max(agg, key=lambda k: agg[k][1])
        dom_ms = agg[dom][1] / agg[dom][0]
        k_run = case["K"]
        if train:
            k_run = case["projector"].effective_tap_counts()[2]
        elif case.get("kern") is not None:
            k_run = dpc_amd.util.point_cloud.effective_tap_counts(case["cfg"], case["kern"])[2]
        save_xy = lib.saves_xy(case["B"], case["N"], case["D"], k_run)
        # the bytes the kernels must move for this run's clouds (projector workloads: occupied planes counted from
        # the clouds themselves; training step: the dense model, the clouds are the decoder's output)
        fused_path = dpc_amd.ops.uses_fused_path(lib, case["B"], case["N"], dpc_amd.util.point_cloud._meta(
            dpc_amd.default_config(vox_size=case["D"])), (k_run,) * 3)
        cs_on = bool(fused_path) and lib.chunk_sparse(case["B"], case["N"], case["D"], k_run)
        occ = None if (train or not fused_path) else plane_occupancy(case, k_run, cs_on)
        by_kernel = {k: kernel_bytes(k, case, save_xy, occ) for k in per_step}
        rd, wr = by_kernel[dom]
        launches_per_step = agg[dom][0] / psteps
        impl_step = sum(a + b for a, b in by_kernel.values())
        ent, tsrc, tnote = pmc_traffic(lib, args, case) if not train else ({}, None, "training step: projector traffic not taken")
        traffic = ent.get(dom)
        step_traffic = ent.get("_step_total")
        ient, isrc, inote = (pmc_traffic(lib, args, case, "issue.json", "scripts/gpu_round6.sh (section sq)") if not train
                             else ({}, None, "training step: SQ counters not taken"))
        issue = ient.get(dom) if isinstance(ient.get(dom), dict) else None
        proj_ms = sum(per_step.values()) if train else ms_step     # config 3: the projector's share of the step
        ceil = hbm_ceilings(lib, device)
        ach = (rd + wr) / (dom_ms * 1e-3) / 1e9
        kceil = mixed_ceiling(ceil, rd, wr)
        srd, swr = sum(a for a, _ in by_kernel.values()), sum(b for _, b in by_kernel.values())
        sceil = mixed_ceiling(ceil, srd, swr)
        V = 4 * case["D"] ** 3
        model_step = dpc_amd.synthetic.algorithmic_bytes_per_view(case["N"], case["D"], case["D"]) * case["B"]
        model_kernel = {"zfwd": 1, "zbwd": 2, "splat_xy": 1, "gather_yx": 1}.get(dom)
        hbm_frac = (traffic if traffic is not None else rd + wr) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        bound, bound_why = binding_bound(hbm_frac, issue)
        roof = {"bound": bound, "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                # the instruction-issue side of the same kernel (SQ counters of THIS build): `frac` says how far the kernel is from
                # the HBM roof, issue.valu_busy how far from the vector-issue roof -- `bound` names the larger (binding_bound)
                "issue": issue, "issue_source": isrc, **({"issue_note": inote} if inote else {}),
                **({"bound_basis": bound_why} if bound_why else {}),
                "traffic": traffic, "traffic_source": tsrc, **({"traffic_note": tnote} if tnote else {}),
                "measured_achieved": None if traffic is None else traffic / (dom_ms * 1e-3) / 1e9,
                "kernel_ms": dom_ms, "kernel_launches_per_step": launches_per_step,
                "kernel_bytes": {"read": rd, "written": wr,
                                 "basis": (("occupied planes of this run's clouds (%d of %d)%s + point records + images"
                                            % (occ[0], case["B"] * case["D"],
                                               (", of which the marked 128-byte chunks (%d of %d: chunk-sparse layout; G2: %d) and their marks"
                                                % (occ[2], case["B"] * case["D"] ** 3 // 32, occ[3])) if occ[2] is not None else ""))
                                           if occ else "dense planes + point records + images")},
                "chunk_sparse": bool(occ and occ[2] is not None),
                "ceilings": ceil, "kernel_ceiling": kceil, "vs_ceiling": None if not kceil else ach / kceil,
                "taps_run": k_run, "saves_xy_grid": bool(save_xy),
                "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items())},
                "kernel_ms_note": "HIP events around every launch of an EAGER pass (not the replayed graph the step time comes "
                                  "from): each carries ~2 us of event overhead and no overlap with its neighbours, so their sum "
                                  "may exceed step_ms; the rocprofv3 kernel statistics under profiles/ are the tighter figures",
                "kernel_bytes_per_step": {k: a + b for k, (a, b) in sorted(by_kernel.items())},
                "step_ms": proj_ms,
                "step_scope": ("library kernels only (sum of their HIP-event durations inside the training step)"
                               if train else "whole timed step"),
                "step_bytes": impl_step, "step_achieved": impl_step / (proj_ms * 1e-3) / 1e9,
                "step_frac": impl_step / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "step_ceiling": sceil, "step_vs_ceiling": None if not sceil else impl_step / (proj_ms * 1e-3) / 1e9 / sceil,
                "step_measured_bytes": step_traffic,
                "step_measured_achieved": None if step_traffic is None else step_traffic / (proj_ms * 1e-3) / 1e9,
                "model": {"what": "SURVEY.md 8(d) stage model: 8 V + P bytes per view, each stage touching each dense grid once "
                                  "-- what an UNFUSED pipeline must move; this one moves 5 V (<= 11 taps) or 6 V and skips "
                                  "empty planes, so these rates may exceed the peak and are not roofline fractions",
                          "step_bytes": model_step, "step_rate": model_step / (proj_ms * 1e-3) / 1e9,
                          "step_rate_over_peak": model_step / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "kernel_bytes": None if model_kernel is None else model_kernel * V * case["B"],
                          "kernel_rate": None if model_kernel is None else model_kernel * V * case["B"] / (dom_ms * 1e-3) / 1e9},
                "which_is_which": "achieved / frac / step_*: bytes this implementation must move (kernel_bytes) over HIP-event "
                                  "time, against the 8 TB/s spec peak; vs_ceiling: against this run's measured read / write "
                                  "streams mixed like the kernel's bytes; traffic / measured_*: PMC bytes (rocprofv3) of this "
                                  "very build, else null; model.*: the survey's unfused 8 V model, for reference only"}

    if rank == 0:
        global_views = getattr(args, "global_views", None) or world * case["B"]    # (strong scaling: uneven slices add up)
        views = global_views * args.steps
        cfg_idx = {1: 0, 2: 1, 3: 2 if world == 1 else 3, 5: 4}[args.config]
        if train:
            workload = ("BASELINE.json configs[%d]: chair_unsupervised training step (encoder+decoder+pose nets, stock "
                        "PyTorch; HIP projector + silhouette-loss epilogue; Adam), %d models x %d views x %d pose "
                        "candidates = %d views per GPU of N=%d pts -> %d^3, K=%d, sigma=%.1f, %.1f M parameters%s"
                        % (cfg_idx, case["models"], case["views_per_model"], case["candidates"], case["B"], case["N"], case["D"], case["K"], case["sigma"],
                           case["params"] / 1e6, ", %s (RCCL all-reduce)" % case["reducer"] if case.get("reducer") else ""))
        else:
            workload = ("BASELINE.json configs[%d]%s: pointcloud_project_fast fwd+bwd, N=%d, grid %d^3, K=%d, sigma=%.1f, "
                        "batch %d views per GPU, %s point clouds, dproj=(proj-gt)/B"
                        % (cfg_idx, " (projector only)" if args.config == 3 else "", case["N"], case["D"], case["K"],
                           case["sigma"], case["B"], args.points))
        line = {
            "metric": "projected views/sec (fwd+bwd), %d pts->%d^3->%d^2 at bs=%d per GPU"
                      % (case["N"], case["D"], case["D"], case["B"]),
            "value": views / elapsed, "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": ("dry run on the CPU emulation library: NOT a measurement" if DRY_RUN else
                     "ranks share one GPU (DPC_BENCH_SHARE_GPU): NOT a measurement"
                     if os.environ.get("DPC_BENCH_SHARE_GPU") == "1" else "synthetic"),
            "config": {"workload": workload, "global_batch": global_views, "K": case["K"], "sigma": case["sigma"],
                       "taps_run": (roof or {}).get("taps_run"),
                       "hip_graph": bool(args.graph), **({"hip_graph_note": graph_note} if graph_note else {}),
                       "burn_in_s": 0.0 if DRY_RUN else args.burn_in,
                       "training_step": bool(train),
                       "parallelism": ((("models sharded x%d (%s), gradient all-reduce over RCCL" % (world, case["reducer"]))
                                        if case.get("reducer") else "one process, no reducer") if train else
                                       ("views sharded x%d, no data-path collective" % world))
                                      + ("" if not dist_on else " [%s%s%s]" % (dd.collective_library() or "no process group",
                                                                              ", forced one-rank group" if args.force_dist and world == 1 else "",
                                                                              "; rank 0: " + numa_note if numa_note else ""))},
            "timing": None if not blocks else {
                "repeats": len(blocks), "steps_per_block": args.steps, "clock": "HIP events, rank 0",
                "ms_per_step_median": pctl(blocks, 50), "ms_per_step_p10": pctl(blocks, 10),
                "ms_per_step_p90": pctl(blocks, 90),
                "value_median": global_views / (pctl(blocks, 50) * 1e-3)},
            "roofline": roof,
        }
        if train:
            line["steps_per_s"] = args.steps / elapsed
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    dd.finalize()


if __name__ == "__main__":
    main()
