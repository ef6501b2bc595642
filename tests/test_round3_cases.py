"""Round-3 additions to the CPU tier: statistics of the fused dropout's keyed permutation (inclusion and
pair-inclusion frequencies over 10^4 (seed, instance) keys at the training shape's point counts), and the same
white-box read of the draw through the C ABI that the GPU tier uses, on the emulation library."""
import numpy as np
import pytest

import parity_cases
from helpers import synth


_RANKS = {}


def _ranks_of_10240_keys(N):
    """the permutation's ranks for 32 seeds x 320 instances (they do not depend on `keep`: computed once for both cases)"""
    from oracle import dropout_ref
    if N not in _RANKS:
        _RANKS[N] = np.stack([dropout_ref.dropout_rank(N, seed, b) for seed in range(1000, 1032) for b in range(320)])
    return _RANKS[N]


@pytest.mark.parametrize("keep", [560, 4000])      # pc_point_dropout 0.07 and 0.5 of 8000 points
def test_dropout_permutation_statistics(keep):
    """dpc/util/point_cloud.py:293-319 draws int(N keep_prob) of N points uniformly without replacement,
    independently per instance and step.  The keyed Feistel permutation that replaces np.random.choice must look
    the same to first and second order: every point is kept with probability keep/N, every pair with
    keep (keep-1) / (N (N-1)) -- chi-squares over 10 240 keys (32 seeds x 320 instances) within 5 sigma, no
    single z-score beyond 6."""
    N = 8000
    masks = _ranks_of_10240_keys(N) < np.uint64(keep)
    st = parity_cases.dropout_statistics(masks, keep, parity_cases.dropout_pairs(N, np.random.default_rng(7)))
    parity_cases.assert_dropout_statistics(st)
    # neighbouring instances and neighbouring seeds overlap like independent draws: mean |A & B| = keep^2 / N
    m = masks.reshape(32, 320, N)
    e, sd = keep * keep / N, np.sqrt(keep * (keep / N) * (1 - keep / N) * (N - keep) / (N - 1))
    for ov in ((m[:, :-1] & m[:, 1:]).sum(-1), (m[:-1] & m[1:]).sum(-1)):
        assert abs(ov.mean() - e) < 5.0 * sd / np.sqrt(ov.size), (ov.mean(), e)


def test_emu_dropout_draw_read_back_through_the_c_abi(emu):
    """The kernels' draw (decoded from point_index after dpc_project_forward) equals oracle/dropout_ref.py."""
    from oracle import dropout_ref
    B, N, D, K, keep, seed = 3, 500, 32, 5, 137, 4242
    inp = synth.make_inputs(B, N, 77)
    pc = (0.25 * inp["pc"] / np.maximum(np.abs(inp["pc"]).max(), 1e-6)).astype(np.float32)    # |p| <= 0.44: inside the cube under any rotation
    got = parity_cases.fused_dropout_kept_mask_from_library("cpu", pc, inp["pose"], D, K, keep, seed)
    assert np.array_equal(got, dropout_ref.kept_mask(B, N, keep, seed))


def test_emu_views_per_cloud_equals_explicit_replication(emu):
    """pointcloud_project_fast(views_per_cloud=R) on [B/R,N,3] clouds == the same call on the tf_repeat_0 copies
    (model_pc.py:23-32,270-279): images bit for bit, per-instance gradients alike, and the point gradient equal to the
    sum over each cloud's R instances."""
    import torch
    import dpc_amd
    C, R, N, D, K = 2, 3, 80, 32, 5
    inp = synth.make_inputs(C * R, N, 321)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, 0.9, device="cpu")
    t = lambda a: torch.tensor(a, requires_grad=True)
    clouds = t(inp["pc"][::R].copy())                       # C clouds
    pose, scale = t(inp["pose"]), t(inp["scale"])
    w = torch.tensor(np.random.default_rng(2).standard_normal((C * R, D, D, 1)).astype(np.float32))
    a = dpc_amd.pointcloud_project_fast(cfg, clouds, pose, None, None, kern, scaling_factor=scale, views_per_cloud=R)
    ga = torch.autograd.grad(a["proj"], [clouds, pose, scale], w)
    clouds2, pose2, scale2 = t(inp["pc"][::R].copy()), t(inp["pose"]), t(inp["scale"])
    rep = torch.repeat_interleave(clouds2, R, dim=0)
    b = dpc_amd.pointcloud_project_fast(cfg, rep, pose2, None, None, kern, scaling_factor=scale2)
    gb = torch.autograd.grad(b["proj"], [clouds2, pose2, scale2], w)
    assert a["proj"].shape == (C * R, D, D, 1) and a["tr_pc"].shape == (C * R, N, 3)
    assert float((a["proj"] - b["proj"]).abs().max()) == 0.0 and float((a["tr_pc"] - b["tr_pc"]).abs().max()) == 0.0
    assert ga[0].shape == (C, N, 3)
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= 1e-6 * max(1.0, float(y.abs().max()))
    with pytest.raises(ValueError):
        dpc_amd.pointcloud_project_fast(cfg, clouds, pose[:4], None, None, kern, views_per_cloud=R)


def test_emu_model_replicates_inside_the_kernels(emu):
    """ModelPointCloud.replicate_outputs leaves `all_points` unbuilt; compute_projection hands points_1 and the
    replication factor to the projector; the result equals the explicit path (pc_replicate_in_kernel=False)."""
    import torch
    import dpc_amd
    res = {}
    for in_kernel in (True, False):
        cfg = dpc_amd.default_config(vox_size=32, pc_gauss_kernel_size=5, pc_num_points=60, predict_pose=True,
                                     pose_predict_num_candidates=2, step_size=2, batch_size=2, pc_point_dropout=1.0,
                                     pc_replicate_in_kernel=in_kernel)
        m = dpc_amd.model_pc.ModelPointCloud(cfg, global_step=0, device="cpu")
        inp = synth.make_inputs(2, 60, 5)
        pts = torch.tensor(inp["pc"], requires_grad=True)
        poses = torch.tensor(synth.make_inputs(8, 4, 6)["pose"])
        outputs = m.replicate_outputs({"points_1": pts, "poses": poses, "scaling_factor": torch.full((2, 1), 0.8)})
        assert outputs.points_replication() is not None and outputs.points_replication()[1] == 4
        outputs = m.compute_projection({}, outputs, is_training=True)
        assert (outputs.points_replication() is not None) == in_kernel       # the copies were only built on the explicit path
        g = torch.autograd.grad(outputs["projs"].sum(), [pts])[0]
        res[in_kernel] = (outputs["projs"].detach(), g)
        assert outputs["all_points"].shape == (8, 60, 3)                    # still there for whoever asks
    assert float((res[True][0] - res[False][0]).abs().max()) == 0.0
    assert float((res[True][1] - res[False][1]).abs().max()) <= 1e-6 * float(res[False][1].abs().max())


@pytest.mark.parametrize("kw", [dict(N=100), dict(B=4, C=1, N=100, with_valid=False), dict(B=6, C=2, N=100, S=32, rep=3)])
def test_emu_fused_candidate_loss(emu, kw):
    parity_cases.fused_candidate_loss_equals_the_image_epilogue("cpu", **kw)


def test_emu_dense_gather_flow(emu):
    """a one-strip plane with more than 128 points: k_gather_yx blurs whole rows (x in registers while staging, y from
    the LDS tile) and every point reads its 2 x 2 cells -- against the float64 NumPy oracle"""
    parity_cases.fused_path_against_numpy_oracle("cpu", *parity_cases.DENSE_GATHER_CASE_EMU)


def test_grad_buckets_partition_and_views():
    """GradBuckets on one rank: every parameter's .grad is a view into exactly one flat bucket (reverse parameter order,
    size limit respected, strides of the parameter kept), backward accumulates into the buckets in place, finish()
    re-arms the countdown, zero_() clears what the next step accumulates into."""
    import torch
    import dpc_amd
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.LeakyReLU(), torch.nn.Conv2d(8, 4, 3), torch.nn.Flatten(),
                              torch.nn.Linear(4 * 4 * 4, 16), torch.nn.Linear(16, 2))
    net = net.to(memory_format=torch.channels_last)
    params = list(net.parameters())
    red = dpc_amd.distributed.GradBuckets(params, bucket_mb=2048 / (1 << 20))        # 2 KiB buckets: several of them
    assert len(red.buckets) >= 3
    seen = []
    for flat, plist in red.buckets:
        assert flat.numel() == sum(p.numel() for p in plist)
        for p in plist:
            assert p.grad is not None and p.grad.shape == p.shape and p.grad.stride() == p.stride()
            assert p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
            seen.append(p)
    assert [id(p) for p in seen] == [id(p) for p in reversed(params)]                  # reverse order, each exactly once
    x = torch.randn(5, 3, 8, 8)
    ref = torch.autograd.grad(net(x).square().sum(), params)
    for step in range(2):                                                              # second step: after zero_()
        net(x).square().sum().backward()
        red.finish()
        for p, g in zip(params, ref):
            assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-7)
        assert all(n == len(pl) for n, (_, pl) in zip(red._pending, red.buckets))      # re-armed
        red.zero_()
        assert all(float(p.grad.abs().max()) == 0.0 for p in params)


def test_replicated_outputs_is_lazy_and_dict_like():
    """ModelPointCloud.replicate_outputs: `all_points` is built on first read only, exactly as tf_repeat_0 would, and
    an explicit assignment (a caller replacing the clouds) wins over the lazy value."""
    import torch
    import dpc_amd
    mp = dpc_amd.model_pc
    pts = torch.arange(2 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3)
    out = mp.ReplicatedOutputs({"points_1": pts, "x": 1}, 3)
    assert out.points_replication() is not None and out.points_replication()[1] == 3 and "all_points" in out
    assert out["x"] == 1 and out.points_replication() is not None                       # other keys do not materialise it
    ap = out["all_points"]
    assert out.points_replication() is None and torch.equal(ap, mp.tf_repeat_0(pts, 3)) and out["all_points"] is ap
    out2 = mp.ReplicatedOutputs({"points_1": pts}, 2)
    out2["all_points"] = pts[:1]
    assert out2.points_replication() is None and out2.get("all_points").shape == (1, 3, 3)
    out3 = mp.ReplicatedOutputs({"points_1": pts}, 2)
    assert dict(out3.items())["all_points"].shape == (4, 3, 3)                          # items() / values() materialise
