"""Round-2 parity cases, CPU tier (kernel emulation) -- the GPU tier repeats them in test_gpu_parity.py:
asymmetric filters (adjoint passes use reversed taps), fused point dropout, knife-edge inputs without
nudging, elementwise gradient tolerance, argument validation added after the round-1 review."""
import numpy as np
import pytest
import torch

import dpc_amd
import parity_cases
from helpers import close_elementwise, load
from run_case import run_product


@pytest.mark.parametrize("D,K", parity_cases.ASYM_CASES)
def test_emu_asymmetric_filters(emu, D, K):
    parity_cases.asymmetric_filters_against_cpu_oracle("cpu", D, K)


def test_emu_student_loss(emu):
    parity_cases.student_loss_equals_the_quaternion_composite("cpu")
    parity_cases.student_loss_equals_the_quaternion_composite("cpu", n=300, C=2, seed=9)      # more samples than threads


def test_emu_fused_l2_epilogue(emu):
    parity_cases.fused_l2_epilogue_equals_the_autograd_loss("cpu")
    parity_cases.fused_l2_epilogue_equals_the_autograd_loss("cpu", B=2, N=100, D=16, K=5)      # generic path: same kernel tail


def test_emu_fused_dropout(emu):
    parity_cases.fused_dropout_equals_explicit_subset("cpu")
    # the device permutation against the oracle's at awkward sizes (2^k +- 1, tiny)
    for N, keep in ((65, 7), (129, 128), (33, 1), (5, 2)):
        parity_cases.fused_dropout_equals_explicit_subset("cpu", B=2, N=N, D=32, K=5, keep=keep, seed=1234 + N, extras=False)


@pytest.mark.parametrize("D,Dz", [(32, 33), (33, 33)])      # fused path / generic path (D-1 = 32: exact interior faces)
def test_emu_knife_edges(emu, D, Dz):
    parity_cases.knife_edge_inputs_match_reference_conventions("cpu", D, Dz)


@pytest.mark.parametrize("name", ["tiny", "tiny_focal", "k21", "tiny_matrix"])
def test_emu_point_gradients_elementwise(emu, name):
    """dpc against the fp64 goldens entry by entry (|err| <= 2e-5 max|ref| + 1e-3 |ref|), not max-abs / max-magnitude."""
    g = load(name)
    _, gr = run_product(name, g, "cpu", grads=True)
    ok, worst = close_elementwise(gr["dpc"], g["dpc_f64"])
    assert ok, worst


def test_emu_deep_grid_takes_the_generic_path(emu):
    parity_cases.deep_grid_takes_the_generic_path("cpu", D=32, Dz=288)


def test_emu_degenerate_clouds(emu):
    parity_cases.degenerate_clouds_against_numpy_oracle("cpu", heavy=False)


def test_dropout_reference_permutation_properties():
    from oracle import dropout_ref
    for N in (7, 64, 1000, 8000):
        r = dropout_ref.dropout_rank(N, 99, 3)
        assert sorted(r.tolist()) == list(range(N))
    m = dropout_ref.kept_mask(16, 8000, 560, 5)
    assert (m.sum(1) == 560).all()
    # marginal keep rate per point over instances ~ 560/8000 (binomial, 16 draws): nothing systematically kept
    assert m.mean(0).max() <= 7 / 16


def test_dropout_permutation_is_a_bijection_for_any_size_and_key():
    """Property (hypothesis): the keyed unbalanced-Feistel permutation with cycle walking is a bijection of [0, N)
    for every N (powers of two, their neighbours, primes, 1), seed and instance; different instances of one seed
    get different permutations."""
    from hypothesis import given, settings, strategies as st
    from oracle import dropout_ref

    @settings(max_examples=60, deadline=None)
    @given(st.one_of(st.integers(1, 70), st.sampled_from([127, 128, 129, 1023, 1024, 1025, 4093, 8000, 8191, 8192, 8193, 16000])),
           st.integers(0, 2 ** 32 - 1), st.integers(0, 4095))
    def check(N, seed, b):
        r = dropout_ref.dropout_rank(N, seed, b)
        assert r.shape == (N,) and np.array_equal(np.sort(r), np.arange(N, dtype=r.dtype))
        if N >= 64:
            assert not np.array_equal(r, dropout_ref.dropout_rank(N, seed, b + 1))
    check()


def test_matrix_pose_focal_has_no_gradient(emu):
    """The matrix branch ignores focal_length (camera.py:5-13): its gradient is None, not uninitialised memory."""
    g = load("tiny_matrix")
    cfg = dpc_amd.default_config(vox_size=int(g["D"]), pc_gauss_kernel_size=int(g["K"]), pose_quaternion=False)
    pc = torch.tensor(g["pc"], requires_grad=True)
    pose = torch.tensor(g["pose"], requires_grad=True)
    focal = torch.full((pc.shape[0], 1), 1.9, requires_grad=True)
    out = dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, dpc_amd.smoothing_kernel(cfg, float(g["sigma"]), device="cpu"),
                                          focal_length=focal)
    gr = torch.autograd.grad(out["proj"].sum(), [pc, focal], allow_unused=True)
    assert gr[1] is None and torch.isfinite(gr[0]).all()
    tr = dpc_amd.pc_perspective_transform(cfg, pc, pose, None, focal)
    gr = torch.autograd.grad(tr.sum(), [pc, focal], allow_unused=True)
    assert gr[1] is None


def test_silhouette_loss_argument_validation(emu):
    from dpc_amd import ops
    proj = torch.rand(8, 16, 16, 1, requires_grad=True)
    gt_u8 = (torch.rand(2, 32, 32, 1) > 0.5).to(torch.uint8)
    loss, winners, _ = ops.SilhouetteLoss.apply(proj, gt_u8, None, 4)           # uint8 masks are converted, not misread
    ref, _, _ = ops.SilhouetteLoss.apply(proj, gt_u8.float(), None, 4)
    assert float(loss) == float(ref) and winners.shape == (2,)
    with pytest.raises(ValueError):
        ops.SilhouetteLoss.apply(proj, torch.rand(3, 32, 32, 1), None, 4)       # wrong number of masks
    with pytest.raises(ValueError):
        ops.SilhouetteLoss.apply(proj, torch.rand(2, 32, 32, 1), torch.ones(5), 4)
    with pytest.raises(ValueError):
        ops.SilhouetteLoss.apply(proj, torch.rand(2, 8, 8, 1), None, 4)         # GT smaller than the prediction
    with pytest.raises(ValueError):
        ops.SilhouetteLoss.apply(proj, torch.rand(2, 32, 32, 1), None, 3)       # 8 instances / 3 candidates


def test_fused_dropout_needs_the_fused_path(emu):
    cfg = dpc_amd.default_config(vox_size=18, pc_gauss_kernel_size=5)
    pc = torch.zeros(1, 50, 3)
    q = torch.tensor([[1.0, 0, 0, 0]])
    with pytest.raises(ValueError, match="fused"):
        dpc_amd.pointcloud_project_fast(cfg, pc, q, None, None, dpc_amd.smoothing_kernel(cfg, 1.0, device="cpu"),
                                        point_dropout=(10, 1))
