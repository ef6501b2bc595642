"""Round-3 additions to the CPU tier: statistics of the fused dropout's keyed permutation (inclusion and
pair-inclusion frequencies over 10^4 (seed, instance) keys at the training shape's point counts), and the same
white-box read of the draw through the C ABI that the GPU tier uses, on the emulation library."""
import numpy as np
import pytest

import parity_cases
from helpers import synth


@pytest.mark.parametrize("keep", [560, 4000])      # pc_point_dropout 0.07 and 0.5 of 8000 points
def test_dropout_permutation_statistics(keep):
    """dpc/util/point_cloud.py:293-319 draws int(N keep_prob) of N points uniformly without replacement,
    independently per instance and step.  The keyed Feistel permutation that replaces np.random.choice must look
    the same to first and second order: every point is kept with probability keep/N, every pair with
    keep (keep-1) / (N (N-1)) -- chi-squares over 10 240 keys (32 seeds x 320 instances) within 5 sigma, no
    single z-score beyond 6."""
    from oracle import dropout_ref
    N = 8000
    masks = np.stack([dropout_ref.dropout_rank(N, seed, b) < np.uint64(keep) for seed in range(1000, 1032)
                      for b in range(320)])
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
