"""Tier 1: pin the two CPU restatements in oracle/ against the golden vectors
generated from the reference's own source (tests/golden/make_goldens.py), and
check analytic known answers.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import (ALL_CASES, case_params, load, maxabs, onp, rcpu, relerr,
                     run_numpy_oracle, run_reference_cpu)

GRAD_KEYS = ("dpc", "dpose", "dtrans", "dscale", "dfocal")


@pytest.mark.parametrize("name", ALL_CASES)
def test_numpy_oracle_fp64_matches_reference_fp64(name):
    """Independent float64 restatement (hand-derived backward) vs the reference
    source run in float64 under the shim: forward <= 1e-9, grads <= 1e-8 rel."""
    g = load(name)
    fw, bw, _ = run_numpy_oracle(name, g, np.float64)
    assert maxabs(fw["tr_pc"], g["tr_pc_f64"]) < 1e-12
    assert maxabs(fw["proj"], g["proj_f64"]) < 1e-9
    if "proj_depth_f64" in g:
        assert maxabs(fw["proj_depth"], g["proj_depth_f64"]) < 1e-8
    if "voxels_f64" in g:
        assert maxabs(fw["voxels"], g["voxels_f64"]) < 1e-9
    if "drc_probs_f64" in g:
        assert maxabs(fw["drc_probs"], g["drc_probs_f64"]) < 1e-9
    if "grid_raw_f64" in g:
        assert maxabs(fw["G0"], g["grid_raw_f64"]) < 1e-12
        assert maxabs(fw["G2"], g["grid_blur_f64"]) < 1e-9
    for k in GRAD_KEYS:
        if k + "_f64" in g:
            assert relerr(bw[k], g[k + "_f64"]) < 1e-8, k


@pytest.mark.parametrize("name", ALL_CASES)
def test_reference_cpu_fp32_matches_reference_fp32(name):
    """torch-CPU op-for-op restatement in fp32 vs the reference source in fp32.
    Same ops in the same order => agreement to a few ulp."""
    g = load(name)
    out, grads = run_reference_cpu(name, g, torch.float32)
    assert maxabs(out["tr_pc"].detach().numpy(), g["tr_pc_f32"]) < 2e-6
    assert maxabs(out["proj"].detach().numpy(), g["proj_f32"]) < 5e-6
    if "proj_depth_f32" in g:
        assert maxabs(out["proj_depth"].detach().numpy(), g["proj_depth_f32"]) < 5e-5
    if "voxels_f32" in g:
        assert maxabs(out["voxels"].detach().numpy(), g["voxels_f32"]) < 5e-6
    for k in GRAD_KEYS:
        if k + "_f32" in g:
            assert relerr(grads[k], g[k + "_f32"]) < 2e-4, k


@pytest.mark.parametrize("name", ["tiny", "cfg1"])
def test_fp32_reference_error_budget(name):
    """How far the reference's own fp32 arithmetic is from fp64 truth: the
    <=1e-4 silhouette tolerance leaves > 10x headroom for atomics ordering."""
    g = load(name)
    assert maxabs(g["proj_f32"], g["proj_f64"]) < 1e-5


def test_digests_match():
    for name in ("cfg1", "mid"):
        g = load(name)
        fw, _, _ = run_numpy_oracle(name, g, np.float64, grads=False)
        for key, arr in (("raw", fw["G0"]), ("blur", fw["G2"])):
            flat = arr.reshape(-1)
            assert abs(flat.sum() - g[key + "_sum"]) < 1e-6 * max(1.0, abs(g[key + "_sum"]))
            assert maxabs(flat[g[key + "_idx"]], g[key + "_val"]) < 1e-9


def test_nan_points_are_dropped_forward():
    g = load("tiny_nan")
    inp = {k: g[k].astype(np.float64) for k in ("pc", "pose", "trans", "scale")}
    taps = onp.smoothing_taps(16, -1, 5, 0.8)
    fw = onp.project_forward(inp["pc"], inp["pose"], inp["trans"], inp["scale"], None, taps, Dz=16, D=16)
    assert np.isfinite(fw["proj"]).all()
    assert maxabs(fw["proj"], g["proj_f64"]) < 1e-9


# ---- analytic known answers --------------------------------------------------------
def test_kat_gauss_taps():
    k = onp.gauss_kernel_1d(5, 0.8)
    assert np.allclose(k, [0.02193, 0.22851, 0.49912, 0.22851, 0.02193], atol=5e-6)
    assert abs(k.sum() - 1) < 1e-15
    with pytest.raises(ValueError):
        onp.gauss_kernel_1d(4, 1.0)
    tx, ty, tz = onp.smoothing_taps(16, 8, 5, 0.8)
    assert len(tx) == 5 and len(tz) == 3


def test_kat_point_at_cell_centre_and_corner_weights():
    D = 9
    node = lambda i: i / (D - 1) - 0.5
    tr = np.array([[[node(3), node(4), node(5)]]])
    G = onp.voxelize_fwd(tr, D, D)
    assert G[0, 3, 4, 5] == 1.0 and G.sum() == 1.0
    fr = (0.25, 0.5, 0.75)
    tr = np.array([[[node(2) + fr[0] / (D - 1), node(2) + fr[1] / (D - 1), node(2) + fr[2] / (D - 1)]]])
    G = onp.voxelize_fwd(tr, D, D)
    for k in range(2):
        for j in range(2):
            for l in range(2):
                w = (fr[0] if k else 1 - fr[0]) * (fr[1] if j else 1 - fr[1]) * (fr[2] if l else 1 - fr[2])
                assert abs(G[0, 2 + k, 2 + j, 2 + l] - w) < 1e-12
    assert abs(G.sum() - 1) < 1e-12


def test_kat_mass_conservation_and_outliers():
    rng = np.random.default_rng(0)
    tr = rng.uniform(-0.6, 0.6, (3, 200, 3))
    tr[0, 0, :] = 0.5          # exactly on the closed boundary: valid, upper corners skipped
    tr[0, 1, :] = np.nan
    valid = np.all((tr >= -0.5) & (tr <= 0.5), axis=-1)
    G = onp.voxelize_fwd(tr, 8, 12)
    assert np.allclose(G.sum((1, 2, 3)), valid.sum(1), atol=1e-9)
    dtr = onp.voxelize_bwd(tr, rng.standard_normal(G.shape), 8, 12)
    assert (dtr[~valid] == 0).all()


def test_kat_blur_of_interior_delta_is_outer_product():
    G = np.zeros((1, 16, 16, 16))
    G[0, 8, 7, 9] = 1.0
    t = onp.smoothing_taps(16, -1, 5, 0.8)
    out = onp.blur3d(G, t)
    exp = np.einsum("i,j,k->ijk", t[2], t[1], t[0])
    assert maxabs(out[0, 6:11, 5:10, 7:12], exp) < 1e-15
    assert abs(out.sum() - 1) < 1e-12


def test_kat_blur_is_self_adjoint():
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((2, 1, 6, 7, 7))
    t = [onp.gauss_kernel_1d(5, 0.9), onp.gauss_kernel_1d(3, 0.7), onp.gauss_kernel_1d(5, 1.3)]
    lhs = (onp.blur3d(a, t) * b).sum()
    rhs = (a * onp.blur3d(b, t, order=("z", "y", "x"))).sum()
    assert abs(lhs - rhs) < 1e-12


@pytest.mark.parametrize("Dz,expect", [(64, 6.398e-4), (128, 1.279e-3), (256, 2.557e-3)])
def test_kat_empty_ray_background(Dz, expect):
    p, proj = onp.drc_fwd(np.zeros((1, Dz, 1, 1)))
    assert abs(proj[0, 0, 0] - expect) < 2e-6
    assert abs(proj[0, 0, 0] - (1 - (1 - 1e-5) ** Dz)) < 1e-8     # up to the e^eps factor on term 0


def test_kat_saturated_ray():
    G = np.zeros((1, 32, 1, 1))
    G[0, 5] = 1.0
    p, proj = onp.drc_fwd(G)
    assert abs(proj[0, 0, 0] - 1) < 2e-5
    assert abs(p.sum() - 1) < 5e-5


def test_drc_bwd_matches_finite_differences():
    rng = np.random.default_rng(2)
    G = rng.uniform(0.02, 0.9, (1, 6, 2, 2))
    gamma = rng.standard_normal((7, 1, 2, 2))
    an = onp.drc_bwd(G, gamma)
    h = 1e-6
    for idx in [(0, 0, 0, 0), (0, 3, 1, 0), (0, 5, 1, 1)]:
        Gp, Gm = G.copy(), G.copy()
        Gp[idx] += h
        Gm[idx] -= h
        fd = ((onp.drc_fwd(Gp)[0] * gamma).sum() - (onp.drc_fwd(Gm)[0] * gamma).sum()) / (2 * h)
        assert abs(fd - an[idx]) < 1e-6 * max(1, abs(fd))


def test_full_backward_matches_finite_differences_fp64():
    """Central differences (fp64) of the numpy forward vs the hand-derived
    backward, through the whole chain, on sampled coordinates."""
    g = load("tiny")
    inp = {k: g[k].astype(np.float64) for k in ("pc", "pose", "trans", "scale")}
    taps = onp.smoothing_taps(16, -1, 5, 0.8)
    w = g["w_proj"].astype(np.float64)
    wd = g["w_depth"].astype(np.float64)

    def loss(**over):
        a = dict(inp)
        a.update(over)
        fw = onp.project_forward(a["pc"], a["pose"], a["trans"], a["scale"], None, taps, Dz=16, D=16)
        return (fw["proj"] * w).sum() + (fw["proj_depth"] * wd).sum()

    fw = onp.project_forward(inp["pc"], inp["pose"], inp["trans"], inp["scale"], None, taps, Dz=16, D=16)
    bw = onp.project_backward(inp["pc"], inp["pose"], inp["trans"], inp["scale"], None, taps, fw,
                              dproj=w, dproj_depth=wd)
    h = 1e-6
    rng = np.random.default_rng(5)
    checks = [("pose", "dpose", (0, 1)), ("pose", "dpose", (1, 3)), ("trans", "dtrans", (0, 0)),
              ("trans", "dtrans", (1, 2)), ("scale", "dscale", (0, 0))]
    checks += [("pc", "dpc", (int(rng.integers(2)), int(rng.integers(4, 64)), int(rng.integers(3)))) for _ in range(6)]
    for name, gname, idx in checks:
        ap, am = inp[name].copy(), inp[name].copy()
        ap[idx] += h
        am[idx] -= h
        fd = (loss(**{name: ap}) - loss(**{name: am})) / (2 * h)
        assert abs(fd - bw[gname][idx]) < 2e-5 * max(1.0, abs(fd)), (name, idx, fd, bw[gname][idx])


def test_reference_cpu_rejects_unsupported():
    cfg = rcpu.Cfg(vox_size=8)
    with pytest.raises(ValueError):
        rcpu.gauss_kernel_1d(4, 1.0)
    with pytest.raises(NotImplementedError):
        rcpu.pointcloud2voxels3d_fast(cfg, torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))


@pytest.mark.parametrize("name", ["vox_analytical", "vox_none", "vox_sum"])
def test_numpy_gauss_voxeliser_matches_reference(name):
    """oracle restatement of pointcloud2voxels (point_cloud.py:17-57) + explicit backward vs
    goldens produced by the reference's own source under autodiff."""
    from helpers import onp
    g = load("slow_path")
    B, N, G, nsum, nana = (int(v) for v in g[name + "_meta"])
    mode = "sum" if nsum else ("analytical" if nana else None)
    sigma = float(g[name + "_sigma"])
    vox, raw = onp.gauss_voxelize_fwd(g[name + "_pc"], G, sigma, mode)
    assert maxabs(vox[..., None], g[name + "_vox_f64"]) < 1e-12
    d = onp.gauss_voxelize_bwd(g[name + "_pc"], G, sigma, raw, g[name + "_w"][..., 0].astype(np.float64), mode)
    assert relerr(d, g[name + "_dpc_f64"]) < 1e-10
