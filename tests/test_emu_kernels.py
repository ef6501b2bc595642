"""CPU tier for the kernels: the SAME kernel source (csrc/dpc_kernels.hip)
compiled for the thread-per-lane HIP emulator (tests/hipemu) and driven through
the product Python API + C ABI, checked against the goldens generated from the
reference source.  This pins indexing / tiling / reduction logic before GPU
minutes are spent; the `-m gpu` tests repeat the same checks on the real
library."""
import numpy as np
import pytest
import torch

from helpers import ALL_CASES, DRC_VARIANT_CASES, load, maxabs, relerr
from run_case import run_product
import parity_cases

SMALL = [c for c in ALL_CASES if c not in ("mid", "cfg1")]
GRAD_KEYS = ("dpc", "dpose", "dtrans", "dscale", "dfocal")

# fp32 tolerances.  Forward: the north-star bar is 1e-4 max-abs on the
# silhouette; we hold 2e-5 against fp64 truth.  Gradients: relative to the
# largest reference gradient magnitude of that tensor.
TOL_PROJ = 2e-5
TOL_DEPTH = 2e-4
TOL_GRAD = 2e-4


@pytest.mark.parametrize("name", SMALL + DRC_VARIANT_CASES)
def test_emu_forward_backward_matches_goldens(emu, name):
    g = load(name)
    res, gr = run_product(name, g, "cpu", grads=True, touch_lazy=True)
    assert maxabs(res["tr_pc"], g["tr_pc_f64"]) < 2e-6
    assert maxabs(res["proj"], g["proj_f64"]) < TOL_PROJ
    if "proj_depth_f64" in g:
        assert maxabs(res["proj_depth"], g["proj_depth_f64"]) < TOL_DEPTH
    if "voxels_f64" in g:
        assert maxabs(res["voxels"], g["voxels_f64"]) < TOL_PROJ
    if "drc_probs_f64" in g:
        assert maxabs(res["drc_probs"], g["drc_probs_f64"]) < TOL_PROJ
    for k in GRAD_KEYS:
        if k + "_f64" in g:
            assert relerr(gr[k], g[k + "_f64"]) < TOL_GRAD, k


def test_emu_cfg1(emu):
    g = load("cfg1")
    res, gr = run_product("cfg1", g, "cpu", grads=True)
    assert maxabs(res["proj"], g["proj_f64"]) < TOL_PROJ
    assert maxabs(res["proj_depth"], g["proj_depth_f64"]) < TOL_DEPTH
    for k in ("dpc", "dpose", "dscale"):
        assert relerr(gr[k], g[k + "_f64"]) < TOL_GRAD, k


def test_emu_stage_level_api(emu):
    parity_cases.stage_level_api_matches_cpu_oracle("cpu")


@pytest.mark.parametrize("D,K", parity_cases.ODD_CASES)
def test_emu_odd_sizes_and_generic_tap_counts(emu, D, K):
    parity_cases.odd_sizes_and_generic_tap_counts("cpu", D, K)


def test_emu_nan_points_dropped(emu):
    g = load("tiny_nan")
    res, _ = run_product("tiny_nan", g, "cpu", grads=False)
    assert np.isfinite(res["proj"]).all()
    assert maxabs(res["proj"], g["proj_f64"]) < TOL_PROJ


@pytest.mark.parametrize("case", parity_cases.FUSED_CASES[:2])
def test_emu_fused_path_against_numpy_oracle(emu, case):
    parity_cases.fused_path_against_numpy_oracle("cpu", *case)


@pytest.mark.parametrize("name", ["tiny_rgb", "tiny_rgb_div"])
def test_emu_rgb_channels(emu, name):
    parity_cases.rgb_case_matches_goldens("cpu", name)


@pytest.mark.parametrize("name", parity_cases.LOSS_CASES)
def test_emu_silhouette_loss(emu, name):
    parity_cases.silhouette_loss_matches_reference("cpu", name)


@pytest.mark.parametrize("name,tag", parity_cases.NN_CASES)
def test_emu_nn_distance(emu, name, tag):
    parity_cases.nn_distance_matches_reference("cpu", name, tag)


def test_emu_nn_distance_gradient(emu):
    parity_cases.nn_distance_gradient("cpu")


@pytest.mark.parametrize("name", parity_cases.SLOW_VOX_CASES)
def test_emu_gauss_voxeliser(emu, name):
    parity_cases.gauss_voxeliser_matches_reference("cpu", name)


def test_emu_slow_projector(emu):
    parity_cases.slow_projector_matches_reference("cpu")


def test_emu_gauss_voxeliser_multitile(emu):
    parity_cases.gauss_voxeliser_multitile_against_numpy_oracle("cpu", N=70, G=66)


def test_emu_fused_edge_planes(emu):
    parity_cases.fused_edge_planes_against_numpy_oracle("cpu")

