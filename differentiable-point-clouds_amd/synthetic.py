"""Seeded synthetic workloads of SURVEY.md section 8(d) (numpy only).

Shared by ``bench.py``, the golden generator and the parity tests so that the
HIP path, the oracle and the committed fixtures all see identical inputs.

  points  [B,N,3] 99 % "shell": unit directions x radius 0.30 + 0.01 N(0,1)
                  (surface-like clustering -> realistic scatter contention),
                  1 % uniform in [-0.7,0.7]^3 (exercises outlier filtering)
  pose    [B,4]   N(0,1)^4, UNNORMALISED (the op normalises)
  scale   [B,1]   U(0.5, 1.0)
  gt      [B,D,D,1] centred filled disk of radius 0.3 D
"""
import numpy as np

SEED0 = 20260927

# BASELINE.json configs that are projector-only (cfg3/4 are the training step)
CONFIGS = {
    1: dict(B=4, N=1000, D=64, K=11, sigma=1.0),
    2: dict(B=32, N=8000, D=128, K=11, sigma=1.6),
    5: dict(B=8, N=16000, D=256, K=11, sigma=2.0),
}


def make_points(rng, B, N, kind="shell"):
    if kind == "shell":
        d = rng.standard_normal((B, N, 3))
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        rad = 0.30 + 0.01 * rng.standard_normal((B, N, 1))
        pts = d * rad
        n_out = max(1, N // 100)
        pts[:, :n_out, :] = rng.uniform(-0.7, 0.7, (B, n_out, 3))
    elif kind == "ball":
        d = rng.standard_normal((B, N, 3))
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        pts = d * (0.38 * rng.uniform(0, 1, (B, N, 1)) ** (1.0 / 3.0))
    else:
        raise ValueError(kind)
    return pts.astype(np.float32)


def make_inputs(B, N, seed, kind="shell"):
    rng = np.random.default_rng(seed)
    pc = make_points(rng, B, N, kind)
    pose = rng.standard_normal((B, 4)).astype(np.float32)
    scale = rng.uniform(0.5, 1.0, (B, 1)).astype(np.float32)
    return dict(pc=pc, pose=pose, scale=scale)


def config_inputs(cfg_id, B=None, kind="shell", seed_offset=0):
    c = dict(CONFIGS[cfg_id])
    if B is not None:
        c["B"] = B
    inp = make_inputs(c["B"], c["N"], SEED0 + cfg_id + seed_offset, kind)
    inp.update(c)
    return inp


def disk_gt(B, D, radius=0.3):
    yy, xx = np.meshgrid(np.arange(D), np.arange(D), indexing="ij")
    c = (D - 1) / 2.0
    m = ((yy - c) ** 2 + (xx - c) ** 2) <= (radius * D) ** 2
    return np.broadcast_to(m.astype(np.float32)[None, :, :, None], (B, D, D, 1)).copy()


def algorithmic_bytes_per_view(N, Dz, D):
    """8 V + P, V = 4 Dz D^2, P = 24 N + 128 N + 12 D^2."""
    V = 4 * Dz * D * D
    return 8 * V + 24 * N + 128 * N + 12 * D * D
