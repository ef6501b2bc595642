"""Mirror of dpc/util/camera.py:5-13 (only what the hot path uses)."""
import numpy as np


def intrinsic_matrix(cfg, dims=3, inverse=False):
    """diag(1, f, f[, 1]) -- the matrix-pose branch of pc_perspective_transform
    (dpc/util/point_cloud.py:194) multiplies the extrinsic by it; the HIP
    transform kernel folds it into the row scaling of the camera matrix."""
    val = float(cfg.focal_length)
    if inverse:
        val = 1.0 / val
    m = np.eye(dims, dtype=np.float32)
    m[1, 1] = val
    m[2, 2] = val
    return m
