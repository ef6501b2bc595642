"""Mirror of dpc/util/drc.py (differentiable ray consistency) over the HIP
kernels.  Same names, argument order and tensor layouts as the reference:
voxels [B,Dz,D,D,1] -> proj [B,D,D,1], event probabilities p [Dz+1,B,D,D,1].

The reference's switches (default_config.yaml:88-90): drc_logsum=true is the log-space
form with the clip to [eps, 1-eps] and the eps "unity"; drc_logsum=false is the plain
product form p_i = c_i * prod_{j<i}(1-c_j) -- the same kernels with eps = 0 (the kernels
never take a log either way); drc_tf_cumulative only chooses between tf.cumsum and an
equivalent python loop, which is the same sequence of operations.  In the product form
the kernels keep occupancies inside [0, 1] (the projector's grids always are).
"""
import torch

from .. import _capi, ops


def _meta(cfg, Dz, D, collapse=_capi.DPC_COLLAPSE_DRC):
    return ops.ProjMeta(Dz=int(Dz), D=int(D), camera_distance=float(cfg.camera_distance),
                        focal_length=float(cfg.focal_length),
                        eps=float(cfg.drc_logsum_clip_val) if getattr(cfg, "drc_logsum", True) else 0.0,
                        max_depth=float(cfg.max_depth),
                        pose_quaternion=bool(getattr(cfg, "pose_quaternion", True)),
                        collapse_mode=collapse)


def _grid4(voxels):
    if voxels.dim() == 5:
        if voxels.shape[-1] != 1:
            raise ValueError("voxels must be [B,Dz,D,D,1]")
        return voxels.reshape(voxels.shape[:4])
    if voxels.dim() != 4:
        raise ValueError("voxels must be [B,Dz,D,D,1]")
    return voxels


def drc_event_probabilities_impl(voxels, cfg, flip_h=False):
    """dpc/util/drc.py:47-102.  Returns (p, proj) both with a trailing 1."""
    v = _grid4(voxels)
    proj, p = ops.DrcProjection.apply(v, _meta(cfg, v.shape[1], v.shape[2]), 1 if flip_h else 0)
    return p.unsqueeze(-1), proj.unsqueeze(-1)


def drc_event_probabilities(voxels, cfg):
    return drc_event_probabilities_impl(voxels, cfg)[0]


def drc_projection(voxels, cfg):
    """dpc/util/drc.py:110-123 -> (proj [B,D,D,1], p [Dz+1,B,D,D,1])."""
    p, proj = drc_event_probabilities_impl(voxels, cfg)
    return proj, p


def drc_depth_grid(cfg, z_size, device=None, dtype=torch.float32):
    """dpc/util/drc.py:139-143: psi_i = i/Dz - 0.5 + camera_distance, psi_Dz = max_depth."""
    zs = torch.as_tensor(float(z_size), dtype=dtype, device=device)
    i_s = torch.arange(0, int(z_size), dtype=dtype, device=device)
    di_s = i_s / zs - 0.5 + cfg.camera_distance
    last = torch.full((1,), float(cfg.max_depth), dtype=dtype, device=device)
    return torch.cat([di_s, last], dim=0)


def drc_depth_projection(p, cfg):
    """dpc/util/drc.py:146-153: sum_i p_i psi_i over [Dz+1,B,D,D,1].  (On the
    fused path the depth image comes out of k_zfwd directly; this standalone
    form is a single broadcast-multiply-reduce on the device.)"""
    z_size = p.shape[0] - 1
    psi = drc_depth_grid(cfg, z_size, device=p.device, dtype=p.dtype).reshape(-1, 1, 1, 1, 1)
    return (p * psi).sum(dim=0)


def project_volume_rgb_integral(cfg, p, rgb):
    """dpc/util/drc.py:126-136: sum_i p_i rgb_i with a white background behind the grid.
    p [Dz+1,B,D,D,1], rgb [B,Dz,D,D,3] -> [B,D,D,3] (one broadcast-multiply-reduce)."""
    rgb = rgb.permute(1, 0, 2, 3, 4)
    background = torch.ones_like(rgb[:1])
    rgb_full = torch.cat([rgb, background], dim=0)
    return (p * rgb_full).sum(dim=0)
