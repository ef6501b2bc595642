"""torch-CPU restatement of the reference projector GRAPH, op for op: eight
dense zero-filled scatter grids summed, three single-channel ``conv3d`` passes,
log-space ``cumsum`` ray collapse, backward by autograd -- i.e. the amount of
work the reference's TF1 CPU path does, not a clever fused version.

TEST INFRASTRUCTURE ONLY (oracle).  It is the parity yardstick that travels to
the GPU box and the "reference CPU path" timed by ``bench.py``'s
``cpu_baseline`` leg (kind = "port").  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py`` may import it; the product
package never does.

Parity status: pinned against the reference's own source run under
``oracle/tf_shim`` (goldens in ``tests/golden``); TensorFlow leaf-op
semantics are restated, not checked against a TF binary (none installable).

Reference lines followed (/root/reference):
  dpc/util/point_cloud.py:60-136 (voxeliser), :139-145 (blur), :157-216
  (transform), :229-290 (orchestration, clips, flips); dpc/util/drc.py:47-123,
  :139-153; dpc/util/gauss_kernel.py:5-11,:27-54; dpc/util/quaternion.py:62-117;
  dpc/util/camera.py:5-13.
"""
import math

import torch
import torch.nn.functional as F


class Cfg(dict):
    """Attribute-style config with the reference defaults the path reads
    (dpc/resources/default_config.yaml)."""
    DEFAULTS = dict(
        vox_size=64, vox_size_z=-1, camera_distance=2.0, focal_length=1.875,
        pose_quaternion=True, pc_gauss_kernel_size=11,
        pc_separable_gauss_filter=True, ptn_max_projection=False,
        drc_logsum=True, drc_logsum_clip_val=1e-5, drc_tf_cumulative=True,
        max_depth=10.0, pc_rgb_stop_points_gradient=False,
        pc_rgb_clip_after_conv=False, pc_rgb_divide_by_occupancies=False,
        pc_rgb_divide_by_occupancies_epsilon=0.01)

    def __init__(self, **kw):
        super().__init__(self.DEFAULTS)
        self.update(kw)

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def gauss_kernel_1d(size, sigma, dtype=torch.float32):
    size = int(size)
    if size % 2 != 1:
        raise ValueError("even kernel sizes need TF's asymmetric SAME padding; unsupported")
    xx = torch.arange(-size // 2 + 1.0, size // 2 + 1.0, dtype=dtype)
    k = torch.exp(-xx ** 2 / (2.0 * sigma ** 2))
    return k / k.sum()


def smoothing_kernel(cfg, sigma, dtype=torch.float32):
    """-> [k_x, k_y, k_z] shaped [1,1,K,1,1], [1,K,1,1,1], [Kz,1,1,1,1]."""
    fsz = cfg.pc_gauss_kernel_size
    k = gauss_kernel_1d(fsz, sigma, dtype)
    kz = k
    if cfg.vox_size_z != -1:
        ratio = cfg.vox_size_z / cfg.vox_size
        fz = int(math.floor(fsz * ratio))
        if fz % 2 == 0:
            fz += 1
        kz = gauss_kernel_1d(fz, sigma * ratio, dtype)
    return [k.reshape(1, 1, -1, 1, 1), k.reshape(1, -1, 1, 1, 1), kz.reshape(-1, 1, 1, 1, 1)]


def _hamilton(a, b):
    w1, x1, y1, z1 = a.unbind(-1)
    w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                        w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2), dim=-1)


def quaternion_rotate(pc, q):
    q = q / q.norm(dim=-1, keepdim=True)
    q = q.unsqueeze(1)
    qc = q * q.new_tensor([1.0, -1.0, -1.0, -1.0])
    p4 = F.pad(pc, (1, 0))
    return _hamilton(_hamilton(q, p4), qc)[:, :, 1:4]


def pc_perspective_transform(cfg, point_cloud, transform, predicted_translation=None,
                             focal_length=None):
    cd = cfg.camera_distance
    f = cfg.focal_length if focal_length is None else focal_length.unsqueeze(-1)
    if cfg.pose_quaternion:
        p2 = quaternion_rotate(point_cloud, transform)
        if predicted_translation is not None:
            p2 = p2 + predicted_translation.unsqueeze(1)
        xs, ys, zs = p2[:, :, 2:3], p2[:, :, 1:2], p2[:, :, 0:1]
        zs = zs + cd
        xs = xs * f
        ys = ys * f
    else:
        if predicted_translation is not None:
            raise ValueError("translation requires a quaternion pose")
        intr = torch.eye(4, dtype=point_cloud.dtype)
        intr[1, 1] = intr[2, 2] = float(cfg.focal_length)
        full = intr.unsqueeze(0) @ transform
        xyz1 = F.pad(point_cloud, (0, 1), value=1.0)
        p2 = xyz1 @ full.transpose(1, 2)
        xs, ys, zs = p2[:, :, 2:3], p2[:, :, 1:2], p2[:, :, 0:1]
    xs = xs / zs
    ys = ys / zs
    zs = zs - cd
    if predicted_translation is not None:
        zs = zs - predicted_translation.unsqueeze(1)[:, :, 0:1]
    return torch.cat([zs, ys, xs], dim=2)


def pointcloud2voxels3d_fast(cfg, pc, rgb=None):
    if rgb is not None:
        raise NotImplementedError("RGB channels are SURVEY.md 8(f) scope")
    D = cfg.vox_size
    Dz = cfg.vox_size_z if cfg.vox_size_z != -1 else D
    B, N = pc.shape[0], pc.shape[1]
    valid = ((pc >= -0.5) & (pc <= 0.5)).all(dim=-1).reshape(-1)
    size = pc.new_tensor([Dz, D, D]).reshape(1, 1, 3)
    g = (pc + 0.5) * (size - 1)
    fl = torch.floor(g)
    idx = fl.to(torch.int64)
    b = torch.arange(B).reshape(B, 1, 1).expand(B, N, 1)
    idx = torch.cat([b, idx], dim=2).reshape(-1, 4)[valid]
    r = g - fl
    rr = [1.0 - r, r]
    grids = []
    for k in range(2):
        for j in range(2):
            for i in range(2):
                upd = (rr[k][:, :, 0] * rr[j][:, :, 1] * rr[i][:, :, 2]).reshape(-1)[valid]
                loc = idx + idx.new_tensor([[0, k, j, i]])
                ok = (loc[:, 1] < Dz) & (loc[:, 2] < D) & (loc[:, 3] < D)
                loc, upd = loc[ok], upd[ok]
                grid = torch.zeros(B, Dz, D, D, dtype=pc.dtype)
                grid = grid.index_put((loc[:, 0], loc[:, 1], loc[:, 2], loc[:, 3]), upd, accumulate=True)
                grids.append(grid)
    out = grids[0]
    for gr in grids[1:]:
        out = out + gr
    return out, None


def _conv3d_same(x, filt):
    kd, kh, kw = filt.shape[0], filt.shape[1], filt.shape[2]
    w = filt.reshape(1, 1, kd, kh, kw).to(x.dtype)
    return F.conv3d(x.permute(0, 4, 1, 2, 3), w, padding=(kd // 2, kh // 2, kw // 2)).permute(0, 2, 3, 4, 1)


def smoothen_voxels3d(cfg, voxels, kernel):
    if not cfg.pc_separable_gauss_filter:
        raise NotImplementedError("dense 3-D kernel path")
    for k in kernel:
        voxels = _conv3d_same(voxels, k)
    return voxels


def drc_event_probabilities(voxels, cfg):
    if not (cfg.drc_logsum and cfg.drc_tf_cumulative):
        raise NotImplementedError("only the default log-space cumsum DRC")
    e = cfg.drc_logsum_clip_val
    inp = voxels.permute(1, 0, 2, 3, 4)
    inp = torch.clamp(inp, e, 1.0 - e)
    y = torch.log(inp)
    x = torch.log(1.0 - inp)
    r = torch.cumsum(x, dim=0)
    unit = torch.ones_like(inp[:1]) * e
    p = torch.exp(torch.cat([unit, r], dim=0) + torch.cat([y, unit], dim=0))
    return p


def drc_projection(voxels, cfg):
    p = drc_event_probabilities(voxels, cfg)
    return p[:-1].sum(dim=0), p


def drc_depth_projection(p, cfg):
    Dz = p.shape[0] - 1
    zsz = torch.tensor(float(Dz), dtype=p.dtype)
    psi = torch.arange(0, Dz, dtype=p.dtype) / zsz - 0.5 + cfg.camera_distance
    psi = torch.cat([psi, torch.tensor([cfg.max_depth], dtype=p.dtype)]).reshape(-1, 1, 1, 1, 1)
    return (p * psi).sum(dim=0)


def pointcloud_project_fast(cfg, point_cloud, transform, predicted_translation,
                            all_rgb, kernel=None, scaling_factor=None, focal_length=None):
    if all_rgb is not None:
        raise NotImplementedError("RGB channels are SURVEY.md 8(f) scope")
    tr_pc = pc_perspective_transform(cfg, point_cloud, transform, predicted_translation, focal_length)
    voxels, _ = pointcloud2voxels3d_fast(cfg, tr_pc, None)
    voxels_raw = voxels.unsqueeze(-1)
    voxels = torch.clamp(voxels_raw, 0.0, 1.0)
    voxels_clip = voxels
    if kernel is not None:
        voxels = smoothen_voxels3d(cfg, voxels, kernel)
    voxels_blur = voxels
    if scaling_factor is not None:
        voxels = torch.clamp(voxels * scaling_factor.reshape(-1, 1, 1, 1, 1), 0.0, 1.0)
    if cfg.ptn_max_projection:
        proj = voxels.amax(dim=1)
        drc_probs = proj_depth = None
    else:
        proj, drc_probs = drc_projection(voxels, cfg)
        drc_probs = torch.flip(drc_probs, dims=[2])
        proj_depth = drc_depth_projection(drc_probs, cfg)
    proj = torch.flip(proj, dims=[1])
    return {"proj": proj, "voxels": voxels, "tr_pc": tr_pc, "voxels_rgb": None,
            "proj_rgb": None, "drc_probs": drc_probs, "proj_depth": proj_depth,
            # extras (not in the reference dict) for stage-level parity tests
            "_voxels_raw": voxels_raw, "_voxels_clip": voxels_clip, "_voxels_blur": voxels_blur}
