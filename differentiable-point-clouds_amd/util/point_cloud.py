"""Mirror of the reference's projector API (dpc/util/point_cloud.py) on
PyTorch-ROCm tensors, executed by the hand-written HIP kernels.

Same function names, argument order, tensor layouts and dict keys as the
reference, so that a caller written against ``util.point_cloud`` (e.g.
dpc/models/model_pc.py:241-244, dpc/run/predict.py:56-57,130-132) consumes it
unchanged:

    pointcloud_project_fast(cfg, point_cloud, transform, predicted_translation,
                            all_rgb, kernel=None, scaling_factor=None,
                            focal_length=None) -> dict
    pointcloud2voxels3d_fast(cfg, pc, rgb) -> (voxels [B,Dz,D,D], voxels_rgb)
    smoothen_voxels3d(cfg, voxels [B,Dz,D,D,1], kernel) -> [B,Dz,D,D,1]
    pc_perspective_transform(cfg, point_cloud, transform,
                             predicted_translation=None, focal_length=None)

TF1 graph mode never computes a tensor nobody fetches; eager PyTorch would.
``pointcloud_project_fast`` therefore returns a dict whose cheap entries
(``proj``, ``proj_depth``, ``tr_pc``) come from ONE fused autograd node and
whose bulky entries (``voxels`` [B,Dz,D,D,1], ``drc_probs`` [Dz+1,B,D,D,1])
are materialised on first access through the stage-level kernels, with
autograd wired through ``tr_pc``.
"""
import torch

import weakref

from .. import _capi, ops
from . import drc as _drc


def _dims(cfg):
    D = int(cfg.vox_size)
    Dz = int(cfg.vox_size_z) if cfg.vox_size_z != -1 else D
    return Dz, D


def _meta(cfg, collapse=None):
    # (a dpc_amd Config counts its edits: the derived tuple is reused until the config changes; any other mapping --
    # the reference's EasyDict -- is read afresh)
    edits = cfg.__dict__.get("_edits") if isinstance(cfg, dict) and hasattr(cfg, "__dict__") else None
    if edits is not None and collapse is None:
        hit = _META_CACHE.get(id(cfg))
        if hit is not None and hit[0] == edits and hit[1]() is cfg:
            return hit[2]
    Dz, D = _dims(cfg)
    c = collapse
    if c is None:
        c = _capi.DPC_COLLAPSE_MAX if cfg.ptn_max_projection else _capi.DPC_COLLAPSE_DRC
    meta = _drc._meta(cfg, Dz, D, c)
    if edits is not None and collapse is None:
        if len(_META_CACHE) > 64:
            _META_CACHE.clear()
        _META_CACHE[id(cfg)] = (edits, weakref.ref(cfg), meta)        # (weak: the cache keeps no config alive)
    return meta


_META_CACHE = {}


def _flat_taps(cfg, kernel, device):
    """Reference kernels are a list of 3 conv3d filters [1,1,K,1,1], [1,K,1,1,1],
    [Kz,1,1,1,1] applied in list order; which axis each one blurs is read off
    its shape (dpc/util/gauss_kernel.py:27-32).

    Filters made by ``smoothing_kernel`` from a host-side sigma carry ``dpc_support`` = the largest tap offset whose
    weight is still >= 1e-8 of the centre tap (util/gauss_kernel.py).  Of such a filter only the central taps are
    handed to the kernels -- as a view into the same buffer, rounded up to the next tap count the library has
    compiled kernels for (dpc_compiled_taps) -- so that a K = 21 configuration whose sigma has been annealed to 0.8
    runs the 11-tap kernels (and their cheaper saved-state layout) instead of multiplying by twenty zeros.  Results
    equal the full filter's to fp32 rounding; ``cfg.pc_trim_gauss_taps = False`` switches it off."""
    if kernel is None:
        return None, None, None
    if not cfg.pc_separable_gauss_filter or isinstance(kernel, torch.Tensor):
        raise NotImplementedError("dense 3-D Gaussian kernel (pc_separable_gauss_filter=false)")
    # the filters only change when sigma does (every step at most, usually far less often), the
    # projector runs every step: remember the flattened taps of the last filter list seen
    if not all(isinstance(k, torch.Tensor) for k in kernel):
        return _flat_taps_uncached(kernel, device)
    trim = bool(getattr(cfg, "pc_trim_gauss_taps", True))
    # _version: in-place edits invalidate; dpc_support: the host-side tag moves with sigma (also in place under graph replay)
    key = [trim, device]
    for k in kernel:
        key.append(k._version)
        key.append(getattr(k, "dpc_support", None))
    hit = _TAPS_CACHE.get("key")
    if hit == key:
        filters = _TAPS_CACHE["filters"]
        if len(filters) == len(kernel) and all(a is b for a, b in zip(filters, kernel)):
            return _TAPS_CACHE["taps"]
    out = _flat_taps_uncached(kernel, device)
    if trim:
        out = _trim_taps(kernel, out)
    _TAPS_CACHE.update(key=key, filters=list(kernel), taps=out)
    return out


_TAPS_CACHE = {}


def _trim_taps(kernel, taps):
    """central slices (views) of the flattened taps, per axis, for filters tagged with their effective support"""
    support = {}
    for k, axis in zip(kernel, _filter_axes(kernel)):
        h = getattr(k, "dpc_support", None)
        if h is not None:
            support[axis] = int(h)
    lib = _capi.get_library()
    out = []
    for axis, t in zip("xyz", taps):
        h = support.get(axis)
        if t is not None and h is not None:
            full = int(t.numel())
            k_eff = lib.compiled_taps(2 * max(h, 1) + 1)
            if 0 < k_eff < full:
                off = (full - k_eff) // 2
                t = t[off:off + k_eff]
        out.append(t)
    return tuple(out)


def effective_tap_counts(cfg, kernel, device=None):
    """(Kx, Ky, Kz) the projector will run for this filter list (0 = axis not blurred): what a caller that replays a
    recorded step has to watch -- a HIP graph holds the kernels of ONE set of tap counts (dpc_amd.graphs.RecordedStep)."""
    if kernel is None:
        return (0, 0, 0)
    dev = device if device is not None else kernel[0].device
    return tuple(0 if t is None else int(t.numel()) for t in _flat_taps(cfg, kernel, dev))


def _filter_axes(kernel):
    """the axis ("x" / "y" / "z") each filter of a separable list runs along.  A filter longer than one tap says so by
    its shape ([kd,kh,kw,1,1]); a ONE-tap filter -- gauss_kernel.py:35-54 builds one along z whenever
    round(K * vox_size_z / vox_size) is 1, e.g. vox 112 x 32 deep with K = 5 -- has the shape [1,1,1,1,1] on every axis
    and takes the first axis no other filter of the list claims, in the order the reference builds and applies them
    (x, y, z)."""
    axes = []
    for k in kernel:
        shp = tuple(int(v) for v in k.shape)
        if len(shp) != 5 or shp[3] != 1 or shp[4] != 1:
            raise ValueError("separable kernel filters must be [kd,kh,kw,1,1], got %s" % (shp,))
        if sorted(shp[:3])[:2] != [1, 1]:
            raise ValueError("each separable filter must be 1-D, got %s" % (shp,))
        axes.append("z" if shp[0] > 1 else ("y" if shp[1] > 1 else ("x" if shp[2] > 1 else None)))
    free = [a for a in "xyz" if a not in axes]
    for i, a in enumerate(axes):
        if a is None:
            if not free:
                raise NotImplementedError("two filters along the same axis")
            axes[i] = free.pop(0)
    return axes


def _flat_taps_uncached(kernel, device):
    taps = {"x": None, "y": None, "z": None}
    kernel = [k if (isinstance(k, torch.Tensor) and k.dtype == torch.float32) else torch.as_tensor(k, dtype=torch.float32)
              for k in kernel]
    for k, axis in zip(kernel, _filter_axes(kernel)):
        if taps[axis] is not None:
            raise NotImplementedError("two filters along the same axis")
        if k.numel() == 1 and float(k.reshape(-1)[0]) == 1.0:
            continue                      # the normalised one-tap Gaussian: a pass-through along that axis (no launch)
        taps[axis] = k.reshape(-1) if k.device == device else k.reshape(-1).to(device)
    return taps["x"], taps["y"], taps["z"]


def pc_perspective_transform(cfg, point_cloud, transform, predicted_translation=None, focal_length=None):
    """dpc/util/point_cloud.py:157-216.  Returns [B,N,3] ordered (depth, y, x)."""
    return ops.Transform.apply(point_cloud, transform, predicted_translation, focal_length, _meta(cfg))


def _voxels_rgb_cm(cfg, pc, rgb):
    """channel-major [B,3,Dz,D,D] RGB grid (point_cloud.py:111-118)."""
    Dz, D = _dims(cfg)
    return ops.VoxelizeValues.apply(pc, rgb, Dz, D, bool(getattr(cfg, "pc_rgb_stop_points_gradient", False)))


def pointcloud2voxels3d_fast(cfg, pc, rgb):
    """dpc/util/point_cloud.py:60-136: trilinear scatter-add of [B,N,3] points
    (already in the unit cube) into [B,Dz,D,D]; with rgb [B,N,3] also the
    colour grid [B,Dz,D,D,3]."""
    Dz, D = _dims(cfg)
    voxels = ops.Voxelize.apply(pc, Dz, D)
    voxels_rgb = None
    if rgb is not None:
        voxels_rgb = _voxels_rgb_cm(cfg, pc, rgb).permute(0, 2, 3, 4, 1)
    return voxels, voxels_rgb


def smoothen_voxels3d(cfg, voxels, kernel):
    """dpc/util/point_cloud.py:139-145 on [B,Dz,D,D,1]."""
    tx, ty, tz = _flat_taps(cfg, kernel, voxels.device)
    v = _drc._grid4(voxels)
    out = ops.Blur3d.apply(v, tx, ty, tz)
    return out.unsqueeze(-1) if voxels.dim() == 5 else out


def _convolve_cm(cfg, vox_cm, kernel):
    """per-channel separable blur of a channel-major grid [B,C,Dz,D,D]."""
    tx, ty, tz = _flat_taps(cfg, kernel, vox_cm.device)
    B, C = vox_cm.shape[0], vox_cm.shape[1]
    out = ops.Blur3d.apply(vox_cm.reshape(B * C, *vox_cm.shape[2:]), tx, ty, tz)
    return out.reshape(vox_cm.shape)


def convolve_rgb(cfg, voxels_rgb, kernel):
    """dpc/util/point_cloud.py:148-154 on [B,Dz,D,D,3]."""
    cm = voxels_rgb.permute(0, 4, 1, 2, 3).contiguous()
    return _convolve_cm(cfg, cm, kernel).permute(0, 2, 3, 4, 1)


def _gauss_norm_mode(cfg):
    if getattr(cfg, "pc_normalise_gauss", False):                      # point_cloud.py:43-45
        return 1
    if getattr(cfg, "pc_normalise_gauss_analytical", False):           # point_cloud.py:46-51
        return 2
    return 0


def pointcloud2voxels(cfg, input_pc, sigma):
    """dpc/util/point_cloud.py:17-57: exact Gaussian splat of [B,N,3] points onto the
    vox_size^3 lattice over [-1,1]^3 -> [B,G,G,G,1] in the reference's meshgrid layout
    (axis 1 <- component 1, axis 2 <- component 0, axis 3 <- component 2)."""
    vox = ops.GaussVoxelize.apply(input_pc, float(sigma), int(cfg.vox_size), (1, 0, 2), _gauss_norm_mode(cfg))
    return vox.unsqueeze(-1)


def pointcloud_project(cfg, point_cloud, transform, sigma):
    """dpc/util/point_cloud.py:219-226 (cfg.pc_fast:false): perspective transform ->
    exact Gaussian voxels -> transpose [0,2,1,3,4] -> DRC -> H flip.  Returns
    (proj [B,G,G,1], voxels [B,G,G,G,1]); the transposed layout is produced directly."""
    tr_pc = pc_perspective_transform(cfg, point_cloud, transform)
    vox = ops.GaussVoxelize.apply(tr_pc, float(sigma), int(cfg.vox_size), (0, 1, 2), _gauss_norm_mode(cfg))
    voxels = vox.unsqueeze(-1)
    _, proj = _drc.drc_event_probabilities_impl(voxels, cfg, flip_h=True)       # tf.reverse(proj, [1])
    return proj, voxels


class ProjectionOutputs(dict):
    """The reference's output dict.  ``voxels`` and ``drc_probs`` are computed
    on first access (TF1 graph pruning semantics, see module docstring)."""

    _LAZY = ("voxels", "drc_probs")

    def __init__(self, eager, make_voxels, make_probs):
        super().__init__(eager)
        self._makers = {"voxels": make_voxels, "drc_probs": make_probs}
        for k in self._LAZY:
            dict.__setitem__(self, k, None)

    def _materialise(self, k):
        mk = self._makers.get(k)
        if mk is not None:
            self._makers[k] = None
            dict.__setitem__(self, k, mk())

    def __getitem__(self, k):
        if k in self._LAZY:
            self._materialise(k)
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        for k in self._LAZY:
            self._materialise(k)
        return dict.items(self)

    def values(self):
        for k in self._LAZY:
            self._materialise(k)
        return dict.values(self)

    # dict(out), {**out}, out.copy(): CPython copies a dict subclass's storage directly unless __iter__ is overridden;
    # with it the conversions go through __getitem__ and see the materialised entries
    def __iter__(self):
        return dict.__iter__(self)

    def copy(self):
        return dict(self.items())

    def pop(self, k, *default):
        if k in self._LAZY and k in self:
            self._materialise(k)
        return dict.pop(self, k, *default)


def pointcloud_project_fast(cfg, point_cloud, transform, predicted_translation,
                            all_rgb, kernel=None, scaling_factor=None, focal_length=None, *, point_dropout=None,
                            l2_target=None, views_per_cloud=None, silhouette_target=None):
    """dpc/util/point_cloud.py:229-290.

    ``point_dropout=(num_keep, seed)`` (keyword-only, not in the reference signature) fuses
    pc_point_dropout (point_cloud.py:293-319) into the projector: every instance keeps exactly
    ``num_keep`` of its N points, drawn without replacement from a permutation keyed by (seed,
    instance), inside the depth sort -- no [B,N',3] copy of the cloud, dropped points get a zero
    gradient.  ``tr_pc`` then still holds all N transformed points.  ``point_dropout=state`` with an
    int32 tensor ``{num_keep, seed}`` on the points' device makes the kernels read the pair at run time:
    a step recorded into a HIP graph then draws a new subset on every replay (the caller advances the
    tensor with enqueued work, see ModelPointCloud).

    ``l2_target=(gt, weight)`` (keyword-only) is the silhouette L2 loss of model_pc.py:414-415 fused into
    the collapse kernel: the result gains ``"proj_l2_grad"`` = weight * (proj - gt), the gradient of
    0.5 * weight * sum((proj - gt)^2) w.r.t. ``proj`` -- pass it to backward as the gradient of ``proj``
    (``torch.autograd.grad(out["proj"], inputs, out["proj_l2_grad"])``); gt is [B,D,D] or [B,D,D,1] at the
    projection's own size.

    ``views_per_cloud=R`` (keyword-only): ``point_cloud`` holds B / R clouds and instance b projects cloud b // R --
    the reference's ``tf_repeat_0`` replication over views and pose candidates (model_pc.py:23-32,270-279) as an
    index inside the kernels, so the [B,N,3] copies and their gradient reduction never exist; transform, translation,
    scaling factor and focal length stay per instance ([B, ...]).  Fused path only, no colour channels.

    ``silhouette_target=(masks, num_candidates, valid_samples | None)`` (keyword-only): the silhouette loss of
    model_pc.py:383-423 with the min over pose candidates of :308-337 evaluated inside the collapse kernels -- masks
    [B / C, S, S, 1] (S >= vox_size, resized bilinearly the TF1 way on the fly).  The result gains ``"proj_loss"`` (the
    scalar sum_g valid_g^2 min_c |gt_g - proj_gc|^2 / (2 B/C), differentiable: its gradient w.r.t. ``proj`` is formed
    inside the backward kernels, no dproj image exists), ``"winning_pose_candidates"`` [B / C] and ``"proj_inst_err"`` [B].
    Fused path, DRC collapse."""
    meta = _meta(cfg)
    if point_dropout is not None:
        if all_rgb is not None:
            raise NotImplementedError("fused point dropout with colour channels: use pc_point_dropout")
        if isinstance(point_dropout, torch.Tensor):
            meta = meta._replace(dropout_state=point_dropout)
        else:
            if int(point_dropout[0]) < 1:
                # the reference keeps int(N * keep_prob) points -- an empty cloud when that is 0; the kernels read
                # keep = 0 as "dropout off", so the caller's intent cannot be honoured silently
                raise ValueError("point_dropout would keep %d points" % int(point_dropout[0]))
            meta = meta._replace(dropout_keep=int(point_dropout[0]), dropout_seed=int(point_dropout[1]) & 0xffffffff)
    if l2_target is not None:
        meta = meta._replace(l2_target=l2_target[0].detach(), l2_weight=float(l2_target[1]))
    if views_per_cloud is not None and int(views_per_cloud) > 1:
        if all_rgb is not None:
            raise NotImplementedError("views_per_cloud with colour channels: replicate the cloud explicitly")
        if transform.shape[0] != point_cloud.shape[0] * int(views_per_cloud):
            raise ValueError("views_per_cloud=%d: %d clouds need %d poses, got %d" % (
                int(views_per_cloud), point_cloud.shape[0], point_cloud.shape[0] * int(views_per_cloud), transform.shape[0]))
        meta = meta._replace(views_per_cloud=int(views_per_cloud))
    if silhouette_target is not None:
        sgt, sC, sval = silhouette_target
        sgt = sgt.detach().to(device=point_cloud.device, dtype=torch.float32).contiguous()
        if sval is not None:
            sval = sval.detach().to(device=point_cloud.device, dtype=torch.float32).reshape(-1).contiguous()
        meta = meta._replace(sil_gt=sgt, sil_C=int(sC), sil_valid=sval)
    tx, ty, tz = _flat_taps(cfg, kernel, point_cloud.device)
    proj, proj_depth, tr_pc, l2_grad, sil_loss, sil_win, sil_err = ops.project_fused(
        point_cloud, transform, predicted_translation, scaling_factor, focal_length, tx, ty, tz, meta)
    state = {}

    def make_voxels():
        if meta.dropout_state is not None or 0 < meta.dropout_keep < point_cloud.shape[1]:
            raise NotImplementedError("'voxels' / 'drc_probs' of a projection with fused point dropout "
                                      "(the stage-level kernels see all N points): use pc_point_dropout")
        if "voxels" not in state:
            v, _ = pointcloud2voxels3d_fast(cfg, tr_pc, None)
            v = torch.clamp(v, 0.0, 1.0)
            if kernel is not None:
                v = ops.Blur3d.apply(v, tx, ty, tz)
            if scaling_factor is not None:
                v = torch.clamp(v * scaling_factor.reshape(-1, 1, 1, 1), 0.0, 1.0)
            state["voxels"] = v.unsqueeze(-1)
        return state["voxels"]

    def make_probs():
        if cfg.ptn_max_projection:
            return None
        if "probs" not in state:
            v = make_voxels()
            state["probs"], _ = _drc.drc_event_probabilities_impl(v, cfg, flip_h=True)   # tf.reverse(drc_probs, [2])
        return state["probs"]

    voxels_rgb = proj_rgb = None
    if all_rgb is not None:
        # colour channels (point_cloud.py:244-262,275-279; off by default, pc_rgb): stage-level kernels
        rgb_cm = _voxels_rgb_cm(cfg, tr_pc, all_rgb)                       # [B,3,Dz,D,D]
        if kernel is not None:
            if not cfg.pc_rgb_clip_after_conv:
                rgb_cm = torch.clamp(rgb_cm, 0.0, 1.0)
            rgb_cm = _convolve_cm(cfg, rgb_cm, kernel)
        if cfg.pc_rgb_divide_by_occupancies:
            div = ops.Voxelize.apply(tr_pc, meta.Dz, meta.D).detach()      # stop_gradient(voxels_raw)
            div = ops.Blur3d.apply(div, tx, ty, tz)
            rgb_cm = rgb_cm / (div.unsqueeze(1) + cfg.pc_rgb_divide_by_occupancies_epsilon)
        if cfg.pc_rgb_clip_after_conv:
            rgb_cm = torch.clamp(rgb_cm, 0.0, 1.0)
        voxels_rgb = torch.flip(rgb_cm.permute(0, 2, 3, 4, 1), dims=[2])   # tf.reverse(voxels_rgb, [2])
        probs = make_probs()
        proj_rgb = None if probs is None else _drc.project_volume_rgb_integral(cfg, probs, voxels_rgb)

    eager = {"proj": proj, "tr_pc": tr_pc, "voxels_rgb": voxels_rgb, "proj_rgb": proj_rgb, "proj_depth": proj_depth}
    if l2_target is not None:
        eager["proj_l2_grad"] = l2_grad
    if silhouette_target is not None:
        eager["proj_loss"], eager["winning_pose_candidates"], eager["proj_inst_err"] = sil_loss, sil_win, sil_err
    out = ProjectionOutputs(eager, make_voxels, make_probs)
    return out


def pc_point_dropout(points, rgb, keep_prob, generator=None):
    """dpc/util/point_cloud.py:293-319: keep int(N * keep_prob) points per instance,
    drawn without replacement, independently per instance (the reference calls
    np.random.choice inside a tf.py_func, i.e. a host round trip every step; here
    the draw is a device-side random permutation, no host sync).  The kept points
    come out in random order, as in the reference.  Returns (points, rgb)."""
    B, N = points.shape[0], points.shape[1]
    num_out = int(N * float(keep_prob))
    keys = torch.rand(B, N, device=points.device, generator=generator)
    idx = keys.argsort(dim=1)[:, :num_out]                       # [B, num_out] distinct indices
    gather = lambda t: torch.gather(t, 1, idx.unsqueeze(-1).expand(B, num_out, t.shape[2]))
    out_points = gather(points)
    out_rgb = gather(rgb) if rgb is not None else None
    return out_points, out_rgb
