"""torch.autograd plumbing over the C ABI (include/dpc_hip.h).

PyTorch is used for device memory, the current HIP stream and autograd
bookkeeping only; every number is produced by the hand-written kernels in
csrc/dpc_kernels.hip.  Tensors must be float32 on a ROCm device.
"""
import collections
import ctypes
import os

import torch

from . import _capi, _ext
from ._capi import DpcParams, DpcShape

ProjMeta = collections.namedtuple(
    "ProjMeta", "Dz D camera_distance focal_length eps max_depth pose_quaternion collapse_mode dropout_keep dropout_seed "
                "dropout_state l2_target l2_weight views_per_cloud sil_gt sil_C sil_valid",
    defaults=(0, 0, None, None, 0.0, 0, None, 1, None))
# dropout_state: int32[2] tensor {keep, seed} read by the kernels at run time (hipGraph replays)
# l2_target / l2_weight: [B,D,D(,1)] image and factor of the fused L2 loss epilogue (ProjectFused's 4th output)
# views_per_cloud: R > 1 = the point tensor holds B / R clouds, instance b projects cloud b // R (model_pc.py:270-279's
#                  replication as an index inside the kernels); the point gradient comes back summed per cloud
# sil_gt / sil_C / sil_valid: masks [B/C,S,S(,1)], candidates per (model, view) group, per-group weights | None: the
#                  candidate silhouette loss (model_pc.py:308-337, 383-423) evaluated inside the z kernels
#                  (ProjectFused's 5th..7th outputs: loss, winners, per-instance errors)


# ---------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------
def _lib_for(*tensors, dtypes=(torch.float32,)):
    lib = _capi.get_library()
    host = lib.host_memory
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError("expected a torch.Tensor, got %r" % type(t))
        if t.dtype not in dtypes:
            raise TypeError("the projector computes in float32; got %s" % t.dtype if len(dtypes) == 1 else
                            "expected one of %s, got %s" % (dtypes, t.dtype))
        if host:
            if t.is_cuda:
                raise ValueError("emulation library needs host tensors")
        elif not t.is_cuda:
            raise ValueError("the HIP projector needs tensors on a ROCm device (no CPU fallback)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError("tensors on different devices: %s" % sorted({str(x.device) for x in tensors if x is not None}))
    return lib


def _stream(lib, ref):
    if lib.host_memory:
        return None
    return ctypes.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)


def _stream_int(lib, device):
    """the current HIP stream of `device` as a plain integer (ctypes converts it to the void* argument)"""
    if lib.host_memory:
        return None
    return torch.cuda.current_stream(device).cuda_stream


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _c(t):
    # inside autograd.Function.forward/backward tensors are already outside the graph
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


def _shape(B, N, meta, K=(0, 0, 0)):
    return DpcShape(int(B), int(N), int(meta.Dz), int(meta.D), int(K[0]), int(K[1]), int(K[2]))


def _params(meta, l2_grad=None, sil=None):
    """sil = (gt_ptr, err_parts_ptr, weight_ptr, dloss_ptr, proj_ptr, C, S) of the fused candidate-loss epilogue"""
    if sil is not None:
        return DpcParams(float(meta.camera_distance), float(meta.focal_length), float(meta.eps),
                         float(meta.max_depth), 1 if meta.pose_quaternion else 0, int(meta.collapse_mode), 0,
                         int(meta.dropout_keep), int(meta.dropout_seed) & 0xffffffff,
                         None if meta.dropout_state is None else meta.dropout_state.data_ptr(),
                         None if l2_grad is None else meta.l2_target.data_ptr(),
                         None if l2_grad is None else l2_grad.data_ptr(), float(meta.l2_weight), int(meta.views_per_cloud),
                         *sil)
    return DpcParams(float(meta.camera_distance), float(meta.focal_length), float(meta.eps),
                     float(meta.max_depth), 1 if meta.pose_quaternion else 0, int(meta.collapse_mode), 0,
                     int(meta.dropout_keep), int(meta.dropout_seed) & 0xffffffff,
                     None if meta.dropout_state is None else meta.dropout_state.data_ptr(),
                     None if l2_grad is None else meta.l2_target.data_ptr(),
                     None if l2_grad is None else l2_grad.data_ptr(), float(meta.l2_weight), int(meta.views_per_cloud))


def _check_points(pc, pose, trans, scale, focal, meta):
    if pc.dim() != 3 or pc.shape[-1] != 3:
        raise ValueError("point_cloud must be [B,N,3], got %s" % (tuple(pc.shape),))
    B = pc.shape[0] * max(1, int(getattr(meta, "views_per_cloud", 0) or 1))
    if meta.pose_quaternion:
        if tuple(pose.shape) != (B, 4):
            raise ValueError("Can't create a quaternion from a tensor with shape %s. "
                             "The last dimension must be 4." % (tuple(pose.shape),))
    else:
        if tuple(pose.shape) != (B, 4, 4):
            raise ValueError("camera matrix must be [B,4,4], got %s" % (tuple(pose.shape),))
        if trans is not None:
            raise ValueError("predicted_translation requires a quaternion pose "
                             "(the reference's matrix branch cannot slice it)")
    if trans is not None and tuple(trans.shape) != (B, 3):
        raise ValueError("predicted_translation must be [B,3]")
    for name, t in (("scaling_factor", scale), ("focal_length", focal)):
        if t is not None and t.numel() != B:
            raise ValueError("%s must have B=%d elements, got %s" % (name, B, tuple(t.shape)))


def _taps_of(taps):
    """taps: None or (tx, ty, tz) 1-D float32 tensors (any may be None)."""
    if taps is None:
        return (None, None, None), (0, 0, 0)
    ts, ks = [], []
    for t in taps:
        if t is None:
            ts.append(None)
            ks.append(0)
            continue
        if t.dim() != 1 or not t.is_contiguous():
            t = _c(t).reshape(-1)
        k = t.numel()
        if k % 2 != 1:
            raise ValueError("even Gaussian kernel sizes are not supported (TF pads them asymmetrically)")
        if k > _capi.DPC_MAX_TAPS:
            raise ValueError("kernel size %d > %d" % (k, _capi.DPC_MAX_TAPS))
        ts.append(t)
        ks.append(k)
    return tuple(ts), tuple(ks)


def uses_fused_path(lib, B, N, meta, K):
    """True when this shape takes the fused front/back end (depth-sorted points, per-plane LDS tiles): the
    only path that honours the fused point dropout."""
    shape, params = _shape(B, N, meta, K), _params(meta._replace(views_per_cloud=0, sil_gt=None))
    return bool(lib.dpc_saved_layout(ctypes.byref(shape), ctypes.byref(params)) & 2)


def _poison_mode(v):
    """DPC_POISON_BUFFERS: unset / "0" off; "1" NaN / 0xff over every buffer the kernels must define; "2" the same, but 0x00 over
    the fused path's point index (whose tail holds the chunk marks): a 0xff-filled mark reads "every chunk is there", which
    hides a consumer of marks nobody wrote -- a 0x00-filled one reads "nothing is there" (round 6: the max-collapse path's
    k_gather_yx did exactly that).  The test-suite runs its chunk-sparse and parity cases under both."""
    v = (v or "").strip()
    return 0 if v in ("", "0") else (2 if v == "2" else 1)


_POISON = _poison_mode(os.environ.get("DPC_POISON_BUFFERS"))        # read once: this sits on the per-step host path


def _poison(t):
    """DPC_POISON_BUFFERS=1 (set by the test-suite before import): NaN-fill every buffer the kernels
    are supposed to fully define or deliberately skip, so that a read of a never-written element
    cannot hide."""
    if _POISON and t is not None and t.is_floating_point():
        t.fill_(float("nan"))
    return t


def _a256(n):
    return (n + 255) & ~255


class _FusedPlan(object):
    """Everything about one (library, B, N, grid, taps, collapse) combination that does not change from call to
    call: the DpcShape struct, which buffers the library wants saved (dpc_saved_layout), their byte offsets inside
    ONE arena allocation, and the workspace sizes.  ProjectFused runs every training step; asking the library
    three questions and building ~10 tensors per call was most of its host time."""
    __slots__ = ("shape", "shape_ref", "layout", "ws_fwd", "ws_bwd", "drc", "off_raw", "off_cmask", "off_pindex",
                 "off_blur", "off_sums", "arena_bytes", "n_out", "sil_parts")

    def __init__(self, lib, B, N, meta, K):
        self.shape = _shape(B, N, meta, K)
        self.shape_ref = ctypes.byref(self.shape)
        params = _params(meta._replace(l2_target=None, dropout_state=None, views_per_cloud=0, sil_gt=None))
        self.layout = lib.dpc_saved_layout(self.shape_ref, ctypes.byref(params))
        lib.check(min(self.layout, 0), "dpc_saved_layout")
        self.ws_fwd = lib.dpc_workspace_bytes(self.shape_ref, 0)
        self.ws_bwd = lib.dpc_workspace_bytes(self.shape_ref, 1)
        self.drc = meta.collapse_mode == _capi.DPC_COLLAPSE_DRC
        grid = 4 * B * meta.Dz * meta.D * meta.D
        off = 0
        self.off_raw = self.off_cmask = self.off_pindex = self.off_sums = -1
        if self.layout & 1:                                   # generic path: the dense pre-clip grid
            self.off_raw, off = off, off + _a256(grid)
        if self.layout & 2:                                   # fused path: clip bytes ...
            self.off_cmask, off = off, off + _a256(4 * B * N)
        if self.layout & 4:                                   # ... and the depth-sorted point records
            self.off_pindex, off = off, off + _a256(4 * lib.dpc_point_index_ints(self.shape_ref))
        self.off_blur, off = off, off + _a256(grid)
        if self.drc:
            self.off_sums, off = off, off + _a256(16 * B * meta.D * meta.D)      # [B,D,D,2] float64
        self.arena_bytes = off + 256
        self.n_out = 2 if self.drc else 1                     # proj (+ proj_depth)
        self.sil_parts = int(lib.dpc_sil_parts_per_view(self.shape_ref)) if self.drc else 0


_PLANS = {}


def _fused_plan(lib, B, N, meta, K):
    key = (id(lib), B, N, meta.Dz, meta.D, meta.collapse_mode, K)
    plan = _PLANS.get(key)
    if plan is None:
        if len(_PLANS) > 256:
            _PLANS.clear()
        plan = _PLANS[key] = _FusedPlan(lib, B, N, meta, K)
    return plan


def fused_plan_for(lib, B, N, meta, K):
    """the cached per-shape plan (which buffers are saved, whether the candidate-loss epilogue applies, ...)"""
    return _fused_plan(lib, B, N, meta._replace(l2_target=None, dropout_state=None, views_per_cloud=0, sil_gt=None,
                                                 sil_valid=None), K)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _at(base, off):
    return None if off < 0 else base + off


# ---------------------------------------------------------------------------
# fused hot path
# ---------------------------------------------------------------------------
class ProjectFused(torch.autograd.Function):
    """pointcloud_project_fast as ONE autograd node: (pc, pose, trans, scale,
    focal) -> (proj [B,D,D,1], proj_depth [B,D,D,1] | None, tr_pc [B,N,3], l2_grad [B,D,D,1] | None,
    sil_loss [] | None, sil_winners [B/C] int64 | None, sil_inst_err [B] | None).
    l2_grad (meta.l2_target set) = l2_weight * (proj - l2_target), written by the collapse kernel itself:
    the gradient of 0.5 * l2_weight * sum((proj - target)^2) w.r.t. proj, ready to be passed back as
    grad_outputs; not differentiable.
    sil_* (meta.sil_gt set): the candidate silhouette loss of model_pc.py:308-337 / 383-423 evaluated inside the
    collapse kernels; sil_loss is differentiable -- its gradient w.r.t. proj is formed inside the backward kernels.
    meta.views_per_cloud = R > 1: pc is [B/R,N,3] (instance b projects cloud b // R), dpc comes back per cloud.

    Host path: one cached plan per shape (_FusedPlan), a handful of device allocations in forward (tr_pc; one per
    image -- independent tensors, so in-place edits of an output are legal; one arena for everything saved for backward -- the kernels take raw pointers, so the arena is never cut
    into tensor views) plus the workspace, and pointers passed to ctypes as plain integers."""

    @staticmethod
    def forward(ctx, pc, pose, trans, scale, focal, tx, ty, tz, meta):
        _check_points(pc, pose, trans, scale, focal, meta)
        lib = _lib_for(pc, pose, trans, scale, focal, tx, ty, tz)
        pc, pose, trans, scale, focal = _c(pc), _c(pose), _c(trans), _c(scale), _c(focal)
        (tx, ty, tz), K = _taps_of((tx, ty, tz))
        R = max(1, int(meta.views_per_cloud or 1))
        B, N = pc.shape[0] * R, pc.shape[1]              # instances = clouds x views per cloud
        D = meta.D
        dev = pc.device
        plan = _fused_plan(lib, B, N, meta, K)
        if R > 1 and not plan.layout & 2:
            raise ValueError("views_per_cloud needs the fused path (vox_size a multiple of 4 in (16, 256], odd kernel size 3..31)")
        tgt = meta.l2_target
        if tgt is not None:
            if not plan.drc:
                raise ValueError("the fused L2 epilogue lives in the DRC collapse kernel (ptn_max_projection is off it)")
            _lib_for(pc, tgt)
            if tgt.numel() != B * D * D or tuple(tgt.shape[:3]) != (B, D, D) or not tgt.is_contiguous():
                raise ValueError("l2 target must be a contiguous [B,D,D] or [B,D,D,1] image, got %s for B=%d, D=%d"
                                 % (tuple(tgt.shape), B, D))
        st = meta.dropout_state
        if st is not None:
            if st.dtype != torch.int32 or st.numel() != 2 or not st.is_contiguous() or st.device != dev:
                raise ValueError("dropout state must be a contiguous int32 tensor {keep, seed} on the points' device")
        if (0 < meta.dropout_keep < N or st is not None) and not plan.layout & 2:
            raise ValueError("fused point dropout needs the fused path (vox_size a multiple of 4 in (16, 256], odd kernel size 3..31, vox_size_z <= 256); use pc_point_dropout for this shape")
        sgt = meta.sil_gt
        sil = None
        if sgt is not None:
            C = int(meta.sil_C)
            if not plan.sil_parts:
                raise ValueError("the fused candidate-loss epilogue needs the fused path with the DRC collapse")
            if C <= 0 or B % C != 0:
                raise ValueError("B=%d instances do not split into groups of %d pose candidates" % (B, C))
            if sgt.dim() not in (3, 4) or sgt.shape[0] != B // C or sgt.shape[1] != sgt.shape[2] or sgt.shape[1] < D \
                    or not sgt.is_contiguous() or sgt.dtype != torch.float32 or sgt.device != dev:
                raise ValueError("silhouette masks must be contiguous float32 [%d,S,S(,1)] with S >= %d on the points' device, got %s"
                                 % (B // C, D, tuple(sgt.shape)))
            sval = meta.sil_valid
            if sval is not None and (sval.numel() != B // C or sval.dtype != torch.float32 or not sval.is_contiguous()
                                     or sval.device != dev):
                raise ValueError("valid_samples must be %d contiguous float32 values on the points' device" % (B // C))
        # the images are separate allocations (not views of one): autograd refuses in-place edits of views made
        # inside a custom Function, and a caller holding `proj` must not pin `proj_depth`
        new_img = lambda: _poison(torch.empty(B, D, D, 1, dtype=torch.float32, device=dev))
        proj = new_img()
        depth = new_img() if plan.drc else None
        l2_grad = new_img() if tgt is not None else None
        tr_pc = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
        arena = torch.empty(plan.arena_bytes, dtype=torch.uint8, device=dev)
        work = torch.empty(plan.ws_fwd + 256, dtype=torch.uint8, device=dev)
        if _POISON:
            tr_pc.fill_(float("nan"))
            arena.fill_(255)                   # 0xffffffff is a NaN, 0xff..ff a NaN double
            work.fill_(255)
            if _POISON == 2 and plan.off_pindex >= 0:
                o = _a256(arena.data_ptr()) - arena.data_ptr() + plan.off_pindex
                arena[o:o + plan.off_blur - plan.off_pindex].zero_()
        if sgt is not None:
            err_parts = torch.empty(B, plan.sil_parts, dtype=torch.float32, device=dev)
            sil = (sgt.data_ptr(), err_parts.data_ptr(), None, None, None, int(meta.sil_C), int(sgt.shape[1]))
        params = _params(meta, l2_grad, sil)
        base = _a256(arena.data_ptr())
        rc = lib.dpc_project_forward(_stream_int(lib, dev), plan.shape_ref, ctypes.byref(params),
                                     pc.data_ptr(), pose.data_ptr(), _ptr(trans), _ptr(scale), _ptr(focal),
                                     _ptr(tx), _ptr(ty), _ptr(tz), tr_pc.data_ptr(), _at(base, plan.off_raw),
                                     _at(base, plan.off_cmask), _at(base, plan.off_pindex), base + plan.off_blur,
                                     _at(base, plan.off_sums), proj.data_ptr(), _ptr(depth),
                                     _a256(work.data_ptr()), plan.ws_fwd)
        lib.check(rc, "dpc_project_forward")
        sil_loss = sil_win = sil_err = sil_w = None
        if sgt is not None:
            # per-instance errors from the collapse kernel's partials, arg-min over the candidates, weights, loss
            C = int(meta.sil_C)
            sil_err = torch.empty(B, dtype=torch.float32, device=dev)
            sil_w = torch.empty(B, dtype=torch.float32, device=dev)
            win32 = torch.empty(B // C, dtype=torch.int32, device=dev)
            sil_loss = torch.empty((), dtype=torch.float32, device=dev)
            rc = lib.dpc_silhouette_select(_stream_int(lib, dev), B, C, plan.sil_parts, err_parts.data_ptr(),
                                           _ptr(meta.sil_valid), sil_err.data_ptr(), win32.data_ptr(), sil_w.data_ptr(),
                                           sil_loss.data_ptr())
            lib.check(rc, "dpc_silhouette_select")
            sil_win = win32.to(torch.int64)
        ctx.meta, ctx.K, ctx.plan = meta, K, plan
        ctx.set_materialize_grads(False)     # unused outputs (depth, tr_pc) arrive as None, not as zero fills
        ctx.scale_shape = None if scale is None else tuple(scale.shape)
        ctx.focal_shape = None if focal is None else tuple(focal.shape)
        ctx.save_for_backward(pc, pose, trans, scale, focal, tx, ty, tz, tr_pc, arena, sgt, sil_w,
                              proj if sgt is not None else None)
        if l2_grad is not None:
            ctx.mark_non_differentiable(l2_grad)
        if sgt is not None:
            ctx.mark_non_differentiable(sil_win, sil_err)
        return proj, depth, tr_pc, l2_grad, sil_loss, sil_win, sil_err

    @staticmethod
    def backward(ctx, dproj, ddepth, dtr, _dl2=None, dsil=None, _dwin=None, _derr=None):
        pc, pose, trans, scale, focal, tx, ty, tz, tr_pc, arena, sgt, sil_w, sil_proj = ctx.saved_tensors
        plan = ctx.plan
        meta = ctx.meta
        lib = _lib_for(pc)
        R = max(1, int(meta.views_per_cloud or 1))
        B, N = pc.shape[0] * R, pc.shape[1]
        dev = pc.device
        sil = None
        if sgt is not None and dsil is not None:
            # the candidate loss's gradient w.r.t. proj is formed inside the collapse VJP from (proj, masks, weights)
            dsil = _c(dsil.to(torch.float32))
            sil = (sgt.data_ptr(), None, sil_w.data_ptr(), dsil.data_ptr(), sil_proj.data_ptr(), int(meta.sil_C),
                   int(sgt.shape[1]))
        params = _params(meta if meta.l2_target is None else meta._replace(l2_target=None), None, sil)
        dproj, ddepth, dtr = _c(dproj), _c(ddepth), _c(dtr)
        if not plan.drc:
            ddepth = None
        if dproj is None and ddepth is None and sil is None:
            dproj = torch.zeros(B, meta.D, meta.D, 1, dtype=torch.float32, device=dev)
        new = lambda *s: _poison(torch.empty(*s, dtype=torch.float32, device=dev))
        dpc = new(B // R, N, 3)               # per cloud: the kernels sum a cloud's R instances
        dpose = _poison(torch.empty_like(pose))
        dtrans = new(B, 3) if trans is not None else None
        dscale = new(B) if scale is not None else None
        # the matrix branch never reads the per-instance focal length (point_cloud.py:191-205): no gradient
        dfocal = new(B) if (focal is not None and meta.pose_quaternion) else None
        work = torch.empty(plan.ws_bwd + 256, dtype=torch.uint8, device=dev)
        if _POISON:
            work.fill_(255)
        base = _a256(arena.data_ptr())
        rc = lib.dpc_project_backward(_stream_int(lib, dev), plan.shape_ref, ctypes.byref(params),
                                      pc.data_ptr(), pose.data_ptr(), _ptr(trans), _ptr(scale), _ptr(focal),
                                      _ptr(tx), _ptr(ty), _ptr(tz), tr_pc.data_ptr(), _at(base, plan.off_raw),
                                      _at(base, plan.off_cmask), _at(base, plan.off_pindex), base + plan.off_blur,
                                      _at(base, plan.off_sums), _ptr(dproj), _ptr(ddepth), _ptr(dtr),
                                      dpc.data_ptr(), dpose.data_ptr(), _ptr(dtrans), _ptr(dscale), _ptr(dfocal),
                                      _a256(work.data_ptr()), plan.ws_bwd)
        lib.check(rc, "dpc_project_backward")
        if dscale is not None:
            dscale = dscale.reshape(ctx.scale_shape)
        if dfocal is not None:
            dfocal = dfocal.reshape(ctx.focal_shape)
        return dpc, dpose, dtrans, dscale, dfocal, None, None, None, None


def project_fused(pc, pose, trans, scale, focal, tx, ty, tz, meta):
    """pointcloud_project_fast as ONE autograd node -> (proj, proj_depth | None, tr_pc, l2_grad | None, sil_loss | None,
    sil_winners | None, sil_inst_err | None): through the compiled binding (csrc/dpc_torch.cpp: the same checks, plan,
    allocations and C-ABI calls as ProjectFused below, in C++) when it is built, through ctypes (ProjectFused) otherwise.
    Both drive the same library; DPC_BINDING=ctypes forces the second."""
    ext = _ext.module()
    if ext is None:
        return ProjectFused.apply(pc, pose, trans, scale, focal, tx, ty, tz, meta)
    lib = _capi.get_library()
    lib_id = lib.__dict__.get("_ext_id")
    if lib_id is None:
        lib_id = lib.__dict__["_ext_id"] = ext.open_library(lib.path, lib.host_memory)
    try:
        out = ext.project_fused(pc, pose, trans, scale, focal, tx, ty, tz, meta.dropout_state, meta.l2_target, meta.sil_gt,
                                meta.sil_valid, lib_id, meta.Dz, meta.D, meta.camera_distance, meta.focal_length, meta.eps,
                                meta.max_depth, bool(meta.pose_quaternion), meta.collapse_mode, meta.dropout_keep,
                                meta.dropout_seed & 0xffffffff, meta.l2_weight, meta.views_per_cloud or 0, meta.sil_C or 1, _POISON)
    except RuntimeError as e:
        msg = str(e)
        if msg.startswith("DPC_RC:"):           # a status code of the C ABI: raise what the ctypes binding raises
            _, rc, what = msg.split("\n", 1)[0].split(":", 2)
            lib.check(int(rc), what.strip())
        raise
    return tuple(out)


# ---------------------------------------------------------------------------
# stage-level nodes (the reference's finer-grained API)
# ---------------------------------------------------------------------------
class Transform(torch.autograd.Function):
    """pc_perspective_transform (dpc/util/point_cloud.py:157-216)."""

    @staticmethod
    def forward(ctx, pc, pose, trans, focal, meta):
        _check_points(pc, pose, trans, None, focal, meta)
        lib = _lib_for(pc, pose, trans, focal)
        pc, pose, trans, focal = _c(pc), _c(pose), _c(trans), _c(focal)
        B, N = pc.shape[0], pc.shape[1]
        shape, params = _shape(B, N, meta), _params(meta)
        tr_pc = torch.empty_like(pc)
        rc = lib.dpc_transform_fwd(_stream(lib, pc), ctypes.byref(shape), ctypes.byref(params),
                                   _p(pc), _p(pose), _p(trans), _p(focal), _p(tr_pc))
        lib.check(rc, "dpc_transform_fwd")
        ctx.meta = meta
        ctx.focal_shape = None if focal is None else tuple(focal.shape)
        ctx.save_for_backward(pc, pose, trans, focal)
        return tr_pc

    @staticmethod
    def backward(ctx, dtr):
        pc, pose, trans, focal = ctx.saved_tensors
        lib = _lib_for(pc)
        B, N = pc.shape[0], pc.shape[1]
        shape, params = _shape(B, N, ctx.meta), _params(ctx.meta)
        dtr = _c(dtr)
        dpc = torch.empty_like(pc)
        dpose = torch.empty_like(pose)
        dtrans = torch.empty_like(trans) if trans is not None else None
        dfocal = (torch.empty(B, dtype=torch.float32, device=pc.device)
                  if (focal is not None and ctx.meta.pose_quaternion) else None)
        scratch = torch.empty(B * 16, dtype=torch.float32, device=pc.device)
        rc = lib.dpc_transform_bwd(_stream(lib, pc), ctypes.byref(shape), ctypes.byref(params),
                                   _p(pc), _p(pose), _p(trans), _p(focal), _p(dtr),
                                   _p(dpc), _p(dpose), _p(dtrans), _p(dfocal), _p(scratch))
        lib.check(rc, "dpc_transform_bwd")
        if dfocal is not None:
            dfocal = dfocal.reshape(ctx.focal_shape)
        return dpc, dpose, dtrans, dfocal, None


class Voxelize(torch.autograd.Function):
    """pointcloud2voxels3d_fast (dpc/util/point_cloud.py:60-136): [B,N,3] -> [B,Dz,D,D]."""

    @staticmethod
    def forward(ctx, tr_pc, Dz, D):
        if tr_pc.dim() != 3 or tr_pc.shape[-1] != 3:
            raise ValueError("pc must be [B,N,3]")
        lib = _lib_for(tr_pc)
        tr_pc = _c(tr_pc)
        B, N = tr_pc.shape[0], tr_pc.shape[1]
        shape = DpcShape(B, N, int(Dz), int(D), 0, 0, 0)
        grid = torch.empty(B, Dz, D, D, dtype=torch.float32, device=tr_pc.device)
        rc = lib.dpc_voxelize_fwd(_stream(lib, tr_pc), ctypes.byref(shape), _p(tr_pc), _p(grid))
        lib.check(rc, "dpc_voxelize_fwd")
        ctx.dims = (Dz, D)
        ctx.save_for_backward(tr_pc)
        return grid

    @staticmethod
    def backward(ctx, dgrid):
        (tr_pc,) = ctx.saved_tensors
        lib = _lib_for(tr_pc)
        Dz, D = ctx.dims
        B, N = tr_pc.shape[0], tr_pc.shape[1]
        shape = DpcShape(B, N, int(Dz), int(D), 0, 0, 0)
        dgrid = _c(dgrid)
        dtr = torch.empty_like(tr_pc)
        rc = lib.dpc_voxelize_bwd(_stream(lib, tr_pc), ctypes.byref(shape), _p(tr_pc), _p(dgrid), _p(dtr))
        lib.check(rc, "dpc_voxelize_bwd")
        return dtr, None, None


class VoxelizeValues(torch.autograd.Function):
    """RGB branch of pointcloud2voxels3d_fast (dpc/util/point_cloud.py:111-118):
    (tr_pc [B,N,3], values [B,N,C]) -> channel-major grid [B,C,Dz,D,D]."""

    @staticmethod
    def forward(ctx, tr_pc, values, Dz, D, stop_points_gradient):
        if values.dim() != 3 or values.shape[:2] != tr_pc.shape[:2]:
            raise ValueError("rgb must be [B,N,C] matching the point cloud")
        lib = _lib_for(tr_pc, values)
        tr_pc, values = _c(tr_pc), _c(values)
        B, N, C = values.shape
        shape = DpcShape(B, N, int(Dz), int(D), 0, 0, 0)
        grid = torch.empty(B, C, Dz, D, D, dtype=torch.float32, device=tr_pc.device)
        rc = lib.dpc_voxelize_values_fwd(_stream(lib, tr_pc), ctypes.byref(shape), C, _p(tr_pc), _p(values), _p(grid))
        lib.check(rc, "dpc_voxelize_values_fwd")
        ctx.dims = (Dz, D, bool(stop_points_gradient))
        ctx.save_for_backward(tr_pc, values)
        return grid

    @staticmethod
    def backward(ctx, dgrid):
        tr_pc, values = ctx.saved_tensors
        lib = _lib_for(tr_pc)
        Dz, D, stop = ctx.dims
        B, N, C = values.shape
        shape = DpcShape(B, N, int(Dz), int(D), 0, 0, 0)
        dgrid = _c(dgrid)
        dvals = torch.empty_like(values)
        dtr = None if stop else torch.empty_like(tr_pc)
        rc = lib.dpc_voxelize_values_bwd(_stream(lib, tr_pc), ctypes.byref(shape), C, _p(tr_pc), _p(values),
                                         _p(dgrid), _p(dvals), _p(dtr))
        lib.check(rc, "dpc_voxelize_values_bwd")
        return dtr, dvals, None, None, None


def _blur(lib, x, taps, K, order):
    B, Dz, D = x.shape[0], x.shape[1], x.shape[2]
    shape = DpcShape(B, 1, Dz, D, K[0], K[1], K[2])
    out = torch.empty_like(x)
    tmp = torch.empty_like(x)
    rc = lib.dpc_blur3d(_stream(lib, x), ctypes.byref(shape), _p(x), _p(out), _p(taps[0]), _p(taps[1]),
                        _p(taps[2]), _p(tmp), order)
    lib.check(rc, "dpc_blur3d")
    return out


class Blur3d(torch.autograd.Function):
    """smoothen_voxels3d, separable (dpc/util/point_cloud.py:139-145) on [B,Dz,D,D]."""

    @staticmethod
    def forward(ctx, vox, tx, ty, tz):
        if vox.dim() != 4 or vox.shape[2] != vox.shape[3]:
            raise ValueError("voxels must be [B,Dz,D,D]")
        lib = _lib_for(vox, tx, ty, tz)
        taps, K = _taps_of((tx, ty, tz))
        ctx.K = K
        ctx.save_for_backward(*[t for t in taps if t is not None])
        ctx.present = [t is not None for t in taps]
        return _blur(lib, _c(vox), taps, K, 0)

    @staticmethod
    def backward(ctx, dout):
        saved = list(ctx.saved_tensors)
        taps = [saved.pop(0) if p else None for p in ctx.present]
        dout = _c(dout)
        lib = _lib_for(dout)
        # order 1 = the adjoint: z first, every tap vector reversed inside the kernels (asymmetric filters too)
        return _blur(lib, dout, taps, ctx.K, 1), None, None, None


class DrcProjection(torch.autograd.Function):
    """drc_projection (dpc/util/drc.py:47-123): [B,Dz,D,D] -> proj [B,D,D], probs [Dz+1,B,D,D]."""

    @staticmethod
    def forward(ctx, vox, meta, flip_h):
        lib = _lib_for(vox)
        vox = _c(vox)
        B, Dz, D = vox.shape[0], vox.shape[1], vox.shape[2]
        shape, params = DpcShape(B, 1, Dz, D, 0, 0, 0), _params(meta)
        proj = torch.empty(B, D, D, dtype=torch.float32, device=vox.device)
        probs = torch.empty(Dz + 1, B, D, D, dtype=torch.float32, device=vox.device)
        rc = lib.dpc_drc_fwd(_stream(lib, vox), ctypes.byref(shape), ctypes.byref(params), _p(vox),
                             _p(proj), _p(probs), int(flip_h))
        lib.check(rc, "dpc_drc_fwd")
        ctx.meta, ctx.flip = meta, int(flip_h)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(vox)
        return proj, probs

    @staticmethod
    def backward(ctx, dproj, dprobs):
        (vox,) = ctx.saved_tensors
        lib = _lib_for(vox)
        B, Dz, D = vox.shape[0], vox.shape[1], vox.shape[2]
        shape, params = DpcShape(B, 1, Dz, D, 0, 0, 0), _params(ctx.meta)
        dproj, dprobs = _c(dproj), _c(dprobs)
        if dproj is None and dprobs is None:
            return torch.zeros_like(vox), None, None
        dvox = torch.empty_like(vox)
        rc = lib.dpc_drc_bwd(_stream(lib, vox), ctypes.byref(shape), ctypes.byref(params), _p(vox),
                             _p(dproj), _p(dprobs), _p(dvox), ctx.flip)
        lib.check(rc, "dpc_drc_bwd")
        return dvox, None, None


class MaxCollapse(torch.autograd.Function):
    """tf.reduce_max(voxels, [1]) (dpc/util/point_cloud.py:264-267)."""

    @staticmethod
    def forward(ctx, vox, flip_h):
        lib = _lib_for(vox)
        vox = _c(vox)
        B, Dz, D = vox.shape[0], vox.shape[1], vox.shape[2]
        shape = DpcShape(B, 1, Dz, D, 0, 0, 0)
        proj = torch.empty(B, D, D, dtype=torch.float32, device=vox.device)
        rc = lib.dpc_max_collapse_fwd(_stream(lib, vox), ctypes.byref(shape), _p(vox), _p(proj), int(flip_h))
        lib.check(rc, "dpc_max_collapse_fwd")
        ctx.flip = int(flip_h)
        ctx.save_for_backward(vox)
        return proj

    @staticmethod
    def backward(ctx, dproj):
        (vox,) = ctx.saved_tensors
        lib = _lib_for(vox)
        B, Dz, D = vox.shape[0], vox.shape[1], vox.shape[2]
        shape = DpcShape(B, 1, Dz, D, 0, 0, 0)
        dvox = torch.empty_like(vox)
        rc = lib.dpc_max_collapse_bwd(_stream(lib, vox), ctypes.byref(shape), _p(vox), _p(_c(dproj)),
                                      _p(dvox), ctx.flip)
        lib.check(rc, "dpc_max_collapse_bwd")
        return dvox, None


class SilhouetteLoss(torch.autograd.Function):
    """add_proj_loss / proj_loss_pose_candidates (dpc/models/model_pc.py:383-423, :308-337):
    proj [B,D,D,1] against gt [B/C,S,S,1] -> (loss [], winners [B/C] int64, inst_err [B])."""

    @staticmethod
    def forward(ctx, proj, gt, valid, num_candidates):
        lib = _lib_for(proj)
        C = int(num_candidates)
        if proj.dim() != 4 or proj.shape[1] != proj.shape[2] or proj.shape[3] != 1:
            raise ValueError("proj must be [B,D,D,1], got %s" % (tuple(proj.shape),))
        B, D = proj.shape[0], proj.shape[1]
        if C <= 0 or B % C != 0:
            raise ValueError("B=%d instances do not split into groups of %d pose candidates" % (B, C))
        if gt.dim() != 4 or gt.shape[0] != B // C or gt.shape[1] != gt.shape[2] or gt.shape[3] != 1:
            raise ValueError("gt must be [%d,S,S,1] (one mask per (model, view)), got %s" % (B // C, tuple(gt.shape)))
        if gt.shape[1] < D:
            raise ValueError("GT size should not be lower than the prediction size")
        # masks arrive as uint8 / bool / float64 from data pipelines: the kernels read float32 on proj's device
        gt = gt.to(device=proj.device, dtype=torch.float32)
        if valid is not None:
            if valid.numel() != B // C:
                raise ValueError("valid_samples must have %d elements, got %d" % (B // C, valid.numel()))
            valid = valid.to(device=proj.device, dtype=torch.float32)
        lib = _lib_for(proj, gt, valid)
        proj, gt = _c(proj), _c(gt)
        S = gt.shape[1]
        dev = proj.device
        if valid is not None:
            valid = _c(valid.reshape(-1))
        inst_err = torch.empty(B, dtype=torch.float32, device=dev)
        weight = torch.empty(B, dtype=torch.float32, device=dev)
        winners = torch.empty(B // C, dtype=torch.int32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        rc = lib.dpc_silhouette_loss_fwd(_stream(lib, proj), B, C, D, S, _p(proj), _p(gt), _p(valid), _p(inst_err),
                                         _p(winners), _p(weight), _p(loss))
        lib.check(rc, "dpc_silhouette_loss_fwd")
        ctx.dims = (B, C, D, S)
        ctx.save_for_backward(proj, gt, weight)
        winners = winners.to(torch.int64)
        ctx.mark_non_differentiable(winners, inst_err)
        return loss, winners, inst_err

    @staticmethod
    def backward(ctx, dloss, _dwin, _derr):
        proj, gt, weight = ctx.saved_tensors
        lib = _lib_for(proj)
        B, C, D, S = ctx.dims
        dproj = torch.empty_like(proj)
        rc = lib.dpc_silhouette_loss_bwd(_stream(lib, proj), B, C, D, S, _p(proj), _p(gt), _p(weight),
                                         _p(_c(dloss.to(torch.float32))), _p(dproj))
        lib.check(rc, "dpc_silhouette_loss_bwd")
        return dproj, None, None, None


class StudentLoss(torch.autograd.Function):
    """add_student_loss, default branch (dpc/models/model_pc.py:338-381): poses [n*C,4] (teachers, no gradient),
    winners [n] int64, student [n,4], weights [n] | None -> scalar loss; one kernel computes the loss and
    d loss / d student."""

    @staticmethod
    def forward(ctx, student, poses, winners, weights, num_candidates, scale):
        C = int(num_candidates)
        if student.dim() != 2 or student.shape[1] != 4:
            raise ValueError("student must be [n,4] quaternions, got %s" % (tuple(student.shape),))
        n = student.shape[0]
        if poses.numel() != n * C * 4:
            raise ValueError("poses must hold %d x %d candidate quaternions, got %s" % (n, C, tuple(poses.shape)))
        if winners.numel() != n or winners.dtype != torch.int64:
            raise ValueError("winners must be %d int64 candidate indices" % n)
        if weights is not None:
            if weights.numel() != n:
                raise ValueError("weights must have %d elements, got %d" % (n, weights.numel()))
            weights = _c(weights.to(device=student.device, dtype=torch.float32).reshape(-1))
        lib = _lib_for(student, poses, weights)
        _lib_for(winners, dtypes=(torch.int64,))
        student, poses, winners = _c(student), _c(poses), _c(winners)
        loss = torch.empty((), dtype=torch.float32, device=student.device)
        dstudent = torch.empty_like(student)
        rc = lib.dpc_student_loss(_stream(lib, student), n, C, _p(poses), _p(winners), _p(student), _p(weights),
                                  float(scale), _p(loss), _p(dstudent))
        lib.check(rc, "dpc_student_loss")
        ctx.save_for_backward(dstudent)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dstudent,) = ctx.saved_tensors
        return dstudent * dloss, None, None, None, None, None


class NNDistance(torch.autograd.Function):
    """point_cloud_distance (dpc/util/point_cloud_distance.py:26-39): Vs [Ns,3], Vt [Nt,3]
    (float32 or float64) -> (proj [Ns,3] = Vt[idx], minDist [Ns], idx [Ns] int32)."""

    @staticmethod
    def forward(ctx, vs, vt):
        lib = _lib_for(vs, vt, dtypes=(torch.float32, torch.float64))
        if vs.dtype != vt.dtype:
            raise TypeError("Vs and Vt must share a dtype, got %s and %s" % (vs.dtype, vt.dtype))
        vs, vt = _c(vs), _c(vt)
        ns, nt = vs.shape[0], vt.shape[0]
        proj = torch.empty(ns, 3, dtype=vs.dtype, device=vs.device)
        dist = torch.empty(ns, dtype=vs.dtype, device=vs.device)
        idx = torch.empty(ns, dtype=torch.int32, device=vs.device)
        rc = lib.dpc_nn_distance(_stream(lib, vs), vs.element_size(), ns, nt, _p(vs), _p(vt), _p(proj), _p(dist),
                                 _p(idx))
        lib.check(rc, "dpc_nn_distance")
        ctx.save_for_backward(vs, vt, proj, dist, idx)
        ctx.mark_non_differentiable(idx)
        return proj, dist, idx

    @staticmethod
    def backward(ctx, dproj, ddist, _didx):
        # what TF autodiff gives through gather_nd / sqrt(reduce_sum(diff^2)): tiny [Ns,3] torch glue
        vs, vt, proj, dist, idx = ctx.saved_tensors
        dvs = torch.zeros_like(vs)
        dsel = torch.zeros_like(proj)
        if dproj is not None:
            dsel = dsel + dproj
        if ddist is not None:
            unit = (proj - vs) / dist.unsqueeze(1)            # d dist / d vt[idx]; inf/nan at dist == 0, as in TF
            dsel = dsel + ddist.unsqueeze(1) * unit
            dvs = dvs - ddist.unsqueeze(1) * unit
        dvt = torch.zeros_like(vt).index_add_(0, idx.to(torch.int64), dsel)
        return dvs, dvt


class GaussVoxelize(torch.autograd.Function):
    """pointcloud2voxels (dpc/util/point_cloud.py:17-57): pc [B,N,3] -> clip(sum of
    per-point Gaussians on a G^3 lattice over [-1,1]^3, 0, 1) as [B,G,G,G]; output axis a
    takes point component perm[a].  O(N G^3): the reference's pc_fast:false path."""

    @staticmethod
    def forward(ctx, pc, sigma, G, perm, normalise):
        lib = _lib_for(pc)
        if pc.dim() != 3 or pc.shape[2] != 3:
            raise ValueError("point cloud must be [B,N,3], got %s" % (tuple(pc.shape),))
        pc = _c(pc)
        B, N, G = pc.shape[0], pc.shape[1], int(G)
        perm_c = (ctypes.c_int * 3)(*[int(p) for p in perm])
        dev = pc.device
        raw = torch.empty(B, G, G, G, dtype=torch.float32, device=dev)
        vox = torch.empty(B, G, G, G, dtype=torch.float32, device=dev)
        inv = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if normalise == 1 else None
        rc = lib.dpc_gauss_voxelize_fwd(_stream(lib, pc), B, N, G, perm_c, float(sigma), int(normalise), _p(pc),
                                        _p(inv), _p(raw), _p(vox))
        lib.check(rc, "dpc_gauss_voxelize_fwd")
        ctx.args = (B, N, G, tuple(int(p) for p in perm), float(sigma), int(normalise))
        ctx.save_for_backward(pc, raw)
        return vox

    @staticmethod
    def backward(ctx, dvox):
        pc, raw = ctx.saved_tensors
        lib = _lib_for(pc)
        B, N, G, perm, sigma, normalise = ctx.args
        perm_c = (ctypes.c_int * 3)(*perm)
        nbytes = lib.dpc_gauss_voxelize_workspace_bytes(B, G)
        ws = _poison(torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=pc.device))
        dpc = torch.empty_like(pc)
        rc = lib.dpc_gauss_voxelize_bwd(_stream(lib, pc), B, N, G, perm_c, sigma, normalise, _p(pc), _p(raw),
                                        _p(_c(dvox)), _p(dpc), _p(ws), nbytes)
        lib.check(rc, "dpc_gauss_voxelize_bwd")
        return dpc, None, None, None, None

