"""Eager, torch-CPU backed stand-in for the ~45 ``tf.*`` symbols that the
reference's five hot-path modules use (dpc/util/{point_cloud,drc,gauss_kernel,
quaternion,camera}.py).

TEST INFRASTRUCTURE, used in the build container only, by
``tests/golden/make_goldens.py``: it lets the reference's *own, unmodified*
Python source run end to end so that its control flow, axis conventions and
quirks define the golden vectors.  Only the leaf-op semantics are restated
here (from the TensorFlow 1.x documentation; TensorFlow itself is not
installed and cannot be): scatter_nd sums duplicates, conv3d is a SAME
zero-padded cross-correlation, clip_by_value passes gradient on the closed
interval, cumsum is inclusive.  Parity is therefore pinned to the reference
source, NOT to a TensorFlow binary -- see DESIGN.md "Oracle".

Nothing under differentiable-point-clouds_amd/ imports this; it never ships
to the GPU box as part of a measured or product path.
"""
import builtins as _bi

import numpy as _np
import torch as _torch

# ---------------------------------------------------------------------------
# dtypes: ``tf.float32`` is a late-bound sentinel so that one import of the
# reference modules (drc.py binds DTYPE = tf.float32 at import time) can be
# run in fp32 (the reference's arithmetic) and in fp64 (error budgeting).
# ---------------------------------------------------------------------------
_FLOAT = _torch.float32


def set_float_dtype(dt):
    """Select what ``tf.float32`` means for subsequent ops (float32|float64)."""
    global _FLOAT
    assert dt in (_torch.float32, _torch.float64)
    _FLOAT = dt


class _DType(object):
    def __init__(self, name):
        self.name = name

    def resolve(self):
        if self.name == "float32":
            return _FLOAT
        return {"int32": _torch.int32, "int64": _torch.int64,
                "bool": _torch.bool, "float64": _torch.float64}[self.name]

    def __repr__(self):
        return "tf." + self.name


float32 = _DType("float32")
float64 = _DType("float64")
int32 = _DType("int32")
int64 = _DType("int64")


def _dt(d):
    if d is None:
        return None
    if isinstance(d, _DType):
        return d.resolve()
    return d


class _Shape(list):
    """list-like static shape with TF's ``as_list``."""

    def as_list(self):
        return list(self)


class Tensor(_torch.Tensor):
    """torch.Tensor whose augmented assignments REBIND (TF graph semantics:
    ``q /= n`` builds a new node; it must not mutate a leaf or a view)."""

    @staticmethod
    def __new__(cls, data):
        return _torch.Tensor._make_subclass(cls, data, data.requires_grad)

    @property
    def shape(self):
        return _Shape(_torch.Tensor.size(self))

    def __iadd__(self, o):
        return self + o

    def __isub__(self, o):
        return self - o

    def __imul__(self, o):
        return self * o

    def __itruediv__(self, o):
        return self / o


def _wrap(t):
    if isinstance(t, Tensor):
        return t
    return t.as_subclass(Tensor)


def convert_to_tensor(x, dtype=None):
    if isinstance(x, _torch.Tensor):
        t = x
        if dtype is not None:
            t = t.to(_dt(dtype))
        return _wrap(t)
    a = _np.asarray(x)
    if dtype is not None:
        return _wrap(_torch.as_tensor(a).to(_dt(dtype)))
    if a.dtype.kind == "f":
        return _wrap(_torch.as_tensor(a).to(_FLOAT))
    if a.dtype.kind in "iu":
        return _wrap(_torch.as_tensor(a.astype(_np.int64)).to(_torch.int32))
    return _wrap(_torch.as_tensor(a))


def constant(value, dtype=None, shape=None):
    t = convert_to_tensor(value, dtype)
    if shape is not None:
        t = _wrap(t.reshape(-1).expand(int(_np.prod(shape))).reshape(list(shape)).clone())
    return t


def _t(x, like=None):
    """tensor-ify python scalars / lists for binary ops."""
    if isinstance(x, _torch.Tensor):
        return x
    if like is not None:
        return _torch.as_tensor(x, dtype=like.dtype)
    return convert_to_tensor(x)


def cast(x, dtype):
    return _wrap(_t(x).to(_dt(dtype)))


def floor(x):
    return _wrap(_torch.floor(x))


def range(start, limit=None, delta=1, dtype=None):  # noqa: A001 (tf name)
    if limit is None:
        start, limit = 0, start
    f = lambda v: v.item() if isinstance(v, _torch.Tensor) else v
    start, limit, delta = f(start), f(limit), f(delta)
    if dtype is None:
        isf = any(isinstance(v, float) for v in (start, limit, delta))
        dtype = float32 if isf else int32
    return _wrap(_torch.arange(start, limit, delta, dtype=_dt(dtype)))


def linspace(a, b, n):
    return _wrap(_torch.linspace(a, b, n, dtype=_FLOAT))


def meshgrid(*xs):
    # tf.meshgrid defaults to indexing='xy'
    return [_wrap(g) for g in _torch.meshgrid(*xs, indexing="xy")]


def expand_dims(x, axis):
    return _wrap(_torch.unsqueeze(_t(x), axis))


def tile(x, multiples):
    return _wrap(_t(x).repeat(*[int(m) for m in multiples]))


def concat(values, axis):
    return _wrap(_torch.cat([_t(v) for v in values], dim=axis))


def reshape(x, shape):
    return _wrap(_torch.reshape(_t(x), [int(s) for s in shape]))


def shape(x):
    return [int(s) for s in _t(x).size()]


def slice(x, begin, size):  # noqa: A001
    idx = []
    for b, s, n in zip(begin, size, x.size()):
        idx.append(_bi.slice(b, n if s == -1 else b + s))
    return _wrap(x[tuple(idx)])


def pad(x, paddings, mode="CONSTANT", constant_values=0):
    assert mode == "CONSTANT"
    p = _np.asarray(paddings.cpu().numpy() if isinstance(paddings, _torch.Tensor) else paddings)
    flat = []
    for lo, hi in p[::-1]:  # torch pads last dim first
        flat += [int(lo), int(hi)]
    return _wrap(_torch.nn.functional.pad(x, flat, value=constant_values))


def stack(values, axis=0):
    return _wrap(_torch.stack(list(values), dim=axis))


def unstack(x, axis=0):
    return [_wrap(t) for t in _torch.unbind(x, dim=axis)]


def squeeze(x, axis=None):
    return _wrap(_torch.squeeze(x) if axis is None else _torch.squeeze(x, axis))


def transpose(x, perm=None):
    if perm is None:
        perm = list(reversed(_bi.range(x.dim())))
    return _wrap(x.permute(*perm))


def reverse(x, axis):
    return _wrap(_torch.flip(x, dims=list(axis)))


def boolean_mask(x, mask):
    return _wrap(x[mask])


def scatter_nd(indices, updates, shape):
    """Duplicate indices accumulate.  Out-of-range rows are dropped (TF GPU
    behaviour; TF CPU raises -- only reachable with a coordinate exactly +0.5,
    whose weight is 0)."""
    shp = [int(s) for s in shape]
    nd = indices.size(-1)
    out = _torch.zeros(shp, dtype=updates.dtype)
    idx = indices.to(_torch.int64)
    ok = _torch.ones(idx.size(0), dtype=_torch.bool)
    for d in _bi.range(nd):
        ok &= (idx[:, d] >= 0) & (idx[:, d] < shp[d])
    idx = idx[ok]
    upd = updates[ok]
    out = out.index_put(tuple(idx[:, d] for d in _bi.range(nd)), upd, accumulate=True)
    return _wrap(out)


def gather_nd(params, indices):
    idx = indices.to(_torch.int64)
    nd = idx.size(-1)
    return _wrap(params[tuple(idx[..., d] for d in _bi.range(nd))])


def add_n(xs):
    out = xs[0]
    for x in xs[1:]:
        out = out + x
    return _wrap(out)


def logical_and(a, b):
    return _wrap(_torch.logical_and(a, b))


def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return [int(a) for a in axis]
    return int(axis)


def reduce_all(x, axis=None, keepdims=False, keep_dims=False):
    k = keepdims or keep_dims
    if axis is None:
        return _wrap(_torch.all(x))
    return _wrap(_torch.all(x, dim=_axes(axis), keepdim=k))


def reduce_sum(x, axis=None, keepdims=False, keep_dims=False):
    k = keepdims or keep_dims
    if axis is None:
        return _wrap(_torch.sum(x))
    return _wrap(_torch.sum(x, dim=_axes(axis), keepdim=k))


def reduce_max(x, axis=None, keepdims=False, keep_dims=False):
    k = keepdims or keep_dims
    if axis is None:
        return _wrap(_torch.max(x))
    return _wrap(_torch.amax(x, dim=_axes(axis), keepdim=k))


def square(x):
    return _wrap(x * x)


def pow(x, y):  # noqa: A001
    return _wrap(_torch.pow(_t(x), y))


def sqrt(x):
    return _wrap(_torch.sqrt(x))


def exp(x):
    return _wrap(_torch.exp(_t(x)))


def log(x):
    return _wrap(_torch.log(_t(x)))


def multiply(a, b):
    a = _t(a)
    return _wrap(a * _t(b, like=a))


def add(a, b):
    a = _t(a)
    return _wrap(a + _t(b, like=a))


def norm(x, axis=None, keepdims=False, keep_dims=False):
    k = keepdims or keep_dims
    return _wrap(_torch.sqrt(_torch.sum(x * x, dim=_axes(axis), keepdim=k)))


def matmul(a, b):
    return _wrap(_torch.matmul(a, b))


def clip_by_value(x, lo, hi):
    """TF1: maximum(minimum(x, hi), lo); Minimum/Maximum gradients use
    less_equal / greater_equal, i.e. the gradient passes on lo <= x <= hi
    (closed).  torch.clamp's backward uses the same closed mask."""
    return _wrap(_torch.clamp(x, min=float(lo), max=float(hi)))


def stop_gradient(x):
    return _wrap(x.detach())


def ones(shape, dtype=float32):
    return _wrap(_torch.ones([int(s) for s in shape], dtype=_dt(dtype)))


def Variable(initial_value, name=None, dtype=None, trainable=True):  # noqa: N802 (tf name)
    return _wrap(_torch.as_tensor(initial_value, dtype=_dt(dtype) if dtype is not None else _FLOAT))


def zeros(shape, dtype=float32):
    return _wrap(_torch.zeros([int(s) for s in shape], dtype=_dt(dtype)))


def ones_like(x, dtype=None):
    return _wrap(_torch.ones_like(x, dtype=_dt(dtype)))


def cumsum(x, axis=0):
    return _wrap(_torch.cumsum(x, dim=axis))


def cumprod(x, axis=0):
    return _wrap(_torch.cumprod(x, dim=axis))


def py_func(func, inp, Tout):
    out = func(*[i.detach().cpu().numpy() if isinstance(i, _torch.Tensor) else i for i in inp])
    return convert_to_tensor(out, Tout)


class _NN(object):
    @staticmethod
    def conv3d(x, filt, strides, padding="SAME"):
        """NDHWC input, [kd,kh,kw,cin,cout] filter, cross-correlation, SAME
        zero padding (odd sizes only: even sizes pad asymmetrically in TF)."""
        assert padding == "SAME" and list(strides) == [1, 1, 1, 1, 1]
        kd, kh, kw, cin, cout = [int(s) for s in filt.size()]
        assert kd % 2 == 1 and kh % 2 == 1 and kw % 2 == 1
        xi = x.permute(0, 4, 1, 2, 3)
        w = filt.permute(4, 3, 0, 1, 2).to(x.dtype)
        y = _torch.nn.functional.conv3d(xi, w, padding=(kd // 2, kh // 2, kw // 2))
        return _wrap(y.permute(0, 2, 3, 4, 1))

    @staticmethod
    def depthwise_conv2d(x, filt, strides, padding="SAME"):
        assert padding == "SAME" and list(strides) == [1, 1, 1, 1]
        kh, kw, cin, mult = [int(s) for s in filt.size()]
        assert kh % 2 == 1 and kw % 2 == 1
        xi = x.permute(0, 3, 1, 2)
        w = filt.permute(2, 3, 0, 1).reshape(cin * mult, 1, kh, kw).to(x.dtype)
        y = _torch.nn.functional.conv2d(xi, w, padding=(kh // 2, kw // 2), groups=cin)
        return _wrap(y.permute(0, 2, 3, 1))


nn = _NN()


# ---------------------------------------------------------------------------
# extras needed to import dpc/models/model_pc.py (caller rows C1/C2)
# ---------------------------------------------------------------------------
def to_float(x):
    return _wrap(_t(x).to(_FLOAT) if isinstance(x, _torch.Tensor) else _torch.tensor(float(x), dtype=_FLOAT))


def to_int32(x):
    return _wrap(_t(x).to(_torch.int32))


def argmin(x, axis=None):
    return _wrap(_torch.argmin(x, dim=axis))


def one_hot(indices, depth):
    return _wrap(_torch.nn.functional.one_hot(indices.to(_torch.int64), int(depth)).to(_FLOAT))


def sigmoid(x):
    return _wrap(_torch.sigmoid(x))


def reduce_mean(x, axis=None, keepdims=False):
    return _wrap(_torch.mean(x) if axis is None else _torch.mean(x, dim=_axes(axis), keepdim=keepdims))


def equal(a, b):
    return _wrap(_t(a) == b)


def not_equal(a, b):
    return _wrap(_t(a) != b)


def less(a, b):
    return _wrap(_t(a) < _t(b))


def where(c, a, b):
    return _wrap(_torch.where(c, a, b))


def _resize_bicubic_legacy(images, oh, ow):
    """ResizeBicubic, align_corners=False, legacy scaler (tensorflow/core/kernels/resize_bicubic_op.cc, r1.x), pixel by
    pixel as the op's reference loop: in = out * (in_size / out_size) [float]; taps floor(in) - 1 .. + 2 clamped; weights
    from the 1024-entry table of the A = -0.75 cubic kernel at lrintf(frac * 1024); rows first along x, then along y."""
    import numpy as _np
    a = -0.75
    tab = _np.zeros((1025, 2), _np.float32)
    for i in range(1025):
        x = _np.float32(i * 1.0 / 1024)
        tab[i, 0] = ((a + 2) * float(x) - (a + 3)) * float(x) * float(x) + 1
        x = _np.float32(x + _np.float32(1.0))
        tab[i, 1] = ((a * float(x) - 5 * a) * float(x) + 8 * a) * float(x) - 4 * a

    def taps(o, i):
        scale = _np.float32(i) / _np.float32(o)
        out = []
        for k in range(o):
            loc = _np.float32(k) * scale
            lo = int(_np.floor(loc))
            off = int(_np.rint(_np.float32(loc - _np.float32(lo)) * _np.float32(1024)))
            w = (tab[off, 1], tab[off, 0], tab[1024 - off, 0], tab[1024 - off, 1])
            idx = tuple(min(max(lo + d, 0), i - 1) for d in (-1, 0, 1, 2))
            out.append((idx, w))
        return out
    src = images.detach().cpu().numpy().astype(_np.float32)
    n, ih, iw, c = src.shape
    ty, tx = taps(oh, ih), taps(ow, iw)
    res = _np.zeros((n, oh, ow, c), _np.float32)
    for y, (yi, yw) in enumerate(ty):
        for x, (xi, xw) in enumerate(tx):
            rows = []
            for r in yi:
                v = src[:, r, xi[0]] * xw[0] + src[:, r, xi[1]] * xw[1] + src[:, r, xi[2]] * xw[2] + src[:, r, xi[3]] * xw[3]
                rows.append(v.astype(_np.float32))
            res[:, y, x] = rows[0] * yw[0] + rows[1] * yw[1] + rows[2] * yw[2] + rows[3] * yw[3]
    return _torch.tensor(res)


class _Image(object):
    class ResizeMethod(object):
        BILINEAR = 0
        BICUBIC = 2

    @staticmethod
    def resize_images(images, size, method=0):
        """TF1 tf.image.resize_images, align_corners=False (legacy): source
        coordinate = dst * (in/out), no half-pixel offset, bilinear."""
        n, ih, iw, c = [int(s) for s in images.size()]
        oh, ow = int(size[0]), int(size[1])
        if method == 2:
            return _wrap(_resize_bicubic_legacy(images, oh, ow))
        assert method == 0, "bilinear or bicubic"

        def axis(o, i):
            src = _torch.arange(o, dtype=_torch.float64) * (i / o)
            lo = _torch.floor(src).to(_torch.int64)
            hi = _torch.clamp(lo + 1, max=i - 1)
            return lo, hi, (src - lo.to(_torch.float64)).to(images.dtype)
        ylo, yhi, yl = axis(oh, ih)
        xlo, xhi, xl = axis(ow, iw)
        top = images[:, ylo][:, :, xlo] * (1 - xl).view(1, 1, -1, 1) + images[:, ylo][:, :, xhi] * xl.view(1, 1, -1, 1)
        bot = images[:, yhi][:, :, xlo] * (1 - xl).view(1, 1, -1, 1) + images[:, yhi][:, :, xhi] * xl.view(1, 1, -1, 1)
        return _wrap(top * (1 - yl).view(1, -1, 1, 1) + bot * yl.view(1, -1, 1, 1))


image = _Image()


def _l2_loss(x):
    return _wrap((x * x).sum() / 2)


_NN.l2_loss = staticmethod(_l2_loss)


class _Noop(object):
    """tf.contrib.{slim,summary}: import-time attribute access and summary
    calls only; nothing on the projector/loss path computes through them."""

    def __getattr__(self, name):
        return _Noop()

    def __call__(self, *a, **k):
        return None


contrib = _Noop()


def truncated_normal_initializer(*a, **k):
    return None
