"""NumPy restatement (float64 by default) of the reference projector, forward
AND explicit backward.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module, and only as the
checker.  Nothing under ``differentiable-point-clouds_amd/`` imports it.

Parity status: the reference (eldar/differentiable-point-clouds) ships no
tests and no golden vectors, and its arithmetic is defined by TensorFlow 1.x
ops (``TensorFlow >= 1.11``, unpinned, not installable here).  This
restatement is pinned against the reference's OWN SOURCE executed unchanged
under the eager shim in ``oracle/tf_shim`` (``tests/golden/make_goldens.py``
-> ``tests/golden/*.npz``); TF leaf-op semantics themselves are restated from
documentation => "parity pinned to reference source, unpinned against a TF
binary".

Written independently of ``oracle/reference_cpu.py`` (which mirrors the
reference graph op for op in torch): this file follows the mathematical
specification and derives the backward pass by hand, which is what the HIP
kernels implement.

Reference lines followed (relative to /root/reference):
  transform   dpc/util/point_cloud.py:157-216, dpc/util/quaternion.py:62-117
  voxelise    dpc/util/point_cloud.py:60-136
  blur        dpc/util/point_cloud.py:139-145, dpc/util/gauss_kernel.py:5-54
  orchestr.   dpc/util/point_cloud.py:229-290
  DRC         dpc/util/drc.py:47-123, depth :139-153
"""
import numpy as np


# --------------------------------------------------------------------------
# Gaussian taps                                    dpc/util/gauss_kernel.py:5-11
# --------------------------------------------------------------------------
def gauss_kernel_1d(size, sigma, dtype=np.float64):
    """Taps at integer offsets range(-size//2+1., size//2+1.), normalised."""
    size = int(size)
    dtype = np.dtype(dtype)
    if size % 2 != 1:
        raise ValueError("only odd kernel sizes are supported (reference uses 11/21)")
    xx = np.arange(-size // 2 + 1.0, size // 2 + 1.0, dtype=dtype)
    k = np.exp(-xx ** 2 / (2.0 * dtype.type(sigma) ** 2))
    return (k / k.sum()).astype(dtype)


def smoothing_taps(vox_size, vox_size_z, ksize, sigma, dtype=np.float64):
    """[taps_x, taps_y, taps_z]                 dpc/util/gauss_kernel.py:35-54"""
    k = gauss_kernel_1d(ksize, sigma, dtype)
    if vox_size_z != -1:
        ratio = vox_size_z / vox_size
        kz = int(np.floor(ksize * ratio))
        if kz % 2 == 0:
            kz += 1
        return [k, k, gauss_kernel_1d(kz, sigma * ratio, dtype)]
    return [k, k, k.copy()]


# --------------------------------------------------------------------------
# quaternion helpers (w,x,y,z)                    dpc/util/quaternion.py:62-83
# --------------------------------------------------------------------------
def _qmul(a, b):
    w1, x1, y1, z1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    w2, x2, y2, z2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
        w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2], axis=-1)


def _qconj(q):
    return q * np.array([1.0, -1.0, -1.0, -1.0], dtype=q.dtype)


# --------------------------------------------------------------------------
# A2/A3  perspective transform                 dpc/util/point_cloud.py:157-216
# --------------------------------------------------------------------------
def transform_fwd(pc, pose, trans=None, focal=None, camera_distance=2.0,
                  focal_length=1.875, pose_quaternion=True):
    """pc [B,N,3]; pose [B,4] (w,x,y,z, unnormalised) or [B,4,4];
    trans [B,3]|None; focal [B,1]|None.  Returns tr_pc [B,N,3] = (depth,y,x)."""
    dt = pc.dtype
    B, N, _ = pc.shape
    cd = dt.type(camera_distance)
    if pose_quaternion:
        f = np.full((B, 1), focal_length, dt) if focal is None else focal.reshape(B, 1).astype(dt)
        qn = pose / np.sqrt((pose * pose).sum(-1, keepdims=True))
        P = np.concatenate([np.zeros((B, N, 1), dt), pc], axis=-1)
        r = _qmul(_qmul(qn[:, None, :], P), _qconj(qn)[:, None, :])
        p2 = r[..., 1:4]
        if trans is not None:
            p2 = p2 + trans[:, None, :]
        zs = p2[..., 0] + cd
        xs = p2[..., 2] * f
        ys = p2[..., 1] * f
        xs = xs / zs
        ys = ys / zs
        zs = zs - cd
        if trans is not None:
            zs = zs - trans[:, None, 0]
    else:
        if trans is not None:
            raise ValueError("translation is only supported with quaternion poses (reference :211-213)")
        intr = np.eye(4, dtype=dt)
        intr[1, 1] = intr[2, 2] = focal_length          # dpc/util/camera.py:5-13
        M = intr[None] @ pose
        xyz1 = np.concatenate([pc, np.ones((B, N, 1), dt)], axis=-1)
        p2 = xyz1 @ np.transpose(M, (0, 2, 1))
        zs = p2[..., 0]
        xs = p2[..., 2] / zs
        ys = p2[..., 1] / zs
        zs = zs - cd
    return np.stack([zs, ys, xs], axis=-1)


def transform_bwd(pc, pose, trans, focal, d_tr, camera_distance=2.0,
                  focal_length=1.875, pose_quaternion=True):
    """Hand-derived VJP of transform_fwd.  Returns dict(dpc,dpose,dtrans,dfocal)."""
    dt = pc.dtype
    B, N, _ = pc.shape
    cd = dt.type(camera_distance)
    dw, dv, du = d_tr[..., 0], d_tr[..., 1], d_tr[..., 2]
    if pose_quaternion:
        f = np.full((B, 1), focal_length, dt) if focal is None else focal.reshape(B, 1).astype(dt)
        nrm = np.sqrt((pose * pose).sum(-1, keepdims=True))
        qn = pose / nrm
        P = np.concatenate([np.zeros((B, N, 1), dt), pc], axis=-1)
        t = _qmul(qn[:, None, :], P)
        r = _qmul(t, _qconj(qn)[:, None, :])
        p2 = r[..., 1:4]
        if trans is not None:
            p2 = p2 + trans[:, None, :]
        Z = p2[..., 0] + cd
        x, y = p2[..., 2], p2[..., 1]
        u = f * x / Z
        v = f * y / Z
        # w = Z - cd - t0 ; u = f x / Z ; v = f y / Z
        dx = du * f / Z
        dy = dv * f / Z
        dZ = -(du * u + dv * v) / Z
        dfocal = ((du * x + dv * y) / Z).sum(-1, keepdims=True)
        dp2 = np.stack([dw + dZ, dy, dx], axis=-1)
        dtrans = None
        if trans is not None:
            dtrans = np.stack([dZ, dy, dx], axis=-1).sum(1)       # dw cancels for t0
        # reverse of r = t (x) q*, t = q (x) P.   <dc, a(x)b>: da = dc(x)b*, db = a*(x)dc
        dr = np.concatenate([np.zeros((B, N, 1), dt), dp2], axis=-1)
        qb = np.broadcast_to(qn[:, None, :], dr.shape)
        dtq = _qmul(dr, qb)                       # dt  = dr (x) conj(q*) = dr (x) q
        dqc = _qmul(_qconj(t), dr)                # d(q*) = t* (x) dr
        dq_hat = _qconj(dqc) + _qmul(dtq, _qconj(P))
        dP = _qmul(_qconj(qb), dtq)
        dpc = dP[..., 1:4]
        dq_hat = dq_hat.sum(1)
        dpose = (dq_hat - qn * (qn * dq_hat).sum(-1, keepdims=True)) / nrm
        return dict(dpc=dpc, dpose=dpose, dtrans=dtrans, dfocal=dfocal)
    intr = np.eye(4, dtype=dt)
    intr[1, 1] = intr[2, 2] = focal_length
    M = intr[None] @ pose
    xyz1 = np.concatenate([pc, np.ones((B, N, 1), dt)], axis=-1)
    p2 = xyz1 @ np.transpose(M, (0, 2, 1))
    Z = p2[..., 0]
    u = p2[..., 2] / Z
    v = p2[..., 1] / Z
    d2 = np.zeros_like(p2)
    d2[..., 2] = du / Z
    d2[..., 1] = dv / Z
    d2[..., 0] = dw - (du * u + dv * v) / Z
    dM = np.einsum("bni,bnj->bij", d2, xyz1)
    dpose = intr[None].transpose(0, 2, 1) @ dM
    dpc = (d2 @ M)[..., 0:3]
    return dict(dpc=dpc, dpose=dpose, dtrans=None, dfocal=None)


# --------------------------------------------------------------------------
# A5  trilinear scatter                         dpc/util/point_cloud.py:60-136
# --------------------------------------------------------------------------
def _cells(tr_pc, Dz, D):
    dt = tr_pc.dtype
    half = dt.type(0.5)
    with np.errstate(invalid="ignore"):
        valid = np.all((tr_pc >= -half) & (tr_pc <= half), axis=-1)
    size = np.array([Dz, D, D], dtype=dt)
    g = (tr_pc + half) * (size - 1)
    g = np.where(valid[..., None], g, 0)
    fl = np.floor(g)
    return valid, fl.astype(np.int64), g - fl, size


def voxelize_fwd(tr_pc, Dz, D):
    """tr_pc [B,N,3] -> raw grid [B,Dz,D,D]; duplicates add; points outside the
    closed cube (or NaN) are dropped; an upper corner equal to the size (only
    with weight 0) is skipped."""
    B, N, _ = tr_pc.shape
    valid, i, r, _ = _cells(tr_pc, Dz, D)
    G = np.zeros((B, Dz, D, D), tr_pc.dtype)
    bidx = np.broadcast_to(np.arange(B)[:, None], (B, N))
    rr = [1.0 - r, r]
    for k in range(2):
        for j in range(2):
            for l in range(2):
                w = rr[k][..., 0] * rr[j][..., 1] * rr[l][..., 2]
                iz, iy, ix = i[..., 0] + k, i[..., 1] + j, i[..., 2] + l
                ok = valid & (iz < Dz) & (iy < D) & (ix < D)
                np.add.at(G, (bidx[ok], iz[ok], iy[ok], ix[ok]), w[ok])
    return G


def voxelize_bwd(tr_pc, dG, Dz, D):
    """d(raw grid) [B,Dz,D,D] -> d tr_pc [B,N,3]; zero for dropped points."""
    B, N, _ = tr_pc.shape
    valid, i, r, size = _cells(tr_pc, Dz, D)
    bidx = np.broadcast_to(np.arange(B)[:, None], (B, N))
    rr = [1.0 - r, r]
    sg = [-1.0, 1.0]
    dr = np.zeros_like(tr_pc)
    for k in range(2):
        for j in range(2):
            for l in range(2):
                iz, iy, ix = i[..., 0] + k, i[..., 1] + j, i[..., 2] + l
                ok = valid & (iz < Dz) & (iy < D) & (ix < D)
                g = np.zeros((B, N), tr_pc.dtype)
                g[ok] = dG[bidx[ok], iz[ok], iy[ok], ix[ok]]
                dr[..., 0] += g * sg[k] * rr[j][..., 1] * rr[l][..., 2]
                dr[..., 1] += g * rr[k][..., 0] * sg[j] * rr[l][..., 2]
                dr[..., 2] += g * rr[k][..., 0] * rr[j][..., 1] * sg[l]
    return dr * (size - 1) * valid[..., None]


# --------------------------------------------------------------------------
# A6  separable blur                          dpc/util/point_cloud.py:139-145
# --------------------------------------------------------------------------
def blur1d(G, taps, axis):
    """Zero-padded (SAME) 1-D correlation along ``axis``."""
    K = len(taps)
    h = K // 2
    n = G.shape[axis]
    out = np.zeros_like(G)
    for m in range(K):
        s = m - h                      # out[i] += taps[m] * G[i+s]
        lo, hi = max(0, -s), min(n, n - s)
        if lo >= hi:
            continue
        dst = [slice(None)] * G.ndim
        src = [slice(None)] * G.ndim
        dst[axis] = slice(lo, hi)
        src[axis] = slice(lo + s, hi + s)
        out[tuple(dst)] += taps[m] * G[tuple(src)]
    return out


def blur3d(G, taps, order=("x", "y", "z")):
    """G [B,Dz,D,D]; taps=[tx,ty,tz].  Reference order is x, y, z
    (gauss_kernel.py:27-32); the adjoint applies z, y, x."""
    ax = {"x": 3, "y": 2, "z": 1}
    tp = {"x": taps[0], "y": taps[1], "z": taps[2]}
    for a in order:
        G = blur1d(G, tp[a], ax[a])
    return G


# --------------------------------------------------------------------------
# A8/A9  DRC                                            dpc/util/drc.py:47-153
# --------------------------------------------------------------------------
def drc_fwd(G3, eps=1e-5):
    """G3 [B,Dz,D,D] -> probs p [Dz+1,B,D,D], proj [B,D,D] (unflipped)."""
    dt = G3.dtype
    e = dt.type(eps)
    c = np.clip(np.moveaxis(G3, 1, 0), e, dt.type(1) - e)
    y = np.log(c)
    x = np.log(dt.type(1) - c)
    r = np.cumsum(x, axis=0, dtype=dt)
    p1 = np.concatenate([np.full_like(r[:1], e), r], axis=0)     # "unity" is eps (drc.py:58-59)
    p2 = np.concatenate([y, np.full_like(y[:1], e)], axis=0)
    p = np.exp(p1 + p2)
    return p, p[:-1].sum(0)


def depth_grid(Dz, camera_distance=2.0, max_depth=10.0, dtype=np.float64):
    dtype = np.dtype(dtype)
    psi = np.arange(Dz, dtype=dtype) / dtype.type(Dz) - dtype.type(0.5) + dtype.type(camera_distance)
    return np.concatenate([psi, np.array([max_depth], dtype)])


def drc_bwd(G3, gamma, eps=1e-5):
    """gamma [Dz+1,B,D,D] = dL/dp  ->  dL/dG3 [B,Dz,D,D].

    a_i = gamma_i p_i;  dL/dc_j = a_j/c_j - (sum_{i>j} a_i)/(1-c_j), masked by
    eps <= G3_j <= 1-eps (closed: TF1 clip_by_value = max(min(x,hi),lo))."""
    dt = G3.dtype
    e = dt.type(eps)
    one = dt.type(1)
    v = np.moveaxis(G3, 1, 0)
    c = np.clip(v, e, one - e)
    p, _ = drc_fwd(G3, eps)
    a = gamma * p
    suffix = np.cumsum(a[::-1], axis=0)[::-1]          # suffix[i] = sum_{k>=i} a_k
    dc = a[:-1] / c - suffix[1:] / (one - c)
    dc = dc * ((v >= e) & (v <= one - e))
    return np.moveaxis(dc, 0, 1)


# --------------------------------------------------------------------------
# A1  orchestration                           dpc/util/point_cloud.py:229-290
# --------------------------------------------------------------------------
def project_forward(pc, pose, trans=None, scale=None, focal=None, taps=None,
                    Dz=64, D=64, camera_distance=2.0, focal_length=1.875,
                    eps=1e-5, max_depth=10.0, pose_quaternion=True,
                    max_projection=False):
    """Returns a dict with every intermediate (G0..G3), the reference's outputs
    in the reference's layouts, and what the backward needs."""
    dt = pc.dtype
    tr = transform_fwd(pc, pose, trans, focal, camera_distance, focal_length, pose_quaternion)
    G0 = voxelize_fwd(tr, Dz, D)
    G1 = np.clip(G0, 0, 1)
    G2 = blur3d(G1, taps) if taps is not None else G1
    if scale is not None:
        sG = scale.reshape(-1, 1, 1, 1) * G2
        G3 = np.clip(sG, 0, 1)
    else:
        G3 = G2
    out = dict(tr_pc=tr, G0=G0, G1=G1, G2=G2, G3=G3, voxels=G3[..., None])
    if max_projection:
        proj = G3.max(1)
        out.update(proj=proj[:, ::-1, :, None], drc_probs=None, proj_depth=None, p=None)
        return out
    p, proj = drc_fwd(G3, eps)
    psi = depth_grid(Dz, camera_distance, max_depth, dt).reshape(-1, 1, 1, 1)
    depth = (p * psi).sum(0)
    out.update(p=p, proj=proj[:, ::-1, :, None], drc_probs=p[:, :, ::-1, :, None],
               proj_depth=depth[:, ::-1, :, None])
    return out


def project_backward(pc, pose, trans, scale, focal, taps, fw, dproj=None,
                     dproj_depth=None, ddrc_probs=None, dvoxels=None, dtr_pc=None,
                     camera_distance=2.0, focal_length=1.875, eps=1e-5,
                     max_depth=10.0, pose_quaternion=True, max_projection=False):
    """Explicit backward.  Upstream grads are in the reference's OUTPUT layouts
    ([B,D,D,1] flipped images, [Dz+1,B,D,D,1] flipped probs, [B,Dz,D,D,1])."""
    dt = pc.dtype
    G0, G2, G3 = fw["G0"], fw["G2"], fw["G3"]
    B, Dz, D, _ = G0.shape
    dG3 = np.zeros_like(G3)
    if dvoxels is not None:
        dG3 += dvoxels[..., 0]
    if max_projection:
        if dproj is not None:
            g = dproj[:, ::-1, :, 0]
            # TF _MinOrMaxGrad: ties share the gradient equally (empty rays: 1/Dz each)
            ind = (G3 == G3.max(1, keepdims=True)).astype(dt)
            dG3 += ind / ind.sum(1, keepdims=True) * g[:, None]
    else:
        gamma = np.zeros((Dz + 1, B, D, D), dt)
        if dproj is not None:
            gamma[:-1] += dproj[:, ::-1, :, 0][None]
        if dproj_depth is not None:
            psi = depth_grid(Dz, camera_distance, max_depth, dt).reshape(-1, 1, 1, 1)
            gamma += psi * dproj_depth[:, ::-1, :, 0][None]
        if ddrc_probs is not None:
            gamma += ddrc_probs[:, :, ::-1, :, 0]
        dG3 += drc_bwd(G3, gamma, eps)
    dscale = None
    if scale is not None:
        s = scale.reshape(-1, 1, 1, 1)
        sG = s * G2
        m = (sG >= 0) & (sG <= 1)
        dscale = (G2 * dG3 * m).sum((1, 2, 3)).reshape(scale.shape)
        dG2 = s * dG3 * m
    else:
        dG2 = dG3
    dG1 = blur3d(dG2, taps, order=("z", "y", "x")) if taps is not None else dG2
    dG0 = dG1 * ((G0 >= 0) & (G0 <= 1))
    d_tr = voxelize_bwd(fw["tr_pc"], dG0, Dz, D)
    if dtr_pc is not None:
        d_tr = d_tr + dtr_pc
    g = transform_bwd(pc, pose, trans, focal, d_tr, camera_distance, focal_length, pose_quaternion)
    g["dscale"] = dscale
    g["dG0"], g["dG2"], g["d_tr"] = dG0, dG2, d_tr
    return g


# ---------------------------------------------------------------------------
# Exact Gaussian voxeliser (slow path, cfg.pc_fast:false)
# ---------------------------------------------------------------------------
def gauss_voxelize_fwd(pc, G, sigma, normalise="analytical", dtype=np.float64):
    """pointcloud2voxels (dpc/util/point_cloud.py:17-57) in the reference's meshgrid layout:
    out[b,i,j,k] sums exp(-((x-r_j)^2 + (y-r_i)^2 + (z-r_k)^2) / 2 sigma^2) over the points,
    r = linspace(-1,1,G).  normalise: None | "sum" (:43-45) | "analytical" (:46-51).
    Returns (clipped [B,G,G,G], raw sums)."""
    pc = np.asarray(pc, dtype=dtype)
    r = np.linspace(-1.0, 1.0, G).astype(dtype)
    k = 1.0 / (2.0 * sigma * sigma)
    ex = np.exp(-(pc[:, :, 0, None] - r) ** 2 * k)          # [B,N,G] along j
    ey = np.exp(-(pc[:, :, 1, None] - r) ** 2 * k)          # along i
    ez = np.exp(-(pc[:, :, 2, None] - r) ** 2 * k)          # along k
    if normalise == "sum":
        ex, ey, ez = ex / ex.sum(-1, keepdims=True), ey / ey.sum(-1, keepdims=True), ez / ez.sum(-1, keepdims=True)
    raw = np.einsum("bni,bnj,bnk->bijk", ey, ex, ez)
    if normalise == "analytical":
        raw = raw / (1.78984352254 * (sigma * G) ** 3)
    return np.clip(raw, 0.0, 1.0), raw


def gauss_voxelize_bwd(pc, G, sigma, raw, dvox, normalise="analytical", dtype=np.float64):
    """Gradient of gauss_voxelize_fwd wrt the points (what autodiff through :17-57 gives):
    closed-interval clip mask, product rule over the three separable factors, quotient rule
    under per-point normalisation."""
    pc = np.asarray(pc, dtype=dtype)
    r = np.linspace(-1.0, 1.0, G).astype(dtype)
    k = 1.0 / (2.0 * sigma * sigma)
    g = np.where((raw >= 0.0) & (raw <= 1.0), dvox, 0.0)
    if normalise == "analytical":
        g = g / (1.78984352254 * (sigma * G) ** 3)
    f, df = [], []
    for a in range(3):
        d = r - pc[:, :, a, None]                            # r_i - c
        e = np.exp(-d * d * k)
        if normalise == "sum":
            s = e.sum(-1, keepdims=True)
            q = (e * d * 2.0 * k).sum(-1, keepdims=True) / s
            e = e / s
        else:
            q = 0.0
        f.append(e)
        df.append(e * (d * 2.0 * k - q))
    fx, fy, fz = f
    dx, dy, dz = df
    out = np.zeros_like(pc)
    out[:, :, 0] = np.einsum("bijk,bni,bnj,bnk->bn", g, fy, dx, fz)
    out[:, :, 1] = np.einsum("bijk,bni,bnj,bnk->bn", g, dy, fx, fz)
    out[:, :, 2] = np.einsum("bijk,bni,bnj,bnk->bn", g, fy, fx, dz)
    return out
