"""Mirror of dpc/util/gauss_kernel.py:5-11,27-54 on torch tensors.

The taps are a handful of floats recomputed whenever sigma changes (it is
annealed with the global step, dpc/models/model_pc.py:35-40,146-153); they are
produced with torch ops on whatever device is asked for and handed to the HIP
blur kernels as device pointers."""
import math

import torch


def _device(device):
    if device is not None:
        return torch.device(device)
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


# Taps below this fraction of the centre tap are not applied (see effective_half_width).  1e-8 is under the fp32
# rounding of the sums the taps enter (2^-24 = 6e-8 relative): the trimmed filter's output equals the full one's to
# fp32 accuracy, while the reference's sigma schedule (3.0 -> 0.2 with K = 21 fixed, dpc/models/model_pc.py:33-38,
# default_config.yaml:49-50) spends half of a run at sigmas whose outer taps are that small.
TAP_DROP_REL = 1e-8


def effective_half_width(sig, half):
    """Largest offset m <= half whose tap exp(-m^2 / 2 sig^2) is still >= TAP_DROP_REL of the centre tap: the filter
    of 2 * half + 1 taps acts as one of 2 * m + 1 (the normalisation still runs over all the taps, as in the reference,
    so the kept taps are bit-identical to the full filter's)."""
    sig = float(sig)
    if not (sig > 0.0) or not math.isfinite(sig):
        return int(half)
    return int(min(half, math.floor(sig * math.sqrt(-2.0 * math.log(TAP_DROP_REL)))))


def _tag_support(t, h):
    """remember, on the host, how many of the filter's taps matter (read by util.point_cloud._flat_taps; tensors
    that do not carry the tag -- anything not made here from a host-side sigma -- are applied in full)"""
    t.dpc_support = int(h)
    return t


def gauss_kernel_1d(l, sig, device=None):
    """Gaussian taps at integer offsets range(-l//2+1., l//2+1.), sum 1, fp32."""
    l = int(l)
    if l % 2 != 1:
        raise ValueError("even kernel sizes are not supported: TF's SAME padding is "
                         "asymmetric for them and the reference only uses odd sizes")
    dev = sig.device if (isinstance(sig, torch.Tensor) and device is None) else _device(device)
    xx = torch.arange(-l // 2 + 1.0, l // 2 + 1.0, dtype=torch.float32, device=dev)
    host_sigma = None if isinstance(sig, torch.Tensor) else float(sig)
    sig = torch.as_tensor(sig, dtype=torch.float32, device=dev)
    kernel = torch.exp(-xx ** 2 / (2.0 * sig ** 2))
    kernel = kernel / kernel.sum()
    if host_sigma is not None:
        _tag_support(kernel, effective_half_width(host_sigma, l // 2))
    return kernel


def gauss_smoothen_image(cfg, img, sigma_rel, kernel=None):
    """dpc/util/gauss_kernel.py:14-24: per-channel separable blur of [B,H,W,C] images, SAME zero
    padding (GT masks / images only; a few launches of stock depthwise conv on small tensors).
    `kernel`: the 1-D taps as a device tensor instead of a sigma -- a step replayed as a HIP graph reads them from a
    buffer at a fixed address that the sigma schedule overwrites in place (ModelPointCloud.enable_graph_replay)."""
    fsz = int(cfg.pc_gauss_kernel_size)
    if kernel is None:
        kernel = gauss_kernel_1d(fsz, sigma_rel, img.device)
    kernel = kernel.reshape(-1).to(img.dtype)
    c = img.shape[-1]
    x = img.permute(0, 3, 1, 2)
    x = torch.nn.functional.conv2d(x, kernel.reshape(1, 1, 1, fsz).repeat(c, 1, 1, 1), padding=(0, fsz // 2), groups=c)
    x = torch.nn.functional.conv2d(x, kernel.reshape(1, 1, fsz, 1).repeat(c, 1, 1, 1), padding=(fsz // 2, 0), groups=c)
    return x.permute(0, 2, 3, 1)


def _filters(kernel_1d, shapes):
    out = [kernel_1d.reshape(*s) for s in shapes]
    h = getattr(kernel_1d, "dpc_support", None)
    if h is not None:
        for f in out:
            _tag_support(f, h)
    return out


def separable_kernels(kernel):
    size = kernel.shape[0]
    return _filters(kernel, [(1, 1, size, 1, 1), (1, size, 1, 1, 1), (size, 1, 1, 1, 1)])


def smoothing_kernel(cfg, sigma, device=None):
    """-> [k_x, k_y, k_z] with the reference's filter shapes [1,1,K,1,1],
    [1,K,1,1,1], [Kz,1,1,1,1] (=> blur x first, z last)."""
    fsz = cfg.pc_gauss_kernel_size
    kernel_1d = gauss_kernel_1d(fsz, sigma, device)
    if cfg.vox_size_z != -1:
        ratio = cfg.vox_size_z / cfg.vox_size
        fsz_z = int(math.floor(fsz * ratio))
        if fsz_z % 2 == 0:
            fsz_z += 1
        kernel_1d_z = gauss_kernel_1d(fsz_z, sigma * ratio, device)
        return _filters(kernel_1d, [(1, 1, fsz, 1, 1), (1, fsz, 1, 1, 1)]) + _filters(kernel_1d_z, [(fsz_z, 1, 1, 1, 1)])
    if not cfg.pc_separable_gauss_filter:
        raise NotImplementedError("dense 3-D Gaussian kernel (pc_separable_gauss_filter=false)")
    return separable_kernels(kernel_1d)
