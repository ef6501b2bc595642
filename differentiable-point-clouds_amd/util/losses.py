"""Mirror of dpc/util/losses.py:23-136 (the loss terms that consume the projector's
outputs) on torch tensors.  Everything here is thin device-side glue over tensors the HIP
kernels produced -- [B,D,D,*] images, or the lazily materialised drc_probs [Dz+1,B,D,D,1];
the silhouette loss itself (add_proj_loss) lives in model_pc.py / ops.SilhouetteLoss.
`regularization_loss` belongs to the optimiser side (weight decay over the CNN variables)
and is not mirrored."""
import torch

from .gauss_kernel import gauss_smoothen_image


def resize_images_bilinear_tf1(images, size):
    """tf.image.resize_images(..., BILINEAR), TF1 legacy sampling (align_corners=False, no
    half-pixel centres): src = dst * in/out.  images [B,H,W,C] -> [B,size0,size1,C]."""
    n, ih, iw, c = images.shape
    oh, ow = int(size[0]), int(size[1])

    def axis(o, i):
        src = torch.arange(o, dtype=torch.float64, device=images.device) * (i / o)
        lo = torch.floor(src).to(torch.int64)
        hi = torch.clamp(lo + 1, max=i - 1)
        return lo, hi, (src - lo.to(torch.float64)).to(images.dtype)
    ylo, yhi, yl = axis(oh, ih)
    xlo, xhi, xl = axis(ow, iw)
    xl = xl.view(1, 1, -1, 1)
    yl = yl.view(1, -1, 1, 1)
    rows_lo, rows_hi = images[:, ylo], images[:, yhi]
    top = rows_lo[:, :, xlo] * (1 - xl) + rows_lo[:, :, xhi] * xl
    bot = rows_hi[:, :, xlo] * (1 - xl) + rows_hi[:, :, xhi] * xl
    return top * (1 - yl) + bot * yl


_BICUBIC_TABLE = {}


def _bicubic_table(device):
    """resize_bicubic_op.cc's coefficient table (TF r1.x): 1024 steps of the cubic convolution kernel with A = -0.75,
    entry 2 i = k(i / 1024), entry 2 i + 1 = k(1 + i / 1024), evaluated in double on a float x, stored as float."""
    key = str(device)
    if key not in _BICUBIC_TABLE:
        a = -0.75
        x = (torch.arange(1025, dtype=torch.float64) / 1024.0).to(torch.float32).to(torch.float64)
        near = ((a + 2) * x - (a + 3)) * x * x + 1
        x1 = (x.to(torch.float32) + 1.0).to(torch.float64)          # `x += 1.0` on the float
        far = ((a * x1 - 5 * a) * x1 + 8 * a) * x1 - 4 * a
        _BICUBIC_TABLE[key] = (near.to(torch.float32).to(device), far.to(torch.float32).to(device))
    return _BICUBIC_TABLE[key]


def resize_images_bicubic_tf1(images, size):
    """tf.image.resize_images(..., ResizeMethod.BICUBIC) as TF1 computes it (model_pc.py:392-397 with
    cfg.bicubic_gt_downsampling): the ResizeBicubic op with align_corners=False and the legacy scaler -- source position
    = dst * (in / out) in float, four taps at floor - 1 .. floor + 2 clamped to the image, weights looked up in a
    1024-step table of the A = -0.75 cubic kernel at lrint(frac * 1024), x pass then y pass in float32.
    Restated after the published op (tensorflow/core/kernels/resize_bicubic_op.cc) and pinned to vectors worked by hand
    out of that definition (tests/test_tf_shim.py::test_resize_images_bicubic_hand_worked_vectors: integer scales pick
    pixels, the (-3, 19, 19, -3) / 32 half-way taps with clamped -- not linear-exact -- borders, an 8 -> 3 row in exact
    rationals); no TensorFlow binary was available, so what stays assumed is the float storage of the table and the
    x-before-y float32 accumulation order (last-place effects).  The reference switches the branch off by default
    (default_config.yaml:96).  images [B,H,W,C] -> [B,size0,size1,C] float32."""
    n, ih, iw, c = images.shape
    oh, ow = int(size[0]), int(size[1])
    near, far = _bicubic_table(images.device)
    img = images.to(torch.float32)

    def axis(o, i):
        scale = torch.tensor(float(i), dtype=torch.float32) / torch.tensor(float(o), dtype=torch.float32)
        loc = torch.arange(o, dtype=torch.float32) * scale
        lo = torch.floor(loc)
        off = torch.round((loc - lo) * 1024.0).to(torch.int64)       # lrintf: half to even, as torch.round
        lo = lo.to(torch.int64)
        idx = torch.stack([torch.clamp(lo + d, 0, i - 1) for d in (-1, 0, 1, 2)]).to(images.device)
        off = off.to(images.device)
        w = torch.stack([far[off], near[off], near[1024 - off], far[1024 - off]])
        return idx, w
    xi, xw = axis(ow, iw)
    yi, yw = axis(oh, ih)
    xw = xw.view(4, 1, 1, ow, 1)
    cols = img[:, :, xi[0]] * xw[0] + img[:, :, xi[1]] * xw[1] + img[:, :, xi[2]] * xw[2] + img[:, :, xi[3]] * xw[3]
    yw = yw.view(4, 1, oh, 1, 1)
    return cols[:, yi[0]] * yw[0] + cols[:, yi[1]] * yw[1] + cols[:, yi[2]] * yw[2] + cols[:, yi[3]] * yw[3]


def resize_images_nearest_tf1(images, size):
    """TF1 legacy nearest neighbour: src = floor(dst * in/out)."""
    n, ih, iw, c = images.shape
    oh, ow = int(size[0]), int(size[1])
    ys = torch.clamp(torch.floor(torch.arange(oh, dtype=torch.float64, device=images.device) * (ih / oh)), max=ih - 1)
    xs = torch.clamp(torch.floor(torch.arange(ow, dtype=torch.float64, device=images.device) * (iw / ow)), max=iw - 1)
    return images[:, ys.to(torch.int64)][:, :, xs.to(torch.int64)]


def _l2_loss(x):                       # tf.nn.l2_loss
    return (x * x).sum() / 2


def drc_loss(cfg, probs, gt_proj):
    """losses.py:23-30: probs [Dz+1,B,D,D,1] against the mask [B,D,D,1]: events inside the
    grid cost (1 - mask), the escape event costs mask."""
    gt_proj2 = gt_proj.unsqueeze(0)
    psi = torch.cat([(1 - gt_proj2).expand(int(cfg.vox_size), -1, -1, -1, -1), gt_proj2], dim=0)
    return (probs * psi).sum()


def drc_rgb_loss(cfg, probs, rgb, gt):
    """losses.py:33-48: probs [Dz+1,B,D,D,1], rgb voxels [B,Dz,D,D,3], gt image [B,D,D,3]."""
    vox_size = int(cfg.vox_size)
    white_bg = torch.ones(rgb.shape[0], 1, vox_size, vox_size, 3, dtype=rgb.dtype, device=rgb.device)
    rgb_pred = torch.cat([rgb, white_bg], dim=1)
    psi = ((gt.unsqueeze(1) - rgb_pred) ** 2).sum(dim=4, keepdim=True)
    return (probs.permute(1, 0, 2, 3, 4) * psi).sum()


def add_drc_loss(cfg, inputs, outputs, weight_scale, add_summary=True):
    """losses.py:51-68."""
    gt = inputs["masks"]
    pred = outputs["drc_probs"]
    num_samples = gt.shape[0]
    if gt.shape[1] != pred.shape[2]:
        gt = resize_images_bilinear_tf1(gt, [pred.shape[2], pred.shape[2]])
    return drc_loss(cfg, pred, gt) / float(num_samples) * weight_scale


def add_proj_rgb_loss(cfg, inputs, outputs, weight_scale, add_summary=True, sigma=None):
    """losses.py:71-92."""
    gt = inputs["images"]
    pred = outputs["projs_rgb"]
    num_samples = pred.shape[0]
    if gt.shape[1] != pred.shape[1]:
        gt = resize_images_bilinear_tf1(gt, [pred.shape[1], pred.shape[1]])
    if getattr(cfg, "pc_gauss_filter_gt_rgb", False):
        smoothed = gauss_smoothen_image(cfg, gt, sigma)
        if getattr(cfg, "pc_gauss_filter_gt_switch_off", False):
            gt = gt if float(sigma) < 1.0 else smoothed
        else:
            gt = smoothed
    return _l2_loss(gt - pred) / float(num_samples) * weight_scale


def add_drc_rgb_loss(cfg, inputs, outputs, weight_scale, add_summary=True):
    """losses.py:95-112."""
    gt = inputs["images"]
    pred = outputs["voxels_rgb"]
    num_samples = pred.shape[0]
    if gt.shape[1] != pred.shape[1]:
        gt = resize_images_bilinear_tf1(gt, [pred.shape[1], pred.shape[1]])
    return drc_rgb_loss(cfg, outputs["drc_probs"], pred, gt) / float(num_samples) * weight_scale


def add_proj_depth_loss(cfg, inputs, outputs, weight_scale, sigma_rel, add_summary=True):
    """losses.py:115-136.  (The reference's resize branch names tf.ResizeMethod, which does not
    exist, so it only ever ran with gt and prediction of equal size; the intended legacy
    nearest-neighbour resize is what runs here.)"""
    gt = inputs["depths"]
    pred = outputs["projs_depth"]
    num_samples = pred.shape[0]
    if cfg.max_depth != cfg.max_dataset_depth:
        far = gt == cfg.max_dataset_depth
        gt = torch.where(far, torch.full_like(gt, float(cfg.max_depth)), gt)
    if gt.shape[1] != pred.shape[1]:
        gt = resize_images_nearest_tf1(gt, [pred.shape[1], pred.shape[1]])
    if getattr(cfg, "pc_gauss_filter_gt", False):
        gt = gauss_smoothen_image(cfg, gt, sigma_rel)
    return _l2_loss(gt - pred) / float(num_samples) * weight_scale
