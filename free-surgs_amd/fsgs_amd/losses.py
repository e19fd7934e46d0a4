"""Photometric / depth losses of the hot path.

`*_torch` functions are the plain-PyTorch statement of the reference's maths
(utils/loss_utils.py:41-127), used as glue until / as the reference for the fused HIP kernels
(csrc/loss.hip) which `rgb_loss_func`, `pearson_depth_loss`, `local_pearson_loss` dispatch to
for CUDA tensors."""
import math

import torch
import torch.nn.functional as F

_WINDOW = 11
_SIGMA = 1.5


def gaussian_window(size=_WINDOW, sigma=_SIGMA):
    g = torch.tensor([math.exp(-((i - size // 2) ** 2) / (2.0 * sigma * sigma)) for i in range(size)])
    return g / g.sum()


def l1_loss(a, b):
    return (a - b).abs().mean()


def ssim_torch(img1, img2):
    """Mean SSIM, 11x11 Gaussian (sigma 1.5), zero padding, C1=1e-4, C2=9e-4 (utils/loss_utils.py:56-96)."""
    x = img1 if img1.dim() == 4 else img1.unsqueeze(0)
    y = img2 if img2.dim() == 4 else img2.unsqueeze(0)
    ch = x.shape[-3]
    g = gaussian_window().to(x)
    w = (g[:, None] * g[None, :]).expand(ch, 1, _WINDOW, _WINDOW).contiguous()
    conv = lambda t: F.conv2d(t, w, padding=_WINDOW // 2, groups=ch)
    mu1, mu2 = conv(x), conv(y)
    s11 = conv(x * x) - mu1 * mu1
    s22 = conv(y * y) - mu2 * mu2
    s12 = conv(x * y) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    return m.mean()


def rgb_loss_torch(img, gt, lambda_dssim=0.2, mask=None):
    """0.8 L1 + 0.2 (1 - SSIM), optional multiplicative mask on both images (utils/loss_utils.py:47-54)."""
    if mask is not None:
        img = img * mask
        gt = gt * mask
    return (1.0 - lambda_dssim) * l1_loss(img, gt) + lambda_dssim * (1.0 - ssim_torch(img, gt))


def pearson_torch(src, tgt):
    """1 - mean(z_src z_tgt), z = (v - mean)/(unbiased std + 1e-6) (utils/loss_utils.py:98-109)."""
    s = src - src.mean()
    t = tgt - tgt.mean()
    s = s / (s.std() + 1e-6)
    t = t / (t.std() + 1e-6)
    return 1 - (s * t).mean()


def draw_patch_corners(H, W, box, p_corr, device):
    """The reference's RNG consumption: two randint draws on the device (utils/loss_utils.py:114-121)."""
    nh, nw = H // box, W // box
    n = int(p_corr * nh * nw)
    x0 = torch.randint(0, H - box, size=(n,), device=device)
    y0 = torch.randint(0, W - box, size=(n,), device=device)
    return x0, y0


def local_pearson_torch(src, tgt, box, p_corr, corners=None):
    """Mean Pearson loss over random box x box patches (utils/loss_utils.py:112-127)."""
    if corners is None:
        corners = draw_patch_corners(src.shape[0], src.shape[1], box, p_corr, src.device)
    x0, y0 = corners
    n = len(x0)
    total = torch.zeros((), device=src.device)
    for i in range(n):
        a, b = int(x0[i]), int(y0[i])
        total = total + pearson_torch(src[a:a + box, b:b + box].reshape(-1), tgt[a:a + box, b:b + box].reshape(-1))
    return total / n


# product entry points (same names as utils/loss_utils.py) ---------------------------------------
def _hip_ops():
    from . import loss_ops  # raises if libfsgs_hip.so is missing: no silent fallback

    return loss_ops


def rgb_loss_func(img, gt, lambda_dssim=0.2, mask=None):
    return _hip_ops().rgb_loss(img, gt, lambda_dssim, mask)


def pearson_depth_loss(src, tgt):
    return _hip_ops().pearson(src, tgt)


def local_pearson_loss(src, tgt, box, p_corr, corners=None):
    return _hip_ops().local_pearson(src, tgt, box, p_corr, corners)
