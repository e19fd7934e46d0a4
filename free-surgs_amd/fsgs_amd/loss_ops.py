"""Autograd bindings of the fused loss kernels (csrc/loss.hip): rgb_loss_func / pearson_depth_loss /
local_pearson_loss of utils/loss_utils.py:41-127 with no host synchronisation anywhere."""
import ctypes as C

import torch

from . import _lib
from .losses import draw_patch_corners


def _f32c(t):
    return t.detach().contiguous().to(torch.float32)


class _RgbLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, mask, lambda_dssim):
        lib = _lib.load()
        if not img.is_cuda:
            raise RuntimeError("fsgs rgb_loss needs CUDA/HIP tensors; there is no CPU fallback")
        if tuple(img.shape) != tuple(gt.shape) or img.dim() not in (3, 4):
            raise ValueError("rgb_loss: image %s and target %s must both be [C,H,W] (or [1,C,H,W])" % (
                tuple(img.shape), tuple(gt.shape)))
        if gt.device != img.device or (mask is not None and mask.device != img.device):
            raise ValueError("rgb_loss: image, target and mask must be on one device")
        x, y = _f32c(img), _f32c(gt)
        if x.dim() == 4:
            x, y = x.reshape(-1, *x.shape[-2:]), y.reshape(-1, *y.shape[-2:])
        Cc, H, W = x.shape
        m = None
        if mask is not None:
            m = _f32c(mask)
            if m.numel() != H * W:
                raise ValueError("mask must broadcast as [1,H,W] (train.py:175-178)")
            m = m.reshape(H, W)
        maps = torch.empty((3, Cc, H, W), dtype=torch.float32, device=x.device)
        sums = torch.empty((int(lib.fsgs_photometric_scratch_bytes(Cc, H, W)),), dtype=torch.uint8, device=x.device)
        out = torch.empty((3,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.fsgs_photometric_loss_forward(Cc, H, W, _lib.ptr(x), _lib.ptr(y), _lib.ptr(m), None,
                                                   float(lambda_dssim), _lib.ptr(maps), _lib.ptr(sums),
                                                   _lib.ptr(out), _lib.current_stream())
        _lib.check(rc, "fsgs_photometric_loss_forward")
        ctx.save_for_backward(x, y, maps)
        ctx.mask = m
        ctx.lambda_dssim = float(lambda_dssim)
        ctx.in_shape = img.shape
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, y, maps = ctx.saved_tensors
        Cc, H, W = x.shape
        up = _f32c(grad_out).reshape(1)
        dimg = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = lib.fsgs_photometric_loss_backward(Cc, H, W, _lib.ptr(x), _lib.ptr(y), _lib.ptr(ctx.mask), None,
                                                    _lib.ptr(maps), _lib.ptr(up), ctx.lambda_dssim, _lib.ptr(dimg),
                                                    _lib.current_stream())
        _lib.check(rc, "fsgs_photometric_loss_backward")
        return dimg.reshape(ctx.in_shape), None, None, None


def rgb_loss(img, gt, lambda_dssim=0.2, mask=None):
    return _RgbLoss.apply(img, gt, mask, lambda_dssim)


class _Pearson(torch.autograd.Function):
    """returns (global_loss, mean_patch_loss); differentiable w.r.t. ONE of src / tgt."""

    @staticmethod
    def forward(ctx, src, tgt, row0, col0, box):
        lib = _lib.load()
        if not src.is_cuda:
            raise RuntimeError("fsgs pearson loss needs CUDA/HIP tensors; there is no CPU fallback")
        s, t = _f32c(src), _f32c(tgt)
        if s.numel() != t.numel():
            raise ValueError("pearson: shape mismatch")
        if s.dim() != 2:
            s, t = s.reshape(1, -1), t.reshape(1, -1)  # utils/loss_utils.py works on any shape (global mean/std)
        H, W = s.shape
        n = 0 if row0 is None else int(row0.numel())
        r0 = None if n == 0 else row0.detach().contiguous().to(torch.int64)
        c0 = None if n == 0 else col0.detach().contiguous().to(torch.int64)
        stats = torch.empty((int(lib.fsgs_pearson_scratch_bytes(H, W, n, int(box))),), dtype=torch.uint8,
                            device=s.device)
        coef = torch.empty((8 * (n + 1),), dtype=torch.float32, device=s.device)
        out = torch.empty((2,), dtype=torch.float32, device=s.device)
        with torch.cuda.device(s.device):
            rc = lib.fsgs_pearson_forward(H, W, n, int(box), _lib.ptr(r0), _lib.ptr(c0), _lib.ptr(s), _lib.ptr(t),
                                          _lib.ptr(stats), _lib.ptr(coef), _lib.ptr(out), _lib.current_stream())
        _lib.check(rc, "fsgs_pearson_forward")
        ctx.save_for_backward(s, t, coef)
        ctx.r0, ctx.c0, ctx.n, ctx.box = r0, c0, n, int(box)
        ctx.shapes = (src.shape, tgt.shape)
        ctx.need = (src.requires_grad, tgt.requires_grad)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_global, g_local):
        lib = _lib.load()
        s, t, coef = ctx.saved_tensors
        H, W = s.shape
        n = ctx.n
        w = torch.empty((n + 1,), dtype=torch.float32, device=s.device)
        w[0] = g_global
        if n:
            w[1:] = g_local / n
        grads = [None, None]
        for which in (0, 1):  # 0: d/dsrc, 1: d/dtgt
            if not ctx.needs_input_grad[which]:
                continue
            g = torch.empty_like(s)
            with torch.cuda.device(s.device):
                rc = lib.fsgs_pearson_backward(H, W, n, ctx.box, _lib.ptr(ctx.r0), _lib.ptr(ctx.c0), _lib.ptr(s),
                                               _lib.ptr(t), _lib.ptr(coef), _lib.ptr(w), 1 if which == 0 else 0,
                                               _lib.ptr(g), _lib.current_stream())
            _lib.check(rc, "fsgs_pearson_backward")
            grads[which] = g.reshape(ctx.shapes[which])
        return grads[0], grads[1], None, None, None


def pearson(src, tgt):
    return _Pearson.apply(src, tgt, None, None, 0)[0]


def local_pearson(src, tgt, box, p_corr, corners=None):
    if corners is None:
        corners = draw_patch_corners(src.shape[0], src.shape[1], box, p_corr, src.device)
    return _Pearson.apply(src, tgt, corners[0], corners[1], box)[1]


def pearson_pair(src, tgt, box, p_corr, corners=None):
    """global and local losses from ONE pair of launches (what the fused trainer uses)."""
    if corners is None:
        corners = draw_patch_corners(src.shape[0], src.shape[1], box, p_corr, src.device)
    return _Pearson.apply(src, tgt, corners[0], corners[1], box)
