"""Autograd binding of the fused render op (csrc/render.hip): everything between the raw
GaussianModel.params + pose and the two rendered images of gaussian_renderer.render()."""
import ctypes as C

import torch

from . import _lib, rasterizer
from .model import PARAM_NAMES


def _f32c(t):
    return t.detach().contiguous().to(torch.float32)


def _args_struct(xyz, f_dc, f_rest, opacity, scaling, rotation, w2c, cam_center, active_deg, max_deg):
    a = _lib.FsgsRenderArgs()
    a.xyz, a.features_dc, a.features_rest = xyz.data_ptr(), f_dc.data_ptr(), f_rest.data_ptr()
    a.opacity, a.scaling, a.rotation = opacity.data_ptr(), scaling.data_ptr(), rotation.data_ptr()
    a.w2c, a.cam_center = w2c.data_ptr(), cam_center.data_ptr()
    a.active_sh_degree, a.max_sh_degree = int(active_deg), int(max_deg)
    return a


_sizes = {}  # (P, W, H, max_pairs) -> (state bytes, scratch bytes) of the fused render: queried once per shape


def _render_sizes(lib, P, W, H, cap):
    hit = _sizes.get((P, W, H, cap))
    if hit is None:
        sb, xb = C.c_size_t(0), C.c_size_t(0)
        _lib.check(lib.fsgs_render_sizes(P, W, H, cap, C.byref(sb), C.byref(xb)), "fsgs_render_sizes")
        if len(_sizes) >= 256:
            _sizes.clear()
        hit = _sizes[(P, W, H, cap)] = (sb.value, xb.value)
    return hit


class _FusedRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, f_dc, f_rest, opacity, scaling, rotation, w2c, means2D, cam_center, settings, active_deg,
                max_deg, gs_grad, cam_grad, param_grads):
        lib = _lib.load()
        if not xyz.is_cuda:
            raise RuntimeError("fsgs fused render needs CUDA/HIP tensors; there is no CPU fallback")
        dev = xyz.device
        t = [_f32c(v) for v in (xyz, f_dc, f_rest, opacity, scaling, rotation, w2c, cam_center)]
        P = int(t[0].shape[0])
        cfg = rasterizer.cached_cfg(settings, 6)  # read-only here; shared with every other call on this settings object
        H, W = cfg.image_height, cfg.image_width
        image = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth_sil = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        args = _args_struct(*t, active_deg, max_deg)
        cap = rasterizer._capacity_for(P, W, H)
        nr = C.c_int64(0)
        with torch.cuda.device(dev):
            stream = _lib.current_stream()
            for _attempt in range(3):
                sbytes, xbytes = _render_sizes(lib, P, W, H, cap)
                state = torch.empty((sbytes,), dtype=torch.uint8, device=dev)
                scratch = torch.empty((xbytes,), dtype=torch.uint8, device=dev)
                rc = lib.fsgs_render_forward(C.byref(cfg), P, C.byref(args), _lib.ptr(image), _lib.ptr(depth_sil),
                                             _lib.ptr(radii), _lib.ptr(state), sbytes, _lib.ptr(scratch), xbytes,
                                             cap, C.byref(nr), stream)
                if rc == _lib.FSGS_ERR_CAPACITY and nr.value > cap:
                    cap = int(nr.value * 1.25) + 1024
                    rasterizer._capacity[(P, W, H)] = cap
                    continue
                _lib.check(rc, "fsgs_render_forward")
                break
            else:
                raise _lib.FsgsError(_lib.FSGS_ERR_CAPACITY, "fsgs_render_forward")
        rasterizer.last_num_rendered = int(nr.value)
        ctx.save_for_backward(*t, radii)
        ctx.misc = (cfg, state, sbytes, cap, int(nr.value), int(active_deg), int(max_deg), bool(gs_grad),
                    bool(cam_grad), bool(param_grads))
        ctx.mark_non_differentiable(radii)
        return image, depth_sil, radii

    @staticmethod
    def backward(ctx, g_image, g_depth_sil, g_radii):
        lib = _lib.load()
        *t, radii = ctx.saved_tensors
        cfg, state, sbytes, cap, nr, active_deg, max_deg, gs_grad, cam_grad, param_grads = ctx.misc
        xyz = t[0]
        dev = xyz.device
        P = int(xyz.shape[0])
        gi = None if g_image is None else _f32c(g_image)
        gd = None if g_depth_sil is None else _f32c(g_depth_sil)
        z = lambda like: torch.empty_like(like)
        grads = _lib.FsgsRenderGrads()
        need_xyz = gs_grad or param_grads
        d_xyz = z(t[0]) if need_xyz else None
        d = [z(v) if param_grads else None for v in t[1:6]]
        d_m2 = torch.empty((P, 3), dtype=torch.float32, device=dev)
        d_w2c = torch.empty((4, 4), dtype=torch.float32, device=dev) if cam_grad else None
        grads.xyz = None if d_xyz is None else d_xyz.data_ptr()
        (grads.features_dc, grads.features_rest, grads.opacity, grads.scaling,
         grads.rotation) = [None if v is None else v.data_ptr() for v in d]
        grads.means2D = d_m2.data_ptr()
        grads.w2c = None if d_w2c is None else d_w2c.data_ptr()
        if P > 0:
            args = _args_struct(*t, active_deg, max_deg)
            scratch = torch.empty((rasterizer.backward_scratch_bytes(cfg, P, cap, P * 64 + 512),), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = lib.fsgs_render_backward(C.byref(cfg), P, C.byref(args), _lib.ptr(radii), _lib.ptr(state), sbytes,
                                              cap, nr, _lib.ptr(gi), _lib.ptr(gd), int(gs_grad), int(cam_grad),
                                              int(param_grads), C.byref(grads), _lib.ptr(scratch), scratch.numel(),
                                              _lib.current_stream())
            _lib.check(rc, "fsgs_render_backward")
        elif d_w2c is not None:
            d_w2c.zero_()
        # (ctx.misc stays: the forward state is read-only, so a retained graph can be backpropagated again)
        return (d_xyz, d[0], d[1], d[2], d[3], d[4], d_w2c, d_m2, None, None, None, None, None, None, None)


def fused_render(pc, w2c, means2D, cam_center, gs_grad=True, cam_grad=True, param_grads=None):
    """-> (image [3,H,W], depth_sil [3,H,W], radii [P]).  gs_grad / cam_grad: transform_to_frame's
    switches (scene/pose_optimizer.py:976-982).  param_grads defaults to gs_grad: with gs_grad=False
    (tracking) the reference still back-propagates into the Gaussian parameters but discards the
    result (train.py:220; SURVEY.md a1 note v), so the pose-only backward is observationally equivalent."""
    if param_grads is None:
        param_grads = gs_grad
    p = pc.params
    ins = [p[k] for k in PARAM_NAMES]
    # order of the op: xyz, f_dc, f_rest, opacity, scaling, rotation
    xyz, f_dc, f_rest, opacity, scaling, rotation = ins
    if not (gs_grad or param_grads):
        xyz = xyz.detach()
    if not param_grads:
        f_dc, f_rest, opacity, scaling, rotation = (v.detach() for v in (f_dc, f_rest, opacity, scaling, rotation))
    w = w2c if cam_grad else w2c.detach()
    return _FusedRender.apply(xyz, f_dc, f_rest, opacity, scaling, rotation, w, means2D, cam_center, pc.cam,
                              pc.active_sh_degree, pc.max_sh_degree, gs_grad, cam_grad, param_grads)
