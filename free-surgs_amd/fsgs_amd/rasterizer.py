"""Host side of the rasteriser operator: the Python surface of the un-vendored
`diff_gaussian_rasterization` package, bound to the HIP C ABI.

Reference call sites: gaussian_renderer/__init__.py:68-69,131 (forward, 3 return values),
scene/pose_optimizer.py:619-632 (settings tuple), train.py:337 (`._replace`).
"""
from typing import NamedTuple

import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---- host copies of the tiny camera tensors, cached so a render does not sync on them ----------
# Keyed by the tensor OBJECT (held alive by the entry, so neither its id nor its address can be recycled while it is
# cached) and its version counter.  A key made of (data_ptr, version, shape) alone is wrong: the caching allocator
# hands the address of a freed camera matrix to the next one, which starts at version 0 again.
_host_cache = {}
_HOST_CACHE_ENTRIES = 1024  # 16 floats each


def _host_floats(t, n):
    hit = _host_cache.get(id(t))
    if hit is None or hit[0] is not t or hit[1] != t._version:
        vals = [float(x) for x in t.detach().reshape(-1).to("cpu", torch.float32).tolist()]
        _host_cache.pop(id(t), None)  # a stale version of the same tensor: re-inserted as the newest entry
        while len(_host_cache) >= _HOST_CACHE_ENTRIES:
            _host_cache.pop(next(iter(_host_cache)))  # oldest first (dicts keep insertion order): a sweep over more
            # cameras than entries re-reads only what fell out, never the whole set at once
        hit = _host_cache[id(t)] = (t, t._version, vals)
    if len(hit[2]) < n:
        raise ValueError("camera tensor has %d elements, expected >= %d" % (len(hit[2]), n))
    return hit[2]


# ---- flavour of the blend kernels (include/fsgs.h FSGS_FLAG_BLEND_*) ---------------------------------------------------
# "auto" (the default): the library picks by the size of the tile grid.  "one" / "quad" force one wave / four waves per
# tile for every configuration struct made from here on -- the parity tests run over both, bench.py --blend measures the
# crossover.  FSGS_BLEND_VARIANT in the environment sets the initial value.
_BLEND_FLAGS = {"auto": 0, "one": _lib.FSGS_FLAG_BLEND_ONE_WAVE, "quad": _lib.FSGS_FLAG_BLEND_QUAD_WAVES}
_blend_variant = os.environ.get("FSGS_BLEND_VARIANT", "auto")
if _blend_variant not in _BLEND_FLAGS:
    raise ValueError("FSGS_BLEND_VARIANT must be one of %s, not %r" % (sorted(_BLEND_FLAGS), _blend_variant))


def set_blend_variant(name):
    """force ("one" / "quad") or release ("auto") the flavour of the blend kernels; returns the previous setting.
    Cached configuration structs are dropped, so the next render of every settings object sees the new value (a
    FastStepper re-makes its own struct when the switch has changed)."""
    global _blend_variant
    if name not in _BLEND_FLAGS:
        raise ValueError("blend variant must be one of %s, not %r" % (sorted(_BLEND_FLAGS), name))
    prev, _blend_variant = _blend_variant, name
    _cfg_cache.clear()
    return prev


def blend_variant():
    return _blend_variant


# ---- bit-reproducible backward (include/fsgs.h FSGS_FLAG_DETERMINISTIC; tests) -----------------------------------------
_deterministic = os.environ.get("FSGS_DETERMINISTIC", "") not in ("", "0")


def set_deterministic(on):
    """every configuration struct made from here on asks for the deterministic backward (fixed-order sums instead of float
    atomics; needs backward_scratch_bytes() of scratch); returns the previous setting"""
    global _deterministic
    prev, _deterministic = _deterministic, bool(on)
    _cfg_cache.clear()
    return prev


def deterministic():
    return _deterministic


def backward_scratch_bytes(cfg, P, max_pairs, rows_bytes):
    """bytes of `scratch` a backward call with this configuration needs: the per-Gaussian accumulator rows alone, or --
    FSGS_FLAG_DETERMINISTIC -- rows + one 64-byte row per pair slot + the dL/dw2c partials"""
    if cfg.flags & _lib.FSGS_FLAG_DETERMINISTIC:
        return max(int(_lib.load().fsgs_deterministic_scratch_bytes(int(P), int(max_pairs))), rows_bytes)
    return rows_bytes


def make_cfg(settings, channels):
    cfg = _lib.FsgsRasterCfg()
    cfg.image_height = int(settings.image_height)
    cfg.image_width = int(settings.image_width)
    cfg.channels = int(channels)
    cfg.flags = _BLEND_FLAGS[_blend_variant] | (_lib.FSGS_FLAG_DETERMINISTIC if _deterministic else 0)
    cfg.tanfovx = float(settings.tanfovx)
    cfg.tanfovy = float(settings.tanfovy)
    cfg.scale_modifier = float(settings.scale_modifier)
    bg = _host_floats(settings.bg, 1)
    for i in range(_lib.MAX_CHANNELS):
        # channels beyond len(bg) (fused depth/silhouette planes) share the last background value
        cfg.bg[i] = bg[i] if i < len(bg) else bg[-1]
    vm = _host_floats(settings.viewmatrix, 16)
    pm = _host_floats(settings.projmatrix, 16)
    for i in range(16):
        cfg.viewmatrix[i] = vm[i]
        cfg.projmatrix[i] = pm[i]
    return cfg


# ---- the configuration struct of a settings object, cached (the drop-in makes two rasteriser calls per
# render() with the SAME settings tuple; rebuilding the 50-field ctypes struct for each was ~40 us of host time) ----------
# Keyed like _host_cache: the settings object itself is kept in the entry (an id() alone can be recycled) together with
# the version counters of its three tensors; callers get a copy of the cached struct.  The camera tensors must be
# updated through versioned in-place ops (or optim.mark_updated after a raw-pointer kernel): a write through `.data` does
# not bump the counter and leaves a stale camera here.
_cfg_cache = {}


def cached_cfg(settings, channels):
    key = (id(settings), int(channels))
    vers = (settings.viewmatrix._version, settings.projmatrix._version, settings.bg._version)
    hit = _cfg_cache.get(key)
    if hit is None or hit[0] is not settings or hit[1] != vers:
        if len(_cfg_cache) >= 256:
            _cfg_cache.pop(next(iter(_cfg_cache)))
        hit = _cfg_cache[key] = (settings, vers, make_cfg(settings, channels))
    # a COPY (one 176-byte memmove against rebuilding 50 ctypes fields): a caller that sets flags on what it gets can
    # never change what the next render with the same settings object sees
    out = _lib.FsgsRasterCfg()
    C.memmove(C.byref(out), C.byref(hit[2]), C.sizeof(out))
    return out


_sizes = {}  # (P, W, H, max_pairs) -> (state bytes, scratch bytes): pure functions of the shape, queried once


def _raster_sizes(lib, P, W, H, cap):
    hit = _sizes.get((P, W, H, cap))
    if hit is None:
        sb, xb = C.c_size_t(0), C.c_size_t(0)
        _lib.check(lib.fsgs_raster_sizes(P, W, H, cap, C.byref(sb), C.byref(xb)), "fsgs_raster_sizes")
        if len(_sizes) >= 256:
            _sizes.clear()
        hit = _sizes[(P, W, H, cap)] = (sb.value, xb.value)
    return hit


last_num_rendered = 0  # R of the most recent forward (bench.py reports it)

# ---- (tile, Gaussian) pair capacity: grow-only estimate per problem shape ----------------------
_capacity = {}


def _capacity_for(P, W, H):
    """max_pairs is split evenly over (tiles x 8) fixed-capacity list segments (csrc/raster_kernels.h binning): start
    with room for 16 keys per segment or 8 pairs per Gaussian, whichever is more; grown on FSGS_ERR_CAPACITY."""
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    return _capacity.get((P, W, H), max(1 << 16, 8 * P, ntiles * 8 * 16))


def _f32c(t):
    return t.detach().contiguous().to(torch.float32)


class RasterState:
    """What forward leaves behind for backward (UPSTREAM: geom/binning/img buffers + num_rendered)."""

    __slots__ = ("buf", "state_bytes", "max_pairs", "num_rendered", "cfg", "P")


def raster_forward(cfg, means3D, colors, opacities, scales, rotations):
    """Launch fsgs_raster_forward; returns (color, depth, radii, RasterState)."""
    lib = _lib.load()
    if not means3D.is_cuda:
        raise RuntimeError("fsgs rasteriser needs CUDA/HIP tensors; there is no CPU fallback")
    dev = means3D.device
    P = int(means3D.shape[0])
    H, W, Cc = cfg.image_height, cfg.image_width, cfg.channels
    out_color = torch.empty((Cc, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    cap = _capacity_for(P, W, H)
    nr = C.c_int64(0)
    with torch.cuda.device(dev):
        stream = _lib.current_stream()
        for _attempt in range(3):
            sbytes, xbytes = _raster_sizes(lib, P, W, H, cap)
            state = torch.empty((sbytes,), dtype=torch.uint8, device=dev)
            scratch = torch.empty((xbytes,), dtype=torch.uint8, device=dev)
            rc = lib.fsgs_raster_forward(
                C.byref(cfg), P, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(opacities), _lib.ptr(scales),
                _lib.ptr(rotations), _lib.ptr(out_color), _lib.ptr(out_depth), _lib.ptr(radii), _lib.ptr(state),
                sbytes, _lib.ptr(scratch), xbytes, cap, C.byref(nr), stream,
            )
            if rc == _lib.FSGS_ERR_CAPACITY and nr.value > cap:
                cap = int(nr.value * 1.25) + 1024
                _capacity[(P, W, H)] = cap
                continue
            _lib.check(rc, "fsgs_raster_forward")
            break
        else:
            raise _lib.FsgsError(_lib.FSGS_ERR_CAPACITY, "fsgs_raster_forward")
    global last_num_rendered
    last_num_rendered = int(nr.value)
    st = RasterState()
    st.buf, st.state_bytes, st.max_pairs, st.num_rendered, st.cfg, st.P = state, sbytes, cap, int(nr.value), cfg, P
    return out_color, out_depth, radii, st


def state_views(st):
    """Typed views into a RasterState buffer (tests / debugging): what UPSTREAM keeps in its geom /
    binning / image buffers."""
    lib = _lib.load()
    H, W, P = st.cfg.image_height, st.cfg.image_width, st.P
    off = (C.c_size_t * 7)()
    _lib.check(lib.fsgs_raster_state_layout(P, W, H, st.max_pairs, off), "fsgs_raster_state_layout")
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)

    def view(i, dtype, count, shape):
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return st.buf[off[i]:off[i] + nbytes].view(dtype).reshape(shape)

    return {
        "xy": view(0, torch.float32, 2 * P, (P, 2)),
        "conic_opacity": view(1, torch.float32, 4 * P, (P, 4)),
        "depth": view(2, torch.float32, P, (P,)),
        "ranges": view(3, torch.int32, 2 * ntiles, (ntiles, 2)),
        "final_T": view(4, torch.float32, H * W, (H, W)),
        "n_contrib": view(5, torch.int32, H * W, (H, W)),
        # per-tile lists live at ranges[tile] inside the whole max_pairs slice (fixed-capacity segments, not contiguous)
        "point_list": view(6, torch.int32, st.max_pairs, (st.max_pairs,)),
    }


def raster_backward(st, means3D, colors, scales, rotations, radii, grad_color):
    lib = _lib.load()
    dev = means3D.device
    P, Cc = st.P, st.cfg.channels
    z = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    dmeans2D, dcolors, dopac = z(P, 3), z(P, Cc), z(P, 1)
    dmeans3D, dscales, drots = z(P, 3), z(P, 3), z(P, 4)
    if P == 0:
        return dmeans2D, dcolors, dopac, dmeans3D, dscales, drots
    scratch = torch.empty((backward_scratch_bytes(st.cfg, P, st.max_pairs, P * 32 + 256),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.fsgs_raster_backward(
            C.byref(st.cfg), P, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(scales), _lib.ptr(rotations),
            _lib.ptr(radii), _lib.ptr(st.buf), st.state_bytes, st.max_pairs, st.num_rendered, _lib.ptr(grad_color),
            _lib.ptr(dmeans2D), _lib.ptr(dcolors), _lib.ptr(dopac), _lib.ptr(dmeans3D), _lib.ptr(dscales),
            _lib.ptr(drots), _lib.ptr(scratch), scratch.numel(), _lib.current_stream(),
        )
    _lib.check(rc, "fsgs_raster_backward")
    return dmeans2D, dcolors, dopac, dmeans3D, dscales, drots


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        if sh is not None and sh.numel() != 0:
            raise NotImplementedError(
                "SH evaluation inside the rasteriser is not on the Free-SurGS path "
                "(shs=None, scene/gaussian_model.py:325); pass colors_precomp")
        if cov3Ds_precomp is not None and cov3Ds_precomp.numel() != 0:
            raise NotImplementedError(
                "cov3D_precomp is not on the Free-SurGS path (cov3D_precomp=None, "
                "scene/gaussian_model.py:309); pass scales and rotations")
        P = means3D.shape[0]
        # the kernels index raw pointers: a wrong trailing dimension would read out of bounds, so say so here
        for name, t, tail in (("means3D", means3D, (3,)), ("scales", scales, (3,)), ("rotations", rotations, (4,))):
            if tuple(t.shape) != (P,) + tail:
                raise ValueError("%s must have shape [%d, %d], got %s" % (name, P, tail[0], tuple(t.shape)))
        if opacities.numel() != P or colors_precomp.shape[0] != P:
            raise ValueError("opacities / colors_precomp must have one row per Gaussian (%d)" % P)
        for name, t in (("colors_precomp", colors_precomp), ("opacities", opacities), ("scales", scales),
                        ("rotations", rotations)):
            if t.device != means3D.device:
                raise ValueError("%s is on %s, means3D on %s" % (name, t.device, means3D.device))
        if not means3D.is_cuda:
            raise RuntimeError("the rasteriser needs CUDA/HIP tensors; there is no CPU fallback")
        m3 = _f32c(means3D)
        col = _f32c(colors_precomp)
        col = col.reshape(P, -1) if P > 0 else col.reshape(0, 3)
        op, sc, rot = _f32c(opacities).reshape(P), _f32c(scales), _f32c(rotations)
        cfg = cached_cfg(raster_settings, col.shape[1] if P > 0 else 3)
        color, depth, radii, st = raster_forward(cfg, m3, col, op, sc, rot)
        ctx.st = st
        ctx.save_for_backward(m3, col, sc, rot, radii)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth):
        # the depth-fork's third output is not back-propagated; Free-SurGS never puts it in a loss
        # (gaussian_renderer/__init__.py:68-70)
        m3, col, sc, rot, radii = ctx.saved_tensors
        gc = _f32c(grad_color)
        # the forward state is read-only here and stays with ctx: a retained graph can be backpropagated again, and
        # without retain_graph autograd frees ctx (and with it the state) as soon as this node is done
        dm2, dcol, dop, dm3, dsc, drot = raster_backward(ctx.st, m3, col, sc, rot, radii, gc)
        return dm3, dm2, None, dcol, dop, dsc, drot, None, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """View-frustum test of UPSTREAM R10 (not called by Free-SurGS): z > 0.2 in view space."""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix.reshape(4, 4)
            p = positions @ V[:3, :3] + V[3, :3]
            return p[:, 2] > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None
        ):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])
        return rasterize_gaussians(
            means3D, means2D,
            empty if shs is None else shs,
            empty if colors_precomp is None else colors_precomp,
            opacities,
            empty if scales is None else scales,
            empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp,
            self.raster_settings,
        )
