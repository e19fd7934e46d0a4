"""CPU reference of the whole `render(...)` for the parity tests (TEST INFRASTRUCTURE, never the product).

Round 6 (VERDICT r5 weak #1a): the glue around the rasteriser is tests/ref_glue.py -- an independent restatement of
gaussian_renderer/__init__.py:49-92 that imports NOTHING from `fsgs_amd`, pinned on CPU against the reference's own `render()`
(tests/golden/render_composition.npz, tests/test_ref_glue.py) -- and its two calls of `GaussianRasterizer` are served by the
CPU oracle (oracle/raster_oracle.c) through `OracleRasterizer`, an nn.Module with UPSTREAM's call signature.  `render_reference`
is what the fused HIP op is compared against end to end.  (`oracle_backend` still lets the CPU harness of
tests/ref_harness.py -- the product's trainer logic on CPU -- rasterise with the oracle; that harness is a different check.)"""
import contextlib

import numpy as np
import torch

from tests import ref_glue


def render_reference(poses, index, pc, gs_grad=True, cam_grad=True):
    """render() on CPU: tests/ref_glue.py around the oracle rasteriser set by `oracle_backend`."""
    return ref_glue.render_two_pass(poses, index, pc, gs_grad=gs_grad, cam_grad=cam_grad, rasterizer=OracleRasterizer)


def cam_from_settings(s):
    """GaussianRasterizationSettings -> the camera dict the oracle takes."""
    n = lambda t: np.asarray(t.detach().cpu().numpy(), np.float64)
    return dict(image_height=int(s.image_height), image_width=int(s.image_width), tanfovx=float(s.tanfovx),
                tanfovy=float(s.tanfovy), viewmatrix=n(s.viewmatrix).reshape(4, 4), projmatrix=n(s.projmatrix).reshape(4, 4),
                bg=n(s.bg).reshape(-1), scale_modifier=float(s.scale_modifier))


class _OracleRasterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, settings, oracle):
        cam = cam_from_settings(settings)
        a = lambda t: np.ascontiguousarray(t.detach().cpu().numpy())
        img, dep, radii, st = oracle.raster_forward(cam, a(means3D), a(colors), a(opacities).reshape(-1), a(scales),
                                                    a(rotations))
        ctx.oracle, ctx.state = oracle, st
        ctx.dtype = means3D.dtype
        radii_t = torch.from_numpy(np.asarray(radii, np.int32).copy())
        ctx.mark_non_differentiable(radii_t)
        return (torch.from_numpy(np.asarray(img).copy()).to(means3D.dtype), radii_t,
                torch.from_numpy(np.asarray(dep).copy()).to(means3D.dtype).unsqueeze(0))

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth):
        g = ctx.oracle.raster_backward(ctx.state, np.ascontiguousarray(g_color.detach().cpu().numpy()))
        t = lambda k, shape=None: torch.from_numpy(np.asarray(g[k]).copy()).to(ctx.dtype).reshape(shape or g[k].shape)
        P = g["means3D"].shape[0]
        return (t("means3D"), t("means2D"), t("colors"), t("opacities", (P, 1)), t("scales"), t("rotations"), None, None)


class OracleRasterizer(torch.nn.Module):
    """Stands in for diff_gaussian_rasterization.GaussianRasterizer on CPU tensors (same call, same 3 returns)."""
    oracle = None

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        assert shs is None and cov3D_precomp is None and colors_precomp is not None
        return _OracleRasterFn.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self.raster_settings,
                                     OracleRasterizer.oracle)


@contextlib.contextmanager
def oracle_backend(oracle):
    """Inside: fsgs_amd.render.render_two_pass rasterises with the CPU oracle (CPU tensors only)."""
    from fsgs_amd import render as R

    old, old_o = R.GaussianRasterizer, OracleRasterizer.oracle
    R.GaussianRasterizer, OracleRasterizer.oracle = OracleRasterizer, oracle
    try:
        yield
    finally:
        R.GaussianRasterizer, OracleRasterizer.oracle = old, old_o


def cpu_cloud(pc):
    """A CPU copy of what render() reads of a GaussianCloud (raw parameters, SH degrees, raster camera)."""
    return ref_glue.Cloud(pc.params, ref_glue.Settings.of(pc.cam), pc.active_sh_degree, pc.max_sh_degree)


def cpu_poses(poses):
    return ref_glue.Poses(poses.r, poses.t, poses.cam_center)


def run_render(render_fn, pc, poses, index, gs_grad, cam_grad, wi, wd, ws, pixel_classes=None):
    """render -> the weighted-sum loss the GPU tests use -> backward; numpy outputs and gradients.
    pixel_classes (list of [H,W] bool masks): additionally the gradients of the loss restricted to each class (the
    backward is linear in the weights) -> (outputs, grads, [grads of class 0, ...])."""
    PARAM_NAMES = ref_glue.Cloud.NAMES

    n = lambda t: None if t is None else t.detach().cpu().numpy().copy()

    def clear():
        for p in pc.params.values():
            p.grad = None
        poses.r.grad = poses.t.grad = None

    def collect(pkg):
        grads = {k: n(pc.params[k].grad) for k in PARAM_NAMES}
        grads["viewspace"] = n(pkg["viewspace_points"].grad) if gs_grad else None
        grads["r"], grads["t"] = n(poses.r.grad), n(poses.t.grad)
        return grads

    clear()
    pkg = render_fn(poses, index, pc, gs_grad=gs_grad, cam_grad=cam_grad)
    loss_of = lambda m: ((pkg["render"] * (wi * m)).sum() + (pkg["render_dep"] * (wd * m)).sum() +
                         (pkg["render_opacity"] * (ws * m)).sum())
    out = {"render": n(pkg["render"]), "render_dep": n(pkg["render_dep"]), "sil": n(pkg["render_opacity"]),
           "unc": n(pkg["uncertainty"]), "radii": n(pkg["radii"]), "vis": n(pkg["visibility_filter"]),
           "presence": n(pkg["presence_mask"])}
    per_class = []
    for m in (pixel_classes or []):
        clear()
        if gs_grad and pkg["viewspace_points"].grad is not None:
            pkg["viewspace_points"].grad = None
        loss_of(torch.as_tensor(m, dtype=wi.dtype, device=wi.device)).backward(retain_graph=True)
        per_class.append(collect(pkg))
    clear()
    if gs_grad and pkg["viewspace_points"].grad is not None:
        pkg["viewspace_points"].grad = None
    loss_of(1.0).backward()
    grads = collect(pkg)
    return (out, grads, per_class) if pixel_classes is not None else (out, grads)


_oracle64 = None


def _to_double(c, p):
    """the same cloud / poses / camera with every tensor in float64 (the glue statements are dtype-agnostic)."""
    for k in c.params:
        c.params[k] = c.params[k].detach().double().requires_grad_(True)
    for k, v in c.variables.items():
        if torch.is_tensor(v) and v.is_floating_point():
            c.variables[k] = v.double()
    c.cam = ref_glue.Settings.of(c.cam, conv=lambda t: t.detach().double())
    p.r = p.r.detach().double().requires_grad_(True)
    p.t = p.t.detach().double().requires_grad_(True)
    p.cam_center = p.cam_center.double()
    return c, p


def reference_render_with_amplitudes(oracle, pc, poses, index, gs_grad, cam_grad, wi, wd, ws, roundoff=True):
    """The CPU reference render (nominal thresholds) and, element by element, how far each of its outputs / gradients
    moves when the oracle's decision thresholds shift by a rounding-sized hair (Oracle.set_thresholds, +1 / -1),
    plus (roundoff) twice the distance of this fp32 reference from the SAME sequence evaluated in fp64 (torch double
    glue + the fp64 oracle build) -- the allowance of tests/util.py:assert_close_attributed, see
    Oracle.flip_amplitudes for the reasoning.  -> (outputs, grads, amp_outputs, amp_grads)."""
    global _oracle64
    render_two_pass = render_reference

    c, p = cpu_cloud(pc), cpu_poses(poses)
    w = [t.detach().cpu() for t in (wi, wd, ws)]
    # gradients per 2x2-interleaved pixel class: flips of one Gaussian at several pixels must not cancel each other
    # inside the amplitude (see Oracle.flip_amplitudes)
    masks = oracle.pixel_class_masks(*w[1].shape[-2:])
    runs = []
    try:
        with oracle_backend(oracle):
            for sign in (0, +1, -1):
                oracle.set_thresholds(sign)
                runs.append(run_render(render_two_pass, c, p, index, gs_grad, cam_grad, *w, pixel_classes=masks))
                if sign == 0:
                    # the glue in front of the rasteriser differs between the implementations in the last bit of the
                    # view-space depth: neighbours of the depth order within a few ulp may be sorted either way
                    oracle.find_order_ties()
    finally:
        oracle._order_h = None
        oracle.set_thresholds(0)
    run64 = None
    if roundoff:
        if _oracle64 is None:
            from oracle.fsgs_oracle import Oracle

            _oracle64 = Oracle(np.float64)
        c64, p64 = _to_double(cpu_cloud(pc), cpu_poses(poses))
        with oracle_backend(_oracle64):
            run64 = run_render(render_two_pass, c64, p64, index, gs_grad, cam_grad, *[t.double() for t in w])

    def spread(sel, rs):
        a = [sel(r) for r in rs]
        if a[0] is None:
            return None
        a = [np.asarray(x, np.float64) for x in a]
        return np.maximum(np.maximum(np.abs(a[1] - a[0]), np.abs(a[2] - a[0])), np.abs(a[1] - a[2]))

    def with_roundoff(amp, nominal, exact):
        if amp is None or run64 is None:
            return amp
        return amp + 2 * np.abs(np.asarray(exact, np.float64) - np.asarray(nominal, np.float64))

    amp_o = {k: with_roundoff(spread(lambda r, k=k: r[0][k], runs), runs[0][0][k], None if run64 is None else run64[0][k])
             for k in ("render", "render_dep", "sil", "unc")}
    amp_o["radii"] = (runs[1][0]["radii"] != runs[0][0]["radii"]) | (runs[2][0]["radii"] != runs[0][0]["radii"])
    amp_o["presence"] = (runs[1][0]["presence"] != runs[0][0]["presence"]) | (runs[2][0]["presence"] != runs[0][0]["presence"])
    amp_g = {}
    for k in runs[0][1]:
        if runs[0][1][k] is None:
            amp_g[k] = None
            continue
        a = sum(spread(lambda r, k=k, c=c: r[2][c][k], runs) for c in range(len(masks)))
        amp_g[k] = with_roundoff(a, runs[0][1][k], None if run64 is None else run64[1][k])
    return runs[0][0], runs[0][1], amp_o, amp_g
