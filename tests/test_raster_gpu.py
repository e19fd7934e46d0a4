"""GPU parity of the HIP rasteriser against the CPU oracle, through the drop-in boundary
(`diff_gaussian_rasterization.GaussianRasterizer` -> C ABI).  Tolerances: SURVEY.md s8d --
images max rel err <= 1e-4 (abs floor 1e-6... widened to 1e-5 of full scale for blended sums),
per-Gaussian gradients <= 1e-4 of the tensor's inf-norm, radii / visibility exact."""
import numpy as np
import pytest
import torch

from fsgs_amd import synth
from tests.util import (ATTRIBUTION_LOG, assert_close_attributed, dump_attribution_log, assert_sign_balanced, c1_poses, sh0_colors, sign_balance,
                        to_camera_frame)

from oracle.fsgs_oracle import usable_cores

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _settings(cam):
    from diff_gaussian_rasterization import GaussianRasterizationSettings

    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    return GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=cam["tanfovx"],
        tanfovy=cam["tanfovy"], bg=t(cam["bg"]), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]).unsqueeze(0),
        projmatrix=t(cam["projmatrix"]).unsqueeze(0), sh_degree=0, campos=t(cam["campos"]), prefiltered=False,
        debug=False)


# image and depth at the operator boundary (same fp32 inputs on both sides), of (max |want| + 1), on BASELINE.json's own
# configurations (C1, C2, C4: _compare(..., image_tol=IMAGE_TOL)); the randomised sweeps keep SURVEY's 1e-4 -- see _compare
IMAGE_TOL = 1e-5


def _run_hip(cam, xyz, col, op, sc, rot, dL):
    from diff_gaussian_rasterization import GaussianRasterizer

    T = lambda a, g=True: torch.tensor(np.asarray(a, np.float32), device=DEV, requires_grad=g)
    m3, c, o, s, r = T(xyz), T(col), T(np.asarray(op).reshape(-1, 1)), T(sc), T(rot)
    m2 = torch.zeros_like(m3, requires_grad=True) + 0
    m2.retain_grad()
    img, radii, depth = GaussianRasterizer(raster_settings=_settings(cam))(
        means3D=m3, means2D=m2, opacities=o, colors_precomp=c, scales=s, rotations=r)
    (img * torch.tensor(dL, device=DEV)).sum().backward()
    g = {"means3D": m3.grad, "means2D": m2.grad, "colors": c.grad, "opacities": o.grad.reshape(-1), "scales": s.grad,
         "rotations": r.grad}
    n = lambda t: t.detach().cpu().numpy()
    return n(img), n(depth)[0], n(radii), {k: n(v) for k, v in g.items()}


def _compare(oracle, cam, xyz, col, op, sc, rot, seed=0, strict=False, tag=None, image_tol=1e-4):
    """HIP vs the fp32 oracle on the same inputs.  Tolerances per SURVEY.md s8d (1e-4 of the tensor's inf-norm; radii
    and visibility exact).  Every element beyond the tolerance needs a WITNESS: the oracle's own value there must move
    by a comparable amount when the decision thresholds shift by a rounding-sized hair
    (tests/util.py:assert_close_attributed, oracle Oracle.flip_amplitudes); strict=True (small scenes) allows no
    outliers at all.  -> (num_rendered, {tensor: (outliers, fragile elements)})."""
    H, W = cam["image_height"], cam["image_width"]
    Cc = np.asarray(col).shape[1]
    P = len(xyz)
    dL = (np.random.default_rng(seed).uniform(-1, 1, (Cc, H, W)) / (Cc * H * W)).astype(np.float32)
    img, dep, radii, g = _run_hip(cam, xyz, col, op, sc, rot, dL)
    oi, od, orad, st = oracle.raster_forward(cam, xyz, col, op, sc, rot)
    og = oracle.raster_backward(st, dL)
    R = st.num_rendered

    def check(amp, oi, od, orad, og):
        zero = lambda a: np.zeros(np.shape(a))
        # ceil(3 sigma) may land on the other side of an integer: only where the oracle's own radius moves with the hair
        rogue_r = (radii != orad) & ~(amp["radii"] if amp is not None else np.zeros(P, bool))
        assert not rogue_r.any(), "radii differ at %s without a ceil() near-tie" % np.nonzero(rogue_r)[0][:8].tolist()
        assert int(((radii > 0) != (orad > 0)).sum()) == 0, "visibility filter differs"
        stats = {}
        # image / depth, of (max + 1).  On BASELINE.json's configurations the callers pass image_tol = IMAGE_TOL = 1e-5: ten
        # times the largest error measured where no near-tie is in play (C1 poses 2 and 6: image 4.8e-7, depth 8.2e-7 of that
        # scale; the 99.99th percentile at C2 / C4 is 1.1e-6 .. 2.3e-6; profiles/r04_full_size_parity.jsonl), not SURVEY's
        # blanket 1e-4, which left two orders of magnitude of slack there.  The randomised sweeps keep 1e-4:
        # their needle and screen-filling footprints are only good to 6 kappa eps (DESIGN s4), one such Gaussian moves a
        # quarter of a 100 x 100 image by 3e-5 -- at 1e-5, 7 of 3 000 soak seeds have hundreds of (witnessed) elements
        # beyond the tolerance (profiles/r04_soak_at_image_tol_1e-5.txt).  Whatever lies beyond the tolerance needs a witness either way.
        stats["image"] = assert_close_attributed(img, oi, zero(oi) if amp is None else amp["image"], "image", tol=image_tol, floor=1.0, tag=tag)
        stats["depth"] = assert_close_attributed(dep, od, zero(od) if amp is None else amp["depth"], "depth", tol=image_tol, floor=1.0, tag=tag)
        # an analytically-zero gradient (d/drotation of an isotropic Gaussian) is cancellation round-off of
        # terms of size ~|dL/dscale|*|scale| in both implementations: floor each norm at 1e-3 of the largest
        # gradient tensor, i.e. an absolute tolerance of 1e-7 of that for such tensors
        floor = 1e-3 * max(float(np.abs(v).max()) for v in og.values())
        for k in ("means3D", "means2D", "colors", "opacities", "scales", "rotations"):
            a, b = g[k].reshape(P, -1), og[k].reshape(P, -1)
            stats[k] = assert_close_attributed(a, b, zero(b) if amp is None else amp[k], k, floor=floor, tag=tag)
        return stats

    n_log = len(ATTRIBUTION_LOG)
    try:  # most cases agree everywhere: the allowances (3 threshold settings x 4 pixel classes + fp64) are only
        return R, check(None, oi, od, orad, og)  # computed when some element is beyond the plain tolerance
    except AssertionError:
        if strict:
            raise
        del ATTRIBUTION_LOG[n_log:]  # (the records of the tensors that passed before the first outlier)
    amp, (oi, od, orad, og, st) = oracle.flip_amplitudes(cam, xyz, col, op, sc, rot, dL)
    return R, check(amp, oi, od, orad, og)


def test_c1_init_scene_eight_poses(oracle32):
    """BASELINE.json configs[0]: 640x512, 20k init Gaussians, 8 poses, pure raster fwd/bwd."""
    oracle32.set_threads(usable_cores())
    W, H, P = 640, 512, 20000
    cam = synth.make_camera(W, H)
    sc = synth.init_scene(W, H, P, seed=0)
    s, r, o = synth.activate(sc)
    col = sh0_colors(sc)
    for i, w2c in enumerate(c1_poses()):
        xyz = to_camera_frame(sc["_xyz"], w2c)
        R, stats = _compare(oracle32, cam, xyz, col, o.reshape(-1), s, r, seed=i, tag="c1/%d" % i, image_tol=IMAGE_TOL)
        assert R > P
        # the witnessed outliers are a handful, and so is the set of fragile pixels the allowance applies to
        assert stats["image"][0] <= 40 and stats["depth"][0] <= 40, stats
        # the achieved error of every tensor, next to the full-size records
        dump_attribution_log("r06_full_size_parity", dict(
            test="raster_op", cfg="C1", pose=i, P=P, num_rendered_upstream=R,
            tensors={k: dict(outliers=v.outliers, size=v.size, max_err=v.max_err, p9999=v.p9999,
                             max_err_plain=v.max_err_plain, scale=v.scale, max_err_zero_amp=v.max_err_zero_amp,
                             zero_amp_fraction=v.zero_amp_fraction) for k, v in stats.items()}))


def test_trained_like_scene_with_view_matrix(oracle32):
    """anisotropic / rotated / mixed-opacity Gaussians and a non-identity raster viewmatrix."""
    W, H, P = 320, 256, 6000
    w2c = synth.pose_matrix(**synth.PERTURBED_POSE)
    cam = synth.make_camera(W, H, w2c=w2c)
    sc = synth.trained_like_scene(W, H, P, seed=2, base_ratio=0.02)
    s, r, o = synth.activate(sc)
    rng = np.random.default_rng(5)
    col = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    _compare(oracle32, cam, sc["_xyz"], col, o.reshape(-1), s, r, seed=9)


def test_one_channel(oracle32):
    """channels = 1 (include/fsgs.h: 1, 3 or 6): the narrowest instantiation of every blend kernel, both flavours (conftest)"""
    W, H, P = 200, 136, 3000
    cam = synth.make_camera(W, H)
    sc = synth.trained_like_scene(W, H, P, seed=4, base_ratio=0.02)
    s, r, o = synth.activate(sc)
    col = np.random.default_rng(6).uniform(0, 1, (P, 1)).astype(np.float32)
    _compare(oracle32, cam, to_camera_frame(sc["_xyz"], synth.pose_matrix(**synth.PERTURBED_POSE)), col, o.reshape(-1), s, r, seed=3)


def test_six_channel_fused_layout(oracle32):
    W, H, P = 160, 128, 1500
    cam = synth.make_camera(W, H)
    xyz, col, op, s, r = synth.random_small_scene(P, cam, seed=4, channels=6)
    _compare(oracle32, cam, xyz.astype(np.float32), col.astype(np.float32), op, s.astype(np.float32),
             r.astype(np.float32), seed=3, strict=True)


@pytest.mark.parametrize("scene", ["c1_init", "trained", "c2_trained"])
def test_final_T_and_last_contributor_match_oracle_by_id(oracle32, scene):
    """The image state kept for the backward: final_T per pixel, and n_contrib.  The HIP lists are the oracle's lists
    minus unreachable pairs, so the POSITION of the last contributor differs while the Gaussian it names must not:
    both are mapped to Gaussian ids through their own sorted lists and compared exactly -- except at pixels whose
    last contributor moves when the oracle's thresholds shift by a rounding-sized hair (Oracle.flip_amplitudes)."""
    from fsgs_amd import rasterizer
    from fsgs_amd.trainer import settings_from_cam

    from oracle.fsgs_oracle import usable_cores

    if scene == "c1_init":
        W, H, P = 640, 512, 20000
        sc = synth.init_scene(W, H, P, seed=0)
        xyz = to_camera_frame(sc["_xyz"], c1_poses()[1])
        col = sh0_colors(sc)
    elif scene == "c2_trained":  # BASELINE.json configs[1] size: 5120 tiles, lists of 200-400 entries
        W, H, P = 1280, 1024, 300_000
        sc = synth.trained_like_scene(W, H, P, seed=0)
        xyz = to_camera_frame(sc["_xyz"], c1_poses()[1])
        col = sh0_colors(sc)
        oracle32.set_threads(usable_cores())
    else:
        W, H, P = 320, 256, 6000
        sc = synth.trained_like_scene(W, H, P, seed=2, base_ratio=0.02)
        xyz = sc["_xyz"]
        col = np.random.default_rng(5).uniform(0, 1, (P, 3)).astype(np.float32)
    cam = synth.make_camera(W, H)
    s, r, o = synth.activate(sc)
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=DEV)
    cfg = rasterizer.make_cfg(settings_from_cam(cam, DEV), 3)
    img, depth, radii, st = rasterizer.raster_forward(cfg, T(xyz), T(col), T(o.reshape(-1)), T(s), T(r))
    v = {k: t.cpu().numpy() for k, t in rasterizer.state_views(st).items()}
    dL = np.zeros((3, H, W), np.float32)
    amp, (oi, od, orad, og, ost) = oracle32.flip_amplitudes(cam, xyz, col, o.reshape(-1), s, r, dL)
    gx = (W + 15) // 16
    yy, xx = np.mgrid[0:H, 0:W]
    tile = (yy // 16) * gx + xx // 16

    def last_id(ranges, plist, n_contrib):
        n = n_contrib.reshape(H, W).astype(np.int64)
        pos = ranges[tile, 0].astype(np.int64) + n - 1
        return np.where(n > 0, plist[np.clip(pos, 0, len(plist) - 1)].astype(np.int64), -1)

    mine = last_id(v["ranges"], v["point_list"], v["n_contrib"])
    ref = last_id(ost.ranges(), ost.point_list(), ost.n_contrib())
    fragile = amp["n_contrib"].reshape(H, W)
    rogue = (mine != ref) & ~fragile
    assert not rogue.any(), "last contributor differs at %d pixels without a near-tie, e.g. %s" % (
        int(rogue.sum()), np.argwhere(rogue)[:4].tolist())
    oracle32.set_threads(1)
    assert int((mine != ref).sum()) <= max(60, 1e-4 * H * W) and int(fragile.sum()) < 0.02 * H * W, (
        int((mine != ref).sum()), int(fragile.sum()))
    # positions: never beyond the tile's list, and a pixel nobody reached has none
    lens = (v["ranges"][:, 1] - v["ranges"][:, 0])[tile]
    assert (v["n_contrib"].reshape(H, W) <= lens).all()
    res = assert_close_attributed(v["final_T"].reshape(-1), ost.final_T().reshape(-1), amp["final_T"].reshape(-1), "final_T",
                                  floor=1.0, tag="final_T/" + scene)
    # the sharpest view of a one-sided path: final_T below the oracle's = this side blended a pair the oracle skipped
    # (alpha >= 1/255 or T >= 1e-4 decided the other way), above = the reverse; over a frame neither may dominate
    pos, neg, z = sign_balance([res._asdict()])
    print("final_T %s: %d witnessed outliers of %d pixels (%d above, %d below the oracle, z = %.2f); last contributor "
          "differs at %d pixels" % (scene, res.outliers, res.size, pos, neg, z, int((mine != ref).sum())))
    assert_sign_balanced([res._asdict()], "final_T " + scene)


def test_edge_cases_empty_ragged_and_culled(oracle32):
    from diff_gaussian_rasterization import GaussianRasterizer

    # ragged image (not a multiple of the tile), everything behind the near plane, empty cloud
    W, H = 77, 45
    cam = synth.make_camera(W, H)
    xyz, col, op, s, r = synth.random_small_scene(300, cam, seed=8)
    _compare(oracle32, cam, xyz.astype(np.float32), col.astype(np.float32), op, s.astype(np.float32),
             r.astype(np.float32))
    behind = xyz.copy()
    behind[:, 2] = 0.1
    img, dep, radii, g = _run_hip(cam, behind, col, op, s, r, np.ones((3, H, W), np.float32))
    assert (radii == 0).all() and np.all(img == 1.0) and all(np.all(v == 0) for v in g.values())
    e = torch.zeros((0, 3), device=DEV)
    img, radii, depth = GaussianRasterizer(raster_settings=_settings(cam))(
        means3D=e, means2D=e, opacities=e[:, :1], colors_precomp=e, scales=e, rotations=torch.zeros((0, 4), device=DEV))
    assert img.shape == (3, H, W) and bool((img == 1).all()) and radii.numel() == 0


def test_thresholds_alpha_clamp_and_termination(oracle32):
    """Stacked near-opaque Gaussians: 0.99 clamp, T < 1e-4 stop, alpha < 1/255 skip all exercised."""
    cam = synth.make_camera(64, 48)
    K = cam["K"]
    n = 40
    z = 0.5 + 0.02 * np.arange(n)
    u, v = 32.5, 24.5
    xyz = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1).astype(np.float32)
    s = (np.full((n, 3), 6.0) * z[:, None] / K[0, 0]).astype(np.float32)
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1))
    op = np.linspace(0.3, 0.999, n).astype(np.float32)
    col = np.random.default_rng(1).uniform(0, 1, (n, 3)).astype(np.float32)
    _compare(oracle32, cam, xyz, col, op, s, rot)


# BASELINE.json C2, C4, and 4x C4's cloud (the largest the densification schedule has been seen to reach is ~1.6 M)
@pytest.mark.parametrize("W,H,P", [(1280, 1024, 300_000), (1920, 1080, 1_000_000), (1920, 1080, 4_000_000)])
def test_full_size_properties(W, H, P):
    """full benchmark sizes, size-independent properties (the oracle itself meets C2 and C4 in
    tests/test_full_size_oracle_gpu.py; it does not reach 4 M): silhouette identity (bg = 1: sum(alpha T) + T_final
    = 1), linearity of the image in the colours, and linearity of every gradient in dL/dpixel."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from simple_knn._C import distCUDA2

    cam = synth.make_camera(W, H)
    knn = lambda pts: distCUDA2(torch.tensor(pts, device=DEV)).cpu().numpy()
    sc = synth.trained_like_scene(W, H, P, seed=0, knn_fn=knn)
    s, r, o = synth.activate(sc)
    T = lambda a: torch.tensor(a, device=DEV)
    m3, sc_t, r_t, o_t = T(sc["_xyz"]), T(s), T(r), T(o)
    rast = GaussianRasterizer(raster_settings=_settings(cam))
    m2 = torch.zeros_like(m3)
    rng = np.random.default_rng(0)
    c1 = T(rng.uniform(0, 1, (P, 3)).astype(np.float32))
    c2 = T(rng.uniform(0, 1, (P, 3)).astype(np.float32))
    ones = torch.ones_like(c1).requires_grad_(True)
    sil, radii, _ = rast(means3D=m3, means2D=m2, opacities=o_t, colors_precomp=ones, scales=sc_t, rotations=r_t)
    assert float((sil.detach() - 1).abs().max()) < 2e-5  # bg = 1 -> every plane is exactly 1
    i1, _, _ = rast(means3D=m3, means2D=m2, opacities=o_t, colors_precomp=c1, scales=sc_t, rotations=r_t)
    i2, _, _ = rast(means3D=m3, means2D=m2, opacities=o_t, colors_precomp=c2, scales=sc_t, rotations=r_t)
    i3, _, _ = rast(means3D=m3, means2D=m2, opacities=o_t, colors_precomp=0.25 * c1 + 0.75 * c2, scales=sc_t,
                    rotations=r_t)
    assert float((i3 - (0.25 * i1 + 0.75 * i2)).abs().max()) < 2e-5
    assert int((radii > 0).sum()) > P // 2
    # the backward is linear in dL/dpixel: grad(a dL1 + b dL2) = a grad(dL1) + b grad(dL2) for every input
    def grads(dL):
        leaves = [x.detach().clone().requires_grad_(True) for x in (m3, c1, o_t, sc_t, r_t)]
        img, _, _ = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], colors_precomp=leaves[1], scales=leaves[3],
                         rotations=leaves[4])
        (img * dL).sum().backward()
        return [x.grad for x in leaves]

    gen = torch.Generator(device=DEV).manual_seed(1)
    dL1 = (torch.rand((3, H, W), device=DEV, generator=gen) - 0.5) / (H * W)
    dL2 = (torch.rand((3, H, W), device=DEV, generator=gen) - 0.5) / (H * W)
    g1, g2, g3 = grads(dL1), grads(dL2), grads(0.3 * dL1 - 1.7 * dL2)
    for a, b, c, name in zip(g1, g2, g3, ("means3D", "colors", "opacities", "scales", "rotations")):
        want = 0.3 * a - 1.7 * b
        scale = max(float(want.abs().max()), 1e-12)
        assert float((c - want).abs().max()) <= 2e-4 * scale, name  # atomics order + fp32 cancellation


@pytest.mark.parametrize("P", [6000, 24000])  # tile lists up to ~200 keys (rank sort) / ~500 keys (two halves + merge)
def test_binning_is_upstream_order_minus_unreachable_pairs(oracle32, P):
    """The HIP binning keeps UPSTREAM's (tile, depth, index) order and drops only (tile, Gaussian) pairs
    that no pixel of the tile can reach (alpha < 1/255 everywhere): per tile the HIP list must be a
    subsequence of the oracle's rect-based list, and every dropped pair must be unreachable."""
    from fsgs_amd import rasterizer
    from fsgs_amd.trainer import settings_from_cam

    W, H = 320, 256
    cam = synth.make_camera(W, H)
    sc = synth.trained_like_scene(W, H, P, seed=3, base_ratio=0.02)
    s, r, o = synth.activate(sc)
    col = np.random.default_rng(0).uniform(0, 1, (P, 3)).astype(np.float32)
    T = lambda a: torch.tensor(a, device=DEV)
    cfg = rasterizer.make_cfg(settings_from_cam(cam, DEV), 3)
    img, depth, radii, st = rasterizer.raster_forward(cfg, T(sc["_xyz"]), T(col), T(o.reshape(-1)), T(s), T(r))
    v = {k: t.cpu().numpy() for k, t in rasterizer.state_views(st).items()}
    oi, od, orad, ost = oracle32.raster_forward(cam, sc["_xyz"], col, o.reshape(-1), s, r)
    assert st.num_rendered < ost.num_rendered  # something was culled
    lens = v["ranges"][:, 1] - v["ranges"][:, 0]
    assert (lens.max() <= 256) if P == 6000 else (256 < lens.max() <= 512 and (lens > 256).mean() > 0.1), (lens.max(), (lens > 256).mean())
    o_ranges, o_list = ost.ranges(), ost.point_list().astype(np.int64)
    xy, co = ost.xy().astype(np.float64), ost.conic_opacity().astype(np.float64)
    gx = (W + 15) // 16
    dropped_total = 0
    yy, xx = np.mgrid[0:16, 0:16]
    for tile in range(o_ranges.shape[0]):
        mine = v["point_list"][v["ranges"][tile, 0]:v["ranges"][tile, 1]].astype(np.int64)
        ref = o_list[o_ranges[tile, 0]:o_ranges[tile, 1]]
        # subsequence check (order preserved)
        it = iter(ref.tolist())
        assert all(any(g == h for h in it) for g in mine.tolist()), "tile %d: not a subsequence" % tile
        dropped = np.setdiff1d(ref, mine)
        dropped_total += len(dropped)
        for g in dropped[:8]:
            dx = xy[g, 0] - ((tile % gx) * 16 + xx)
            dy = xy[g, 1] - ((tile // gx) * 16 + yy)
            power = -0.5 * (co[g, 0] * dx * dx + co[g, 2] * dy * dy) - co[g, 1] * dx * dy
            alpha = np.where(power > 0, 0.0, co[g, 3] * np.exp(np.minimum(power, 0)))
            assert alpha.max() < 1.0 / 255.0, "tile %d: reachable pair %d was culled" % (tile, g)
    assert dropped_total == ost.num_rendered - st.num_rendered


def test_wave_transposing_reduction_selftest():
    """the permlane-swap / DPP transposing reduction of blend_bwd, isolated: out[l] = sum_lanes in[lane][l]."""
    import ctypes as C

    from fsgs_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(0)
    m = rng.standard_normal((64, 64)).astype(np.float32)
    m[:, 5] = np.arange(64)  # asymmetric columns catch lane/value mix-ups
    a = torch.tensor(m, device=DEV)
    out = torch.zeros(64, device=DEV)
    _lib.check(lib.fsgs_selftest_transpose_reduce(_lib.ptr(a), _lib.ptr(out), _lib.current_stream()), "selftest")
    np.testing.assert_allclose(out.cpu().numpy(), m.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    for width in (32, 16):  # lane l receives column l / (64 / width)
        _lib.check(lib.fsgs_selftest_transpose_reduce_n(_lib.ptr(a), _lib.ptr(out), width, _lib.current_stream()),
                   "selftest")
        want = m.astype(np.float64).sum(0)[np.arange(64) // (64 // width)]
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # the sparse variants blend_bwd runs (two Gaussians x 16 slots / x 8 slots, of which 12 / 5 are reduced): every used
    # slot must end up, with its column total, in at least one lane; lanes owning an unused slot are ignored
    out2 = torch.zeros(128, device=DEV)
    for width, slots, used in ((3212, 32, 12), (1605, 16, 5)):
        _lib.check(lib.fsgs_selftest_transpose_reduce_n(_lib.ptr(a), _lib.ptr(out2), width, _lib.current_stream()), "selftest")
        o = out2.cpu().numpy()
        tot, slot = o[:64], o[64:].astype(int)
        per = slots // 2
        live = (slot % per) < used
        assert set(slot[live]) == {s_ for s_ in range(slots) if s_ % per < used}, sorted(set(slot[live]))
        np.testing.assert_allclose(tot[live], m.astype(np.float64).sum(0)[slot[live]], rtol=1e-5, atol=1e-5)
        assert np.array_equal(slot[0::2], slot[1::2])  # lanes l and l ^ 1 own the same slot (the kernel uses the even one)
        if width == 1605:  # ... and so do lanes l and l ^ 32
            assert np.array_equal(slot[:32], slot[32:]) and np.allclose(tot[:32][live[:32]], tot[32:][live[32:]])


def test_alpha_evaluation_is_unbiased_around_the_skip_threshold():
    """VERDICT r2 weak #3, asked directly: does the blend kernels' alpha -- the exponent pre-scaled by log2(e) and ONE
    v_exp_f32 instead of expf (fsgs_device.h: splat_coef / splat_alpha) -- resolve near-ties against 1/255 one way?
    2 M samples whose true alpha (float64) lies within +-2e-3 relative of the threshold, on footprints from round to
    needle-shaped: the relative error of HIP's alpha stays inside an fp32 evaluation's budget and its MEAN is a small
    fraction of that budget (no systematic sign),
    every decision that differs from the float64 one sits within the oracle's flip margin of the threshold
    (Oracle.FLIP_MARGINS: 4e-4 + 5e-7 x the exponent's cancelling terms), and the differing decisions split evenly
    between 'blended although below' and 'skipped although above'."""
    from fsgs_amd import _lib
    from oracle.fsgs_oracle import Oracle

    lib = _lib.load()
    rng = np.random.default_rng(7)
    n = 2_000_000
    # conic of a footprint with sigmas s1 >= s2 (pixels) rotated by th; offset d at a random direction
    s1 = rng.uniform(0.6, 30.0, n)
    s2 = s1 * rng.uniform(0.03, 1.0, n)
    s2 = np.maximum(s2, 0.55)
    th = rng.uniform(0, np.pi, n)
    c, s_ = np.cos(th), np.sin(th)
    ia, ib = 1.0 / (s1 * s1), 1.0 / (s2 * s2)
    A = c * c * ia + s_ * s_ * ib
    B = c * s_ * (ia - ib)
    Cc = s_ * s_ * ia + c * c * ib
    o = rng.uniform(0.01, 1.0, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    ux, uy = np.cos(phi), np.sin(phi)
    q1 = 0.5 * (A * ux * ux + Cc * uy * uy) + B * ux * uy          # power = -q1 r^2
    target = (1.0 / 255.0) * (1.0 + rng.uniform(-2e-3, 2e-3, n))    # the alpha to land on
    r = np.sqrt(np.maximum(np.log(o / target), 0.0) / q1)
    px = np.floor(rng.uniform(0, 1900, n))
    py = np.floor(rng.uniform(0, 1000, n))
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    rec = np.stack([f32(px + r * ux), f32(py + r * uy), f32(A), f32(B), f32(Cc), f32(o), f32(px), f32(py)], 1)
    # truth on the ROUNDED inputs, in float64
    R = rec.astype(np.float64)
    dx, dy = R[:, 0] - R[:, 6], R[:, 1] - R[:, 7]
    t1, t2, t3 = 0.5 * R[:, 2] * dx * dx, 0.5 * R[:, 4] * dy * dy, R[:, 3] * dx * dy
    power = -(t1 + t2) - t3
    alpha = R[:, 5] * np.exp(power)
    use = (power < -1e-3) & (np.abs(alpha * 255.0 - 1.0) < 5e-3)
    assert use.mean() > 0.8
    d_in, d_out = torch.tensor(rec, device=DEV), torch.zeros((n, 2), device=DEV)
    _lib.check(lib.fsgs_selftest_splat_alpha(n, _lib.ptr(d_in), _lib.ptr(d_out), _lib.current_stream()), "selftest")
    out = d_out.cpu().numpy().astype(np.float64)
    rel = (out[use, 0] - alpha[use]) / alpha[use]
    cond = (np.abs(t1) + np.abs(t2) + np.abs(t3))[use]
    # error budget of any fp32 evaluation: ~1 ulp from the multiply + the exponent's rounding x its condition
    budget = 3e-7 + 5e-7 * cond
    assert np.all(np.abs(rel) <= budget), float(np.max(np.abs(rel) / budget))
    mean_bias = float(np.mean(rel / budget))
    assert abs(mean_bias) <= 0.1, mean_bias            # no systematic sign: <= 10 % of the per-sample budget
    hip = out[use, 1] > 0.5
    want = alpha[use] >= 1.0 / 255.0
    differ = hip != want
    m = Oracle.FLIP_MARGINS
    margin = m["alpha_min"] + m["alpha_cond"] * cond
    assert np.all(np.abs(alpha[use] * 255.0 - 1.0)[differ] <= margin[differ]), "a decision flipped outside the flip margin"
    more, less = int((hip & ~want).sum()), int((~hip & want).sum())
    print("alpha near 1/255: %d samples, %d decisions differ from float64 (%d blended although below, %d skipped although "
          "above); mean normalised error %.4f" % (int(use.sum()), int(differ.sum()), more, less, mean_bias))
    if more + less >= 50:
        assert 0.3 <= more / (more + less) <= 0.7, (more, less)


def test_heavy_tile_takes_the_global_memory_sort_path(oracle32):
    """> 2048 Gaussians in one tile (LDS sort capacity) -> in-place global-memory bitonic fallback; duplicate
    depths included so the (depth, index) tie order is exercised."""
    cam = synth.make_camera(64, 48)
    K = cam["K"]
    n = 3000
    rng = np.random.default_rng(3)
    z = np.round(rng.uniform(0.5, 1.5, n), 3)  # many exact depth ties
    u = rng.uniform(18, 30, n)
    v = rng.uniform(18, 30, n)
    xyz = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1).astype(np.float32)
    s = (np.full((n, 3), 1.2) * z[:, None] / K[0, 0]).astype(np.float32)
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1))
    op = rng.uniform(0.01, 0.05, n).astype(np.float32)
    col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    _compare(oracle32, cam, xyz, col, op, s, rot)


def test_large_grid_and_screen_filling_gaussians(oracle32):
    """2304x1040 = 144x65 = 9360 tiles (> 8192: the generic tile-scan path) with a few Gaussians whose 3-sigma
    rect spans the whole grid next to many small ones: the flattened binning walker hands one Gaussian's
    thousands of candidate tiles to all the lanes of its wave."""
    cam = synth.make_camera(2304, 1040)
    K = cam["K"]
    rng = np.random.default_rng(11)
    n_small, n_big = 600, 3
    z = rng.uniform(0.6, 1.4, n_small + n_big)
    u = rng.uniform(0, 2304, n_small + n_big)
    v = rng.uniform(0, 1040, n_small + n_big)
    xyz = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1).astype(np.float32)
    sig_px = np.concatenate([rng.uniform(1.0, 12.0, n_small), [500.0, 350.0, 800.0]])
    s = (sig_px[:, None] * rng.uniform(0.5, 1.0, (n_small + n_big, 3)) * z[:, None] / K[0, 0]).astype(np.float32)
    q = rng.normal(size=(n_small + n_big, 4))
    rot = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    op = np.concatenate([rng.uniform(0.05, 0.9, n_small), [0.3, 0.02, 0.6]]).astype(np.float32)
    col = rng.uniform(0, 1, (n_small + n_big, 3)).astype(np.float32)
    R, _ = _compare(oracle32, cam, xyz, col, op, s, rot)
    assert R > 9360  # the big ones alone reach most tiles


def test_pair_capacity_overflow_is_reported_and_retried(oracle32):
    """The binning kernels are enqueued before the host has seen R.  With a pair buffer that is too small they
    must not write anything, the C ABI must answer FSGS_ERR_CAPACITY together with the needed R, and the Python
    shim's retry with a larger buffer must then produce the oracle's image."""
    from fsgs_amd import rasterizer

    cam = synth.make_camera(160, 128)
    xyz, col, op, scl, rot = synth.random_small_scene(400, cam, seed=5)
    key = (400, 160, 128)
    rasterizer._capacity[key] = 32  # far below the real R
    try:
        R, _ = _compare(oracle32, cam, xyz, col, op, scl, rot)
        assert R > 32
        assert rasterizer._capacity[key] >= R
    finally:
        rasterizer._capacity.pop(key, None)


def sweep_scene(seed):
    """The scene of one seed of the randomised sweep below (also scripts/soak_raster.py, scripts/dev/diag_pixel.py)."""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(5, 150)), int(rng.integers(5, 120))
    P = int(rng.choice([1, 2, 7, 64, 65, 300, 1500]))
    w2c = None
    if seed % 3 == 1:
        w2c = synth.pose_matrix(np.array([1.0, 0, 0, 0]) + 0.05 * rng.standard_normal(4), 0.05 * rng.standard_normal(3))
    cam = synth.make_camera(W, H, w2c=w2c)
    lo, hi = [(0.05, 0.8), (1.5, 6.0), (4.0, 40.0), (0.3, 120.0)][seed % 4]
    ch = (3, 6, 1)[seed % 3 if seed % 5 else 2]
    xyz, col, op, s, r = synth.random_small_scene(P, cam, seed=seed, zmin=0.25, zmax=2.0, scale_px=(lo, hi), channels=ch)
    if w2c is not None:  # random_small_scene places points in the camera frame: move them to the world
        xyz = (np.linalg.inv(w2c) @ np.concatenate([xyz, np.ones((P, 1))], 1).T).T[:, :3]
    s[rng.random(P) < 0.2, 0] *= 12.0      # needles
    op[rng.random(P) < 0.1] = 0.003        # < 1/255: can never contribute
    op[rng.random(P) < 0.1] = 1.0          # clamped to 0.99 in the blend
    xyz[rng.random(P) < 0.05, 2] = 0.2     # exactly on the near-plane cull (z <= 0.2 is culled)
    f = lambda a: np.ascontiguousarray(a, np.float32)
    return cam, f(xyz), f(col), f(op), f(s), f(r)


@pytest.mark.parametrize("seed", range(40))
def test_randomised_small_scenes_match_oracle(oracle32, seed):
    """A seeded sweep over what the fixed cases do not vary together: image sizes that are not multiples of the tile
    (down to less than one tile), cloud sizes from 1 up, footprints from sub-pixel to screen-filling, needle-shaped
    Gaussians, a posed raster camera, opacities at both extremes (below 1/255: never visible; 1.0: clamped to 0.99),
    points on and behind the near plane, 1, 3 or 6 channels."""
    # (the sweep also creates and drops a camera per case: the allocator hands the old matrices' addresses to the new
    # ones, which is how a pointer-keyed host cache of the camera matrices was caught serving stale values)
    cam, xyz, col, op, s, r = sweep_scene(seed)
    _compare(oracle32, cam, xyz, col, op, s, r, seed=seed, tag="sweep/%d" % seed)


def test_witnessed_outliers_of_c1_and_the_sweep_are_few_and_unsigned():
    """element by element an outlier only needs a witness, so a path that lost (or won) EVERY
    near-tie would still pass.  Over what the C1 poses and the 40-seed sweep above left in ATTRIBUTION_LOG (run in this
    process, in file order): the witnessed outliers are a vanishing fraction of each tensor's elements, and the sign of
    got - want over them is that of a fair coin (a biased exp / a one-sided threshold would pile them on one side)."""
    from tests.util import MAX_OUTLIER_FRACTION, dump_attribution_log

    recs = [r for r in ATTRIBUTION_LOG if r["tag"] and r["tag"].split("/")[0] in ("c1", "sweep")]
    if not recs:
        pytest.skip("runs after test_c1_init_scene_eight_poses / the randomised sweep in the same process")
    summary = {}
    for group in ("c1", "sweep"):
        for what in ("image", "depth", "means3D", "means2D", "colors", "opacities", "scales", "rotations"):
            rs = [r for r in recs if r["tag"].startswith(group + "/") and r["what"] == what]
            if not rs:
                continue
            out, size = sum(r["outliers"] for r in rs), sum(r["size"] for r in rs)
            summary["%s/%s" % (group, what)] = dict(cases=len(rs), outliers=out, elements=size, fraction=out / size,
                                                    pos=sum(r["pos"] for r in rs), neg=sum(r["neg"] for r in rs))
            if group == "c1":  # full-size tensors: the fraction itself (the sweep's tiny tensors are bounded per call)
                assert out <= MAX_OUTLIER_FRACTION * size, (group, what, out, size)
    pos, neg, z, share = assert_sign_balanced(recs, "C1 poses + 40-seed sweep")
    summary["sign_balance"] = dict(pos=pos, neg=neg, z=z, positive_share=share)
    print(summary)
    dump_attribution_log("r06_outlier_statistics", summary)


def test_unsupported_channel_count_is_an_error_not_a_wrong_image():
    from diff_gaussian_rasterization import GaussianRasterizer

    cam = synth.make_camera(48, 32)
    xyz, col, op, s, r = synth.random_small_scene(20, cam, seed=0, channels=4)
    T = lambda a: torch.tensor(np.asarray(a, np.float32), device=DEV)
    with pytest.raises(Exception):
        GaussianRasterizer(raster_settings=_settings(cam))(
            means3D=T(xyz), means2D=torch.zeros(20, 3, device=DEV), opacities=T(op).reshape(-1, 1),
            colors_precomp=T(col), scales=T(s), rotations=T(r))


def _scene_tensors(cam, P, seed, channels=3):
    xyz, col, op, s, r = synth.random_small_scene(P, cam, seed=seed, channels=channels)
    T = lambda a: torch.tensor(np.asarray(a, np.float32), device=DEV)
    return T(xyz), T(col), T(op).reshape(-1, 1), T(s), T(r)


def _fwd_bwd(rast, tens, dL, leaves=None):
    leaves = leaves or [t.detach().clone().requires_grad_(True) for t in tens]
    m3, c, o, s, r = leaves
    img, radii, depth = rast(means3D=m3, means2D=torch.zeros_like(m3), opacities=o, colors_precomp=c, scales=s, rotations=r)
    (img * dL).sum().backward()
    return img.detach(), [x.grad.clone() for x in leaves]


def test_interleaved_live_states_and_repeated_backward():
    """the mapping iteration keeps several forward states alive before any backward runs (2 views x 2 passes,
    train.py:236-265): forward A, forward B, backward B, backward A must equal the isolated runs; and a retained graph
    can be backpropagated twice (the saved state is read-only)."""
    from diff_gaussian_rasterization import GaussianRasterizer

    camA, camB = synth.make_camera(96, 64), synth.make_camera(50, 70, w2c=synth.pose_matrix((1, .02, -.01, .03), (.02, 0, .01)))
    rA, rB = GaussianRasterizer(raster_settings=_settings(camA)), GaussianRasterizer(raster_settings=_settings(camB))
    tA, tB = _scene_tensors(camA, 700, 1), _scene_tensors(camB, 300, 2)
    g = torch.Generator(device=DEV).manual_seed(0)
    dA = torch.rand((3, 64, 96), device=DEV, generator=g) - 0.5
    dB = torch.rand((3, 70, 50), device=DEV, generator=g) - 0.5
    imgA, gA = _fwd_bwd(rA, tA, dA)
    imgB, gB = _fwd_bwd(rB, tB, dB)
    lA = [t.detach().clone().requires_grad_(True) for t in tA]
    lB = [t.detach().clone().requires_grad_(True) for t in tB]
    iA, _, _ = rA(means3D=lA[0], means2D=torch.zeros_like(lA[0]), opacities=lA[2], colors_precomp=lA[1], scales=lA[3], rotations=lA[4])
    iB, _, _ = rB(means3D=lB[0], means2D=torch.zeros_like(lB[0]), opacities=lB[2], colors_precomp=lB[1], scales=lB[3], rotations=lB[4])
    (iB * dB).sum().backward(retain_graph=True)
    (iA * dA).sum().backward()
    assert torch.equal(iA.detach(), imgA) and torch.equal(iB.detach(), imgB)
    for got, want in zip([x.grad for x in lA] + [x.grad for x in lB], gA + gB):
        assert (got - want).abs().max() <= 1e-5 * want.abs().max() + 1e-12  # atomics order only
    first = [x.grad.clone() for x in lB]
    (iB * dB).sum().backward()  # second pass over the retained graph: gradients accumulate to exactly twice
    for x, f in zip(lB, first):
        assert (x.grad - 2 * f).abs().max() <= 2e-5 * f.abs().max() + 1e-12


def test_non_contiguous_inputs_and_a_side_stream():
    """views into larger tensors (the reference slices and transposes freely) and a call issued on a non-default
    stream give the results of the contiguous default-stream call; gradients come back in the views' shapes."""
    from diff_gaussian_rasterization import GaussianRasterizer

    cam = synth.make_camera(80, 48)
    rast = GaussianRasterizer(raster_settings=_settings(cam))
    tens = _scene_tensors(cam, 500, 3)
    dL = torch.rand((3, 48, 80), device=DEV) - 0.5
    img0, g0 = _fwd_bwd(rast, tens, dL)
    P = tens[0].shape[0]
    big = torch.zeros((P, 7), device=DEV)
    big[:, 1:4] = tens[0]
    m3 = big[:, 1:4].detach().requires_grad_(True)                      # row stride 7
    col = tens[1].t().contiguous().t().detach().requires_grad_(True)     # column-major
    sc = tens[3][:, [2, 0, 1]][:, [1, 2, 0]].detach().requires_grad_(True)
    rot = torch.cat([tens[4], tens[4]], 1)[:, :4].detach().requires_grad_(True)
    op = tens[2].expand(P, 1).detach().requires_grad_(True)
    assert not m3.is_contiguous() and not col.is_contiguous() and not rot.is_contiguous()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        img1, g1 = _fwd_bwd(rast, None, dL, leaves=[m3, col, op, sc, rot])
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(img1, img0)
    for a, b in zip(g1, g0):
        assert a.shape == b.shape and (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-12


def test_malformed_inputs_raise_before_any_kernel_runs():
    from diff_gaussian_rasterization import GaussianRasterizer

    cam = synth.make_camera(48, 32)
    rast = GaussianRasterizer(raster_settings=_settings(cam))
    m3, col, op, sc, rot = _scene_tensors(cam, 20, 0)
    call = lambda **kw: rast(**{**dict(means3D=m3, means2D=torch.zeros_like(m3), opacities=op, colors_precomp=col,
                                       scales=sc, rotations=rot), **kw})
    call()
    with pytest.raises(ValueError):
        call(rotations=rot[:, :3])          # a quaternion has four components
    with pytest.raises(ValueError):
        call(scales=sc[:10])                # fewer rows than Gaussians
    with pytest.raises(ValueError):
        call(opacities=op[:5])
    with pytest.raises(ValueError):
        call(colors_precomp=col.cpu())      # mixed devices
    with pytest.raises(Exception):
        call(colors_precomp=None)           # neither shs nor colours (upstream's own check)
    with pytest.raises(NotImplementedError):
        call(colors_precomp=None, shs=torch.zeros(20, 16, 3, device=DEV))
