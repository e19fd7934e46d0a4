"""The reference's own numbers fed to the FUSED HIP render (csrc/render.hip) directly, through the C ABI:
tests/golden/eval_sh.npz (utils/sh_utils.py:57-112 + the clamp of scene/gaussian_model.py:319-320),
depth_sil.npz (scene/gaussian_model.py:260-275, including the stored-matrix-row quirk) and pose_glue.npz
(transform_to_frame, scene/pose_optimizer.py:960-989) were written by tests/golden/make_golden.py from the imported
reference.  Here a cloud is BUILT FROM each fixture, rendered by fsgs_render_forward / fsgs_render_backward(_compact), and
what the kernels computed per Gaussian is read back (fsgs_render_state_layout: colours; xy / depth; the SH gradients
against the clamped colour gradient the compact backward reports) and compared with the fixture."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from fsgs_amd import _lib, rasterizer, synth
from fsgs_amd.render_ops import _args_struct
from fsgs_amd.trainer import settings_from_cam

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"
T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=DEV)


class Fused:
    """one fused forward through the C ABI, with typed views of its per-Gaussian state."""

    def __init__(self, cam, xyz, f_dc, f_rest, opacity, scaling, rotation, w2c, cam_center, deg, max_deg=3):
        lib = _lib.load()
        self.lib, self.P = lib, int(xyz.shape[0])
        self.t = [T(v) for v in (xyz, f_dc, f_rest, opacity, scaling, rotation, w2c, cam_center)]
        self.cfg = rasterizer.make_cfg(settings_from_cam(cam, DEV), 6)
        self.H, self.W = self.cfg.image_height, self.cfg.image_width
        P, H, W = self.P, self.H, self.W
        self.args = _args_struct(*self.t, deg, max_deg)
        self.image = torch.empty((3, H, W), device=DEV)
        self.depth_sil = torch.empty((3, H, W), device=DEV)
        self.radii = torch.empty((P,), dtype=torch.int32, device=DEV)
        self.cap = max(1 << 16, 64 * P)
        sb, xb = C.c_size_t(0), C.c_size_t(0)
        _lib.check(lib.fsgs_render_sizes(P, W, H, self.cap, C.byref(sb), C.byref(xb)), "sizes")
        self.sb = sb.value
        self.state = torch.zeros((sb.value,), dtype=torch.uint8, device=DEV)
        self.scratch = torch.zeros((max(xb.value, P * 64 + 512),), dtype=torch.uint8, device=DEV)
        nr = C.c_int64(0)
        _lib.check(lib.fsgs_render_forward(C.byref(self.cfg), P, C.byref(self.args), _lib.ptr(self.image),
                                           _lib.ptr(self.depth_sil), _lib.ptr(self.radii), _lib.ptr(self.state), sb.value,
                                           _lib.ptr(self.scratch), self.scratch.numel(), self.cap, C.byref(nr),
                                           _lib.current_stream()), "forward")
        self.nr = int(nr.value)
        off = (C.c_size_t * 9)()
        _lib.check(lib.fsgs_render_state_layout(P, W, H, self.cap, off), "layout")
        view = lambda i, n, shape: self.state[off[i]:off[i] + 4 * n].view(torch.float32).reshape(shape)
        # [7]: the packed records the blend kernels gather, floats 8..13 of each 64-byte row are the six colours
        self.xy, self.depth, self.colors = view(0, 2 * P, (P, 2)), view(2, P, (P,)), view(7, 16 * P, (P, 16))[:, 8:14]
        torch.cuda.synchronize()

    def backward(self, d_image, d_depth_sil=None):
        """-> (grads dict of the full backward, compact [P,14] gradient of the compact backward)."""
        P = self.P
        g = {k: torch.zeros_like(t) for k, t in zip(("xyz", "features_dc", "features_rest", "opacity", "scaling",
                                                    "rotation"), self.t[:6])}
        g["means2D"] = torch.zeros((P, 3), device=DEV)
        gs = _lib.FsgsRenderGrads()
        for k, v in g.items():
            setattr(gs, k, v.data_ptr())
        gs.w2c = None
        di = T(d_image)
        dd = None if d_depth_sil is None else T(d_depth_sil)
        common = (C.byref(self.cfg), P, C.byref(self.args), _lib.ptr(self.radii), _lib.ptr(self.state), self.sb, self.cap,
                  self.nr, _lib.ptr(di), _lib.ptr(dd))
        _lib.check(self.lib.fsgs_render_backward(*common, 1, 0, 1, C.byref(gs), _lib.ptr(self.scratch),
                                                 self.scratch.numel(), _lib.current_stream()), "backward")
        gc = torch.zeros((P, 14), device=DEV)
        m2 = torch.zeros((P, 3), device=DEV)
        _lib.check(self.lib.fsgs_render_backward_compact(*common, _lib.ptr(gc), _lib.ptr(m2), None, _lib.ptr(self.scratch),
                                                         self.scratch.numel(), _lib.current_stream()), "compact")
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in g.items()}, gc.cpu().numpy()


def _sh_cloud(deg):
    """A cloud whose SH coefficients and view directions are the fixture's: cam_center c0 in front of the raster
    camera, every Gaussian one unit away from it in its fixture direction (so all of them are in view)."""
    g = np.load(os.path.join(G, "eval_sh.npz"))
    sh, dirs = g["sh"], g["dirs"]                      # [P,3,16] channel-major, [P,3] unit
    P = sh.shape[0]
    c0 = np.array([0.0, 0.0, 5.0], np.float32)
    xyz = (c0 + dirs).astype(np.float32)
    f_dc = np.ascontiguousarray(sh[:, :, 0].reshape(P, 1, 3))
    f_rest = np.ascontiguousarray(sh[:, :, 1:].transpose(0, 2, 1))  # [P,15,3]
    rng = np.random.default_rng(0)
    opacity = rng.normal(0.5, 0.5, (P, 1)).astype(np.float32)
    scaling = np.log(rng.uniform(0.05, 0.12, (P, 3))).astype(np.float32)
    rotation = rng.normal(0, 1, (P, 4)).astype(np.float32)
    cam = synth.make_camera(192, 160)
    return g, Fused(cam, xyz, f_dc, f_rest, opacity, scaling, rotation, np.eye(4, dtype=np.float32), c0, deg), xyz, c0


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_fused_sh_colours_and_gradients_equal_the_reference_golden(deg):
    g, f, xyz, c0 = _sh_cloud(deg)
    P = f.P
    assert int((f.radii > 0).sum()) == P, "the fixture cloud must be entirely in view"
    # the directions the kernel sees are the fixture's up to the rounding of (c0 + d) - c0: 1 ulp of 5 = 5e-7
    want = g[f"rgb{deg}"]
    got = f.colors[:, :3].cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    assert ((got == 0) == (want == 0)).mean() > 0.995  # clamp_min(., 0) decided alike (a value within 1e-6 of 0 may differ)
    # depth / silhouette pseudo-colours of the same cloud: (z, 1, z^2) with the identity raster camera
    z = xyz[:, 2].astype(np.float64)
    np.testing.assert_allclose(f.colors[:, 3].cpu().numpy(), z, rtol=1e-6)
    assert bool((f.colors[:, 4] == 1).all())
    np.testing.assert_allclose(f.colors[:, 5].cpu().numpy(), z * z, rtol=2e-6)
    # backward: an arbitrary image gradient; the compact backward reports gcol = clamp-masked dL/dcolour per Gaussian,
    # and the SH gradient is linear in it:  dL/dsh[i,c,k] = basis_k(dir_i) gcol[i,c], so with the fixture's upstream w:
    #   dsh_hip[i,c,:] = dsh_golden[i,c,:] * gcol[i,c] / w[i,c]
    rng = np.random.default_rng(deg)
    dL = (rng.uniform(-1, 1, (3, f.H, f.W)) / (f.H * f.W)).astype(np.float32)
    grads, gc = f.backward(dL)
    gcol = gc[:, 3:6]
    dsh_hip = np.concatenate([grads["features_dc"], grads["features_rest"]], axis=1).transpose(0, 2, 1)  # [P,3,16]
    w = g["w"]
    ok = np.abs(w) > 1e-3
    ratio = np.where(ok, gcol / np.where(ok, w, 1.0), 0.0)
    expect = g[f"dsh{deg}"] * ratio[:, :, None]
    scale = np.abs(expect).max()
    assert scale > 0
    err = np.abs(dsh_hip - expect) * ok[:, :, None]
    assert err.max() <= 1e-4 * scale, (err.max() / scale)
    # a clamped channel has no gradient on either side; coefficients above the active degree have none either
    clamped = want == 0
    assert np.all(gcol[clamped] == 0) and np.all(dsh_hip[clamped] == 0)
    assert np.all(dsh_hip[:, :, (deg + 1) ** 2:] == 0)
    # the direction gradient of the fixture: reachable where the geometric part of dL/dxyz is known -- it is not
    # separable here, so it is covered end to end (tests/test_render_gpu.py) and by the fixture on the torch statement


def test_fused_depth_silhouette_colours_equal_the_reference_golden_with_the_stored_matrix_rows():
    """get_depth_and_silhouette multiplies by viewmatrix[0] AS STORED (scene/gaussian_model.py:266-267): the fixture's
    second case uses a non-identity raster camera, whose stored row 2 -- not the maths row -- gives z."""
    g = np.load(os.path.join(G, "depth_sil.npz"))
    pts = g["pts"].astype(np.float32)
    P = pts.shape[0]
    rng = np.random.default_rng(1)
    mk = lambda cam: Fused(cam, pts, rng.normal(0, 1, (P, 1, 3)), rng.normal(0, 0.1, (P, 15, 3)),
                           rng.normal(0, 1, (P, 1)), np.log(rng.uniform(0.02, 0.05, (P, 3))), rng.normal(0, 1, (P, 4)),
                           np.eye(4, dtype=np.float32), np.zeros(3, np.float32), 0)
    f = mk(synth.make_camera(160, 128))
    seen = (f.radii > 0).cpu().numpy()
    assert seen.sum() >= 20
    np.testing.assert_allclose(f.colors[:, 3:6].cpu().numpy()[seen], g["ds_identity"][seen], rtol=2e-6, atol=1e-6)
    # the posed raster camera of the fixture: viewmatrix_stored = inv(M)^T  ->  w2c = inv(M)
    Vt = g["viewmatrix_stored"].astype(np.float64)
    cam = synth.make_camera(160, 128, w2c=Vt.T)
    np.testing.assert_allclose(np.asarray(cam["viewmatrix"], np.float64).reshape(4, 4), Vt, atol=1e-6)
    f2 = mk(cam)
    seen2 = (f2.radii > 0).cpu().numpy()
    assert seen2.sum() >= 15
    np.testing.assert_allclose(f2.colors[:, 3:6].cpu().numpy()[seen2], g["ds_stored"][seen2], rtol=1e-5, atol=1e-5)


def test_fused_transform_to_frame_equals_the_reference_golden():
    """x_cam = (w2c [x;1])[:3] inside the fused preprocess, on the fixture's points and pose: what the kernel projected
    (state xy, depth) must be the projection of the fixture's transform_to_frame output."""
    g = np.load(os.path.join(G, "pose_glue.npz"))
    xyz, w2c, y = g["ttf_xyz"].astype(np.float32), g["ttf_w2c"].astype(np.float32), g["ttf_11"].astype(np.float64)
    P = xyz.shape[0]
    rng = np.random.default_rng(2)
    cam = synth.make_camera(160, 128)
    f = Fused(cam, xyz, rng.normal(0, 1, (P, 1, 3)), rng.normal(0, 0.1, (P, 15, 3)), rng.normal(0, 1, (P, 1)),
              np.log(rng.uniform(0.02, 0.05, (P, 3))), rng.normal(0, 1, (P, 4)), w2c, np.zeros(3, np.float32), 0)
    seen = (f.radii > 0).cpu().numpy()
    assert seen.sum() >= 8, int(seen.sum())
    np.testing.assert_allclose(f.depth.cpu().numpy()[seen], y[seen, 2], rtol=2e-6)
    PM = np.asarray(cam["projmatrix"], np.float64).reshape(4, 4)  # transposed storage: h = [y;1] @ PM
    h = np.concatenate([y, np.ones((P, 1))], 1) @ PM
    ndc = h[:, :2] / (h[:, 3:4] + 1e-7)
    px = ((ndc[:, 0] + 1) * 160 - 1) * 0.5
    py = ((ndc[:, 1] + 1) * 128 - 1) * 0.5
    got = f.xy.cpu().numpy()[seen]
    np.testing.assert_allclose(got[:, 0], px[seen], atol=3e-4, rtol=1e-5)
    np.testing.assert_allclose(got[:, 1], py[seen], atol=3e-4, rtol=1e-5)


def test_forward_reusing_the_colours_of_an_earlier_forward_is_bit_identical():
    """fsgs_render_forward_reuse_colors: same cloud, another pose -- the colours copied from the earlier forward's packed
    records are the ones a full evaluation gives (they do not depend on the pose), so images, records and the backward's
    inputs are identical bit for bit; a state of the wrong size is refused."""
    rng = np.random.default_rng(7)
    W, H, P = 160, 128, 3000
    cam = synth.make_camera(W, H)
    sc = synth.trained_like_scene(W, H, P, seed=4)
    P = sc["_xyz"].shape[0]
    pose_a = synth.pose_matrix((1.0, 0.01, -0.02, 0.015), (0.02, -0.01, 0.03)).astype(np.float32)
    pose_b = synth.pose_matrix((1.0, -0.02, 0.01, 0.0), (-0.01, 0.02, 0.01)).astype(np.float32)
    mk = lambda w2c: Fused(cam, sc["_xyz"], sc["_features_dc"], sc["_features_rest"], sc["_opacity"], sc["_scaling"],
                           sc["_rotation"], w2c, np.zeros(3, np.float32), 3)
    fa, fb = mk(pose_a), mk(pose_b)
    lib = fa.lib
    image, depth_sil = torch.empty_like(fb.image), torch.empty_like(fb.depth_sil)
    radii = torch.empty_like(fb.radii)
    state, scratch = torch.zeros_like(fb.state), torch.zeros_like(fb.scratch)
    nr = C.c_int64(0)
    call = lambda prev_bytes: lib.fsgs_render_forward_reuse_colors(
        C.byref(fb.cfg), P, C.byref(fb.args), _lib.ptr(image), _lib.ptr(depth_sil), _lib.ptr(radii), _lib.ptr(state), fb.sb,
        _lib.ptr(scratch), scratch.numel(), fb.cap, C.byref(nr), _lib.ptr(fa.state), prev_bytes, fa.cap, _lib.current_stream())
    _lib.check(call(fa.sb), "fsgs_render_forward_reuse_colors")
    torch.cuda.synchronize()
    assert nr.value == fb.nr and torch.equal(radii, fb.radii)
    assert torch.equal(image, fb.image) and torch.equal(depth_sil, fb.depth_sil)
    off = (C.c_size_t * 9)()
    _lib.check(lib.fsgs_render_state_layout(P, W, H, fb.cap, off), "layout")
    rec = lambda st: st[off[7]:off[7] + 64 * P].view(torch.float32)
    flg = lambda st: st[off[8]:off[8] + 4 * P].view(torch.int32)
    assert torch.equal(rec(state), rec(fb.state)) and torch.equal(flg(state), flg(fb.state))
    assert not torch.equal(rec(fa.state), rec(fb.state))  # (the two poses do project differently)
    assert call(fa.sb - 64) == _lib.FSGS_ERR_STATE


def test_pose_only_backward_without_a_means2D_holder_gives_the_same_pose_gradient():
    """fsgs_render_backward(gs_grad=0, cam_grad=1, param_grads=0): with grads.means2D = NULL (the tracking step) the
    per-Gaussian pass is the dedicated reduction kernel, with a holder it is the general one -- the twelve sums of
    dL/dw2c must agree (different summation trees: 1e-5 of the matrix's inf-norm), row 3 stays zero, and the call
    without a holder is still refused when any per-Gaussian gradient is asked for."""
    rng = np.random.default_rng(11)
    W, H, P = 320, 256, 6000
    cam = synth.make_camera(W, H)
    sc = synth.trained_like_scene(W, H, P, seed=5)
    P = sc["_xyz"].shape[0]
    w2c = synth.pose_matrix((1.0, 0.012, -0.02, 0.01), (0.015, -0.01, 0.02)).astype(np.float32)
    f = Fused(cam, sc["_xyz"], sc["_features_dc"], sc["_features_rest"], sc["_opacity"], sc["_scaling"], sc["_rotation"],
              w2c, np.zeros(3, np.float32), 2)
    di = T(rng.normal(0, 1e-3, (3, H, W)).astype(np.float32))
    common = (C.byref(f.cfg), P, C.byref(f.args), _lib.ptr(f.radii), _lib.ptr(f.state), f.sb, f.cap, f.nr, _lib.ptr(di), None)

    def run(with_holder, gs_grad=0, param_grads=0):
        gs = _lib.FsgsRenderGrads()
        m2 = torch.zeros((P, 3), device=DEV)
        dw = torch.full((4, 4), 7.0, device=DEV)  # must be overwritten, not accumulated into
        gs.means2D = m2.data_ptr() if with_holder else None
        gs.w2c = dw.data_ptr()
        rc = f.lib.fsgs_render_backward(*common, gs_grad, 1, param_grads, C.byref(gs), _lib.ptr(f.scratch), f.scratch.numel(),
                                        _lib.current_stream())
        torch.cuda.synchronize()
        return rc, dw.cpu().numpy().astype(np.float64), m2

    rc_a, a, m2 = run(True)
    rc_b, b, _ = run(False)
    assert rc_a == _lib.FSGS_OK and rc_b == _lib.FSGS_OK
    assert np.abs(a[:3]).max() > 0
    assert float(m2.abs().max()) == 0  # gs_grad = False: no screen-space gradient either way (as in the reference)
    assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max(), (a, b)
    assert np.all(a[3] == 0) and np.all(b[3] == 0)
    assert run(False, gs_grad=1)[0] == _lib.FSGS_ERR_INVALID


# axis permutations (proper rotations) as raster view matrices: together their upper-left 2x2 blocks of W Sigma W^T
# visit all six entries of Sigma; entry names index strip_symmetric's order (xx, xy, xz, yy, yz, zz)
_PERMS = {
    "identity": (np.eye(3), np.array([0.0, 0.0, 1.0]), (0, 1, 3)),                                   # xx xy yy
    "yzx": (np.array([[0.0, 1, 0], [0, 0, 1], [1, 0, 0]]), np.array([1.0, 0.0, 0.0]), (3, 4, 5)),    # yy yz zz
    "zxy": (np.array([[0.0, 0, 1], [1, 0, 0], [0, 1, 0]]), np.array([0.0, 1.0, 0.0]), (5, 2, 0)),    # zz zx xx
}


def _sigma_2x2_from_conic(A, B, Cc, fx, fy, z):
    """the HIP preprocess stores the conic = inverse of (J W Sigma W^T J^T + 0.3 I); on the optical axis
    J = diag(fx / z, fy / z), so the block of W Sigma W^T it projected comes back in closed form."""
    det = A * Cc - B * B
    a, b, c = Cc / det, -B / det, A / det
    return (a - 0.3) * z * z / (fx * fx), b * z * z / (fx * fy), (c - 0.3) * z * z / (fy * fy)


@pytest.mark.parametrize("perm", sorted(_PERMS))
@pytest.mark.parametrize("route", ["operator", "fused"])
def test_hip_preprocess_covariance_equals_the_reference_golden(perm, route):
    """tests/golden/covariance.npz = build_covariance_from_scaling_rotation of the imported reference
    (scene/gaussian_model.py:32-36), the in-tree statement of the rasteriser's Sigma = R S^2 R^T.  Each fixture
    Gaussian is put on the optical axis of an axis-permuting raster camera and the conic the HIP preprocess stored
    (operator boundary: fsgs_raster_state_layout; fused render: the packed records of fsgs_render_state_layout, fed the
    RAW scaling / rotation so that exp and normalize run in-kernel) is inverted back to the projected block of Sigma."""
    g = np.load(os.path.join(G, "covariance.npz"))
    Wv, axis, entries = _PERMS[perm]
    n = len(g["scaling"])
    W, H, f = 96, 80, 40.0
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
    w2c = np.eye(4)
    w2c[:3, :3] = Wv
    cam = synth.make_camera(W, H, w2c=w2c, K=K)
    z = np.linspace(0.6, 1.8, n)
    xyz = (axis[None, :] * z[:, None]).astype(np.float32)
    if route == "operator":
        cfg = rasterizer.make_cfg(settings_from_cam(cam, DEV), 3)
        _, _, radii, st = rasterizer.raster_forward(cfg, T(xyz), T(np.full((n, 3), 0.5)), T(np.full(n, 0.5)), T(g["scaling"]),
                                                    T(g["rotation_normalised"]))
        co = rasterizer.state_views(st)["conic_opacity"].cpu().numpy().astype(np.float64)
        A, B, Cc = co[:, 0], co[:, 1], co[:, 2]
    else:
        fz = Fused(cam, xyz, np.zeros((n, 1, 3), np.float32), np.zeros((n, 15, 3), np.float32), np.zeros((n, 1), np.float32),
                   np.log(g["scaling"]), g["rotation_raw"], np.eye(4, dtype=np.float32), np.zeros(3, np.float32), 0)
        radii = fz.radii
        off = (C.c_size_t * 9)()
        _lib.check(fz.lib.fsgs_render_state_layout(n, W, H, fz.cap, off), "layout")
        rec = fz.state[off[7]:off[7] + 64 * n].view(torch.float32).reshape(n, 16).cpu().numpy().astype(np.float64)
        A, B, Cc = rec[:, 2], rec[:, 3], rec[:, 4]
    assert int((radii > 0).sum()) == n
    got = _sigma_2x2_from_conic(A, B, Cc, f, f, z)
    want = g["cov6"].astype(np.float64)
    scale = np.abs(want).max(axis=1)
    for v, e in zip(got, entries):
        assert np.max(np.abs(v - want[:, e]) / scale) <= 2e-5, (perm, route, e, np.max(np.abs(v - want[:, e]) / scale))


def test_rasteriser_consumes_the_settings_the_reference_builds(oracle32):
    """The 12 fields PoseModel.setup_camera handed to GaussianRasterizationSettings in the imported reference
    (tests/golden/camera.npz, a POSED camera at C1 size) go into the drop-in as they are -- shapes [1,4,4], transposed
    storage -- and the result is the oracle's on the same matrices."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    from tests.util import assert_close_attributed

    g = np.load(os.path.join(G, "camera.npz"))
    f = lambda k: g["c1_posed_" + k]
    W, H = int(f("image_width")), int(f("image_height"))
    s = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=float(f("tanfovx")), tanfovy=float(f("tanfovy")), bg=T(f("bg")),
        scale_modifier=float(f("scale_modifier")), viewmatrix=T(f("viewmatrix")), projmatrix=T(f("projmatrix")),
        sh_degree=int(f("sh_degree")), campos=T(f("campos")), prefiltered=bool(f("prefiltered")), debug=bool(f("debug")))
    assert tuple(s.viewmatrix.shape) == (1, 4, 4)
    cam = dict(image_height=H, image_width=W, tanfovx=float(f("tanfovx")), tanfovy=float(f("tanfovy")), bg=f("bg"),
               viewmatrix=f("viewmatrix").reshape(4, 4), projmatrix=f("projmatrix").reshape(4, 4), K=f("K"))
    P = 4000
    xyz, col, op, sc, rot = synth.random_small_scene(P, cam, seed=3)
    w2c = f("w2c").astype(np.float64)   # random_small_scene places points in the camera frame: move them to the world
    xyz = (np.linalg.inv(w2c) @ np.concatenate([xyz, np.ones((P, 1))], 1).T).T[:, :3]
    a32 = lambda a: np.ascontiguousarray(a, np.float32)
    xyz, col, op, sc, rot = a32(xyz), a32(col), a32(op), a32(sc), a32(rot)
    m3 = T(xyz).requires_grad_(True)
    img, radii, depth = GaussianRasterizer(raster_settings=s)(means3D=m3, means2D=torch.zeros_like(m3), opacities=T(op).reshape(-1, 1),
                                                              colors_precomp=T(col), scales=T(sc), rotations=T(rot))
    dL = (np.random.default_rng(0).uniform(-1, 1, (3, H, W)) / (3 * H * W)).astype(np.float32)
    (img * T(dL)).sum().backward()
    amp, (oi, od, orad, og, ost) = oracle32.flip_amplitudes(cam, xyz, col, op, sc, rot, dL)
    assert int(((radii.cpu().numpy() > 0) != (orad > 0)).sum()) == 0 and int((orad > 0).sum()) > P // 2
    assert_close_attributed(img.detach().cpu().numpy(), oi, amp["image"], "image", floor=1.0)
    floor = 1e-3 * max(float(np.abs(v).max()) for v in og.values())
    assert_close_attributed(m3.grad.cpu().numpy(), og["means3D"], amp["means3D"], "means3D", floor=floor)
