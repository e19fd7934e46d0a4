"""The two flavours of the blend kernels (csrc/raster_kernels.h: one wave per 16x16 tile / four waves per tile, one 8x8
quadrant each) against EACH OTHER, on the same inputs, through the same entry points (include/fsgs.h
FSGS_FLAG_BLEND_ONE_WAVE / FSGS_FLAG_BLEND_QUAD_WAVES).  Both walk a pixel's Gaussians in the same order through the same
per-pixel step (blend_fwd_pixel / blend_bwd_pixel), so

  * every forward output is BIT-identical (image, depth planes, final T / last contributor through the backward);
  * the backward's per-Gaussian sums differ only in the order of float additions (64-lane reductions + atomics per
    quadrant instead of four quadrants summed in registers first): held to 5e-5 of each tensor's inf-norm.

Each flavour alone is checked against the CPU oracle by tests/test_raster_gpu.py, test_render_gpu.py and
test_render_golden_gpu.py (run under both: tests/conftest.py); this file adds BASELINE.json's full sizes and the step
driver's kernels (RGB+depth-only forward, pose-only backward, Adam fused into the backward).
Match: gaussian_renderer/__init__.py:68-69 (R6 / R7 of SURVEY.md Appendix A)."""
import numpy as np
import pytest
import torch

from fsgs_amd import rasterizer, synth
from tests.util import sh0_colors, to_camera_frame

pytestmark = pytest.mark.gpu
DEV = "cuda"
# of the tensor's inf-norm (floored at 1e-3 of the largest gradient tensor's): summation order only -- half of SURVEY s8d's 1e-4.
# Measured <= 2.2e-5 (the rotation gradient at C4: a difference of near-cancelling terms, 3.6e-6 at its largest; everything
# else <= 1e-5); each flavour's own arrival order of the atomics moves it by as much from run to run.
GRAD_TOL = 5e-5


def _both(fn):
    out = {}
    for name in ("one", "quad"):
        prev = rasterizer.set_blend_variant(name)
        try:
            out[name] = fn()
        finally:
            rasterizer.set_blend_variant(prev)
    return out["one"], out["quad"]


def _assert_grads_close(a, b, what):
    floor = 1e-3 * max(float(np.abs(v).max()) for v in a.values() if v is not None)
    for k in a:
        if a[k] is None:
            assert b[k] is None, k
            continue
        scale = max(float(np.abs(a[k]).max()), floor)
        err = float(np.abs(a[k] - b[k]).max())
        assert err <= GRAD_TOL * scale, "%s %s: %.3g of %.3g" % (what, k, err, scale)


@pytest.mark.parametrize("W,H,P,kind", [(100, 70, 1500, "trained"), (640, 512, 20000, "init"), (333, 217, 9000, "trained"),
                                        (1280, 1024, 300_000, "trained"), (1920, 1080, 1_000_000, "trained")])
def test_operator_boundary_forward_is_bit_identical_and_gradients_agree(W, H, P, kind):
    from tests.test_raster_gpu import _run_hip

    cam = synth.make_camera(W, H)
    if kind == "init":
        sc = synth.init_scene(W, H, P, seed=0)
    else:
        from simple_knn._C import distCUDA2

        knn = lambda pts: distCUDA2(torch.tensor(pts, device=DEV)).cpu().numpy()
        sc = synth.trained_like_scene(W, H, P, seed=3, knn_fn=knn)
    s, r, o = synth.activate(sc)
    xyz = to_camera_frame(sc["_xyz"], synth.pose_matrix(**synth.PERTURBED_POSE))
    col = sh0_colors(sc)
    dL = (np.random.default_rng(1).uniform(-1, 1, (3, H, W)) / (3 * H * W)).astype(np.float32)
    one, quad = _both(lambda: _run_hip(cam, xyz, col, o.reshape(-1), s, r, dL))
    assert np.array_equal(one[0], quad[0]), "image"
    assert np.array_equal(one[1], quad[1]), "depth"
    assert np.array_equal(one[2], quad[2]), "radii"
    _assert_grads_close(one[3], quad[3], "%dx%d" % (W, H))


@pytest.mark.parametrize("W,H,P", [(320, 256, 6000), (640, 512, 20000), (1280, 1024, 300_000)])
@pytest.mark.parametrize("mode", [(True, False), (False, True), (True, True)])
def test_fused_render_forward_is_bit_identical_and_gradients_agree(W, H, P, mode):
    from fsgs_amd.render import render
    from tests.test_render_gpu import _run, _setup

    gs_grad, cam_grad = mode
    pc, poses = _setup(W, H, P, 3, seed=4)
    g = torch.Generator(device="cpu").manual_seed(5)
    wi = (torch.rand(3, H, W, generator=g) - 0.5).to(DEV) / (H * W)
    wd = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    ws = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    one, quad = _both(lambda: _run(render, pc, poses, gs_grad, cam_grad, wi, wd, ws))
    for k in one[0]:
        assert np.array_equal(one[0][k], quad[0][k]), k
    _assert_grads_close(one[1], quad[1], "%dx%d gs=%d cam=%d" % (W, H, gs_grad, cam_grad))


@pytest.mark.parametrize("W,H,P", [(320, 256, 6000), (640, 512, 20000)])
def test_step_driver_kernels_agree_between_the_flavours(W, H, P):
    """FastStepper: the mapping step (6-plane forward, DEPTH_GRAD_ONLY split backward with Adam fused in) and the
    tracking step (RGB+depth-only forward, pose-only backward): after three steps of each the parameters, the Adam
    state, the densification statistics and the pose agree (Adam normalises the update: compare with an absolute bar of
    a fraction of the learning-rate-sized step, and the statistics tightly)."""
    from fsgs_amd import losses
    from fsgs_amd.fast_step import FastStepper
    from fsgs_amd.flow import FlowTargets
    from fsgs_amd.model import PARAM_NAMES
    from tests.test_fast_step_gpu import _world

    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)

    def run():
        pc, poses, frames, cam = _world(seed=2, W=W, H=H, P=P)
        fs = FastStepper(pc, poses, frames)
        ls = [fs.mapping_step([1], corners=corners)]
        first_image = fs.last["image"].detach().clone()  # forward of the FIRST step: nothing has been updated yet
        ls += [fs.mapping_step([1], corners=corners) for _ in range(2)]
        poses.initialize_tracking_optimizer(50)
        rigid = torch.ones(H, W, dtype=torch.bool, device=DEV)
        tg = FlowTargets((torch.ones(1, H, W, device=DEV)), np.eye(4, dtype=np.float32), frames.K, frames.flows_fw[0], rigid)
        lt = [fs.tracking_step(1, tg, rigid)[0] for _ in range(3)]
        torch.cuda.synchronize()
        return (pc, poses, [float(x) for x in ls], [float(x) for x in lt], first_image)

    one, quad = _both(run)
    assert torch.equal(one[4], quad[4]), "image of the first mapping step's forward"
    for a, b in zip(one[2] + one[3], quad[2] + quad[3]):
        assert abs(a - b) <= 2e-5 * abs(a), (one[2], quad[2], one[3], quad[3])
    for k in PARAM_NAMES:
        pa, pb = one[0].params[k].detach(), quad[0].params[k].detach()
        diff = (pa - pb).abs()
        assert (diff > 1e-5 * pa.abs().max()).float().mean().item() < 1e-3, k
    for k in ("max_radii2D", "denom"):
        assert torch.equal(one[0].variables[k], quad[0].variables[k]), k
    assert torch.allclose(one[0].variables["xyz_gradient_accum"], quad[0].variables["xyz_gradient_accum"], rtol=1e-3, atol=1e-9)
    assert torch.allclose(one[1].r.detach(), quad[1].r.detach(), rtol=0, atol=1e-4)
    assert torch.allclose(one[1].t.detach(), quad[1].t.detach(), rtol=0, atol=1e-4)


def test_flavours_agree_on_the_harness_scene_at_the_tight_tolerance():
    """the pinned harness (tests/test_harness_pin_gpu.py) holds the four-waves backward only to the widened product
    bounds; here the SAME scene (tests/golden/harness_pin.npz: 256x192, 983 Gaussians, SH degree as recorded) goes through both
    flavours of the fused render directly -- forward bit-identical, every gradient within GRAD_TOL of its inf-norm -- so the
    quad backward itself is pinned on that scene, not only its noise."""
    import os

    from fsgs_amd.render import render
    from tests import ref_harness
    from tests.test_render_gpu import _run

    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness_pin.npz")))
    pc, poses, frames = ref_harness.load_inputs(fx, DEV)
    H, W = frames.colors[0].shape[-2:]
    poses.set_pose(1, synth.PERTURBED_POSE["q"], synth.PERTURBED_POSE["t"])
    g = torch.Generator(device="cpu").manual_seed(9)
    wi = (torch.rand(3, H, W, generator=g) - 0.5).to(DEV) / (H * W)
    wd = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    ws = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    for gs_grad, cam_grad in ((True, False), (False, True), (True, True)):
        one, quad = _both(lambda: _run(render, pc, poses, gs_grad, cam_grad, wi, wd, ws))
        for k in one[0]:
            assert np.array_equal(one[0][k], quad[0][k]), k
        _assert_grads_close(one[1], quad[1], "harness scene gs=%d cam=%d" % (gs_grad, cam_grad))


def test_flavour_flags_reach_the_configuration_struct_and_bad_names_are_refused():
    """rasterizer.set_blend_variant -> FsgsRasterCfg.flags (include/fsgs.h: ONE_WAVE 32, QUAD_WAVES 64; neither = the
    library's own choice: forward four waves per tile, backward by the size of the tile grid)"""
    from fsgs_amd import _lib
    from tests.test_raster_gpu import _settings

    assert _lib.FSGS_FLAG_BLEND_ONE_WAVE == 32 and _lib.FSGS_FLAG_BLEND_QUAD_WAVES == 64
    assert rasterizer.blend_variant() == "auto"
    cam = synth.make_camera(64, 48)
    assert rasterizer.make_cfg(_settings(cam), 3).flags & (32 | 64) == 0
    for name, bit in (("quad", 64), ("one", 32)):
        prev = rasterizer.set_blend_variant(name)
        try:
            assert rasterizer.make_cfg(_settings(cam), 3).flags & (32 | 64) == bit
        finally:
            rasterizer.set_blend_variant(prev)
    with pytest.raises(ValueError):
        rasterizer.set_blend_variant("two")
