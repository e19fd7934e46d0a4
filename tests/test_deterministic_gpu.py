"""FSGS_FLAG_DETERMINISTIC (include/fsgs.h; SURVEY.md s7 "deterministic mode for tests"): the backward without float atomics.
Every (tile, Gaussian) pair's totals go to a row of their own and are summed per Gaussian in a fixed order
(csrc/raster_kernels.h det_gather_kernel), dL/dw2c is summed over per-workgroup partials in workgroup order
(csrc/render.hip w2c_finish_kernel).  Two runs must then agree BIT FOR BIT -- at the operator boundary, through the fused
render under autograd, and over whole steps of the step driver (mapping with Adam fused in, two-view mapping, tracking) --
and agree with the product path up to the order of float additions."""
import numpy as np
import pytest
import torch

from fsgs_amd import rasterizer, synth
from tests.util import sh0_colors, to_camera_frame

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def deterministic():
    prev = rasterizer.set_deterministic(True)
    try:
        yield
    finally:
        rasterizer.set_deterministic(prev)


def _close(a, b, what, tol=2e-5):
    floor = 1e-3 * max([float(np.abs(v).max()) for v in a.values() if v is not None] or [0.0])
    for k in a:
        if a[k] is None:
            assert b[k] is None, k
            continue
        scale = max(float(np.abs(a[k]).max()), floor)
        err = float(np.abs(a[k] - b[k]).max())
        assert err <= tol * scale, "%s %s: %.3g of %.3g" % (what, k, err, scale)


@pytest.mark.parametrize("W,H,P,kind", [(320, 256, 6000, "trained"), (640, 512, 20000, "init"), (1280, 1024, 300_000, "trained")])
def test_operator_boundary_backward_is_bit_reproducible(W, H, P, kind):
    from tests.test_raster_gpu import _run_hip

    cam = synth.make_camera(W, H)
    if kind == "init":
        sc = synth.init_scene(W, H, P, seed=0)
    else:
        from simple_knn._C import distCUDA2

        sc = synth.trained_like_scene(W, H, P, seed=3, knn_fn=lambda pts: distCUDA2(torch.tensor(pts, device=DEV)).cpu().numpy())
    s, r, o = synth.activate(sc)
    xyz = to_camera_frame(sc["_xyz"], synth.pose_matrix(**synth.PERTURBED_POSE))
    col = sh0_colors(sc)
    dL = (np.random.default_rng(1).uniform(-1, 1, (3, H, W)) / (3 * H * W)).astype(np.float32)
    product = _run_hip(cam, xyz, col, o.reshape(-1), s, r, dL)
    prev = rasterizer.set_deterministic(True)
    try:
        runs = [_run_hip(cam, xyz, col, o.reshape(-1), s, r, dL) for _ in range(3)]
    finally:
        rasterizer.set_deterministic(prev)
    for other in runs[1:]:
        for k, v in runs[0][3].items():
            assert np.array_equal(v, other[3][k]), "%s differs between two deterministic runs" % k
    assert np.array_equal(product[0], runs[0][0]) and np.array_equal(product[1], runs[0][1])  # (the forward is the same code)
    _close(product[3], runs[0][3], "deterministic vs product, %dx%d" % (W, H))


@pytest.mark.parametrize("mode", [(True, False), (False, True), (True, True)])
def test_fused_render_backward_is_bit_reproducible(mode):
    from fsgs_amd.render import render
    from tests.test_render_gpu import _run, _setup

    gs_grad, cam_grad = mode
    W, H, P = 640, 512, 20000
    pc, poses = _setup(W, H, P, 3, seed=4)
    g = torch.Generator(device="cpu").manual_seed(5)
    wi = (torch.rand(3, H, W, generator=g) - 0.5).to(DEV) / (H * W)
    wd = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    ws = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    product = _run(render, pc, poses, gs_grad, cam_grad, wi, wd, ws)
    prev = rasterizer.set_deterministic(True)
    try:
        runs = [_run(render, pc, poses, gs_grad, cam_grad, wi, wd, ws) for _ in range(3)]
    finally:
        rasterizer.set_deterministic(prev)
    for other in runs[1:]:
        for k, v in runs[0][1].items():
            assert (v is None and other[1][k] is None) or np.array_equal(v, other[1][k]), k
    # the pose gradient is a P-term sum: workgroup-ordered partials against atomics in arrival order
    _close({k: v for k, v in product[1].items() if k not in ("r", "t")}, {k: v for k, v in runs[0][1].items() if k not in ("r", "t")},
           "deterministic vs product")
    if cam_grad:
        _close({k: product[1][k] for k in ("r", "t")}, {k: runs[0][1][k] for k in ("r", "t")}, "pose gradient", tol=2e-4)


def test_step_driver_runs_are_bit_identical(deterministic):
    """FastStepper under the deterministic flag: mapping (Adam fused into the backward), two-view mapping (two compact
    gradients, Adam from their sum) and tracking (pose-only backward, dL/dw2c partials, pose Adam) -- the whole state after
    the same eight steps, twice"""
    from fsgs_amd import losses
    from fsgs_amd.fast_step import FastStepper
    from fsgs_amd.flow import FlowTargets
    from fsgs_amd.model import PARAM_NAMES
    from tests.test_fast_step_gpu import _world

    W, H = 320, 256
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)

    def run():
        pc, poses, frames, cam = _world(seed=2, W=W, H=H, P=6000)
        fs = FastStepper(pc, poses, frames)
        assert fs._cfg().flags & 128
        out = [float(fs.mapping_step([1], corners=corners)) for _ in range(3)]
        out += [float(fs.mapping_step([2, 1], corners=corners)) for _ in range(2)]
        poses.initialize_tracking_optimizer(50)
        rigid = torch.ones(H, W, dtype=torch.bool, device=DEV)
        tg = FlowTargets(torch.ones(1, H, W, device=DEV), np.eye(4, dtype=np.float32), frames.K, frames.flows_fw[0], rigid)
        out += [float(fs.tracking_step(1, tg, rigid)[0]) for _ in range(3)]
        torch.cuda.synchronize()
        state = {k: pc.params[k].detach().clone() for k in PARAM_NAMES}
        for gidx, grp in enumerate(pc.optimizer.param_groups):
            st = pc.optimizer.state[grp["params"][0]]
            state["m%d" % gidx], state["v%d" % gidx] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
        for k in ("max_radii2D", "xyz_gradient_accum", "denom"):
            state[k] = pc.variables[k].detach().clone()
        state["r"], state["t"] = poses.r.detach().clone(), poses.t.detach().clone()
        return out, state

    a, b = run(), run()
    assert a[0] == b[0], (a[0], b[0])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k


def test_deterministic_backward_refuses_a_scratch_without_room_for_the_pair_rows(deterministic):
    """the flag changes what `scratch` must hold: the accumulator rows alone are FSGS_ERR_CAPACITY, not a silent overrun"""
    import ctypes as C

    from fsgs_amd import _lib, render_ops
    from fsgs_amd.render import render
    from tests.test_render_gpu import _setup

    pc, poses = _setup(160, 128, 2000, 1, seed=1)
    lib = _lib.load()
    need = int(lib.fsgs_deterministic_scratch_bytes(2000, 1 << 16))
    assert need >= 2000 * 64 + (1 << 16) * 64
    orig = rasterizer.backward_scratch_bytes
    rasterizer.backward_scratch_bytes = lambda cfg, P, cap, rows: rows
    try:
        pkg = render(poses, 1, pc, gs_grad=True, cam_grad=False)
        with pytest.raises(_lib.FsgsError) as e:
            pkg["render"].sum().backward()
        assert e.value.code == _lib.FSGS_ERR_CAPACITY
    finally:
        rasterizer.backward_scratch_bytes = orig
    assert render_ops is not None and C is not None
