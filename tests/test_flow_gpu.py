"""GPU parity of the flow reprojection loss kernels (csrc/flow.hip) against the golden vectors captured
from the reference's projection_flow_loss and against the torch statement at C1 size."""
import os

import numpy as np
import pytest
import torch

from fsgs_amd import flow, pose

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"
T = lambda a: torch.tensor(np.asarray(a), device=DEV)


def test_flow_loss_matches_reference_golden():
    g = np.load(os.path.join(G, "flow_loss.npz"))
    for tag, rm in (("rigid", T(g["rigid"])), ("norigid", None)):
        r = T(g["q"]).reshape(1, 4, 1).clone().requires_grad_(True)
        t = T(g["t"]).reshape(3, 1).clone().requires_grad_(True)
        w2c = pose.pose_to_w2c(r, t, 0)
        w2c.retain_grad()
        l = flow.projection_flow_loss(T(g["depth_prev"]), g["w2c_prev"], w2c, g["K"], T(g["flow"])[0], rm)
        l.backward()
        np.testing.assert_allclose(l.item(), g[f"loss_{tag}"], rtol=2e-5)
        np.testing.assert_allclose(w2c.grad.cpu().numpy()[:3], g[f"dw2c_{tag}"][:3], rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(r.grad.cpu().numpy(), g[f"dr_{tag}"], rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(t.grad.cpu().numpy(), g[f"dt_{tag}"], rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("H,W", [(512, 640), (1024, 1280), (1080, 1920)])
def test_flow_loss_matches_torch_statement_at_c1_c2_c4_and_caches_targets(H, W):
    """projection_flow_loss (scene/pose_optimizer.py:164-218) at the image sizes of BASELINE.json's configurations: up to
    2.07 M back-projected points, their duplicate rejection, the fused loss + pose gradient."""
    torch.manual_seed(0)
    from fsgs_amd import synth

    K = synth.intrinsics(W, H)
    u = torch.arange(W, device=DEV).float()[None] / W
    v = torch.arange(H, device=DEV).float()[:, None] / H
    depth = (1.0 + 0.3 * torch.sin(6.28 * u) * torch.cos(6.28 * v)).reshape(1, H, W).contiguous()
    depth[0, :9, :11] = 0.0
    fl = torch.randn(2, H, W, device=DEV)
    rigid = torch.rand(H, W, device=DEV) > 0.1
    w_prev = synth.pose_matrix((1, 0.001, 0.002, -0.001), (0.01, 0.0, -0.01)).astype(np.float32)
    targets = flow.FlowTargets(depth, w_prev, K, fl, rigid)
    for it in range(3):  # the per-frame targets are reused over the tracking iterations
        q = torch.tensor([1.0, 0.01 * it, -0.02, 0.015], device=DEV).reshape(1, 4, 1).requires_grad_(True)
        t = torch.tensor([0.02, -0.01, 0.03 * it], device=DEV).reshape(3, 1).requires_grad_(True)
        a = pose.pose_to_w2c(q, t, 0)
        a.retain_grad()
        la = flow.flow_pose_loss(a, targets)
        (2.0 * la).backward()
        ga = a.grad.clone()
        b = a.detach().clone().requires_grad_(True)
        lb = flow.projection_flow_loss_torch(depth, w_prev, b, K, fl, rigid)
        (2.0 * lb).backward()
        assert abs(la.item() - lb.item()) <= 2e-5 * abs(lb.item())
        scale = b.grad[:3].abs().max().item()
        assert (ga[:3] - b.grad[:3]).abs().max().item() <= 1e-3 * scale


def test_flow_loss_empty_and_behind_camera_are_zero():
    H, W = 64, 80
    from fsgs_amd import synth

    K = synth.intrinsics(W, H)
    depth = torch.zeros(1, H, W, device=DEV)
    fl = torch.zeros(2, H, W, device=DEV)
    w = torch.eye(4, device=DEV).requires_grad_(True)
    l = flow.projection_flow_loss(depth, np.eye(4, dtype=np.float32), w, K, fl, None)
    l.backward()
    assert l.item() == 0.0 and not bool(w.grad.abs().sum())
    depth = torch.ones(1, H, W, device=DEV)
    flip = torch.diag(torch.tensor([1.0, 1.0, -1.0, 1.0], device=DEV)).requires_grad_(True)  # everything behind
    l = flow.projection_flow_loss(depth, np.eye(4, dtype=np.float32), flip, K, fl, None)
    assert l.item() == 0.0


def test_pose_kernels_match_reference_golden():
    """LearnPose.forward + gradients (csrc/pose.hip) against the vectors captured from the reference."""
    from fsgs_amd.pose import pose_to_w2c_hip

    g = np.load(os.path.join(G, "pose_glue.npz"))
    for cam in range(g["r"].shape[2]):
        r = T(g["r"]).requires_grad_(True)
        t = T(g["t"]).requires_grad_(True)
        w2c = pose_to_w2c_hip(r, t, cam)
        (w2c * T(g["wsum"])).sum().backward()
        np.testing.assert_allclose(w2c.detach().cpu().numpy(), g[f"w2c_{cam}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r.grad.cpu().numpy(), g[f"dr_{cam}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(t.grad.cpu().numpy(), g[f"dt_{cam}"], rtol=1e-5, atol=1e-7)


def test_fused_flow_pass_matches_the_reference_golden_and_accumulates():
    """fsgs_flow_pose_loss_fused (one pass: loss + dL/dw2c, then dw2c = accumulate * dw2c + upstream * grad)
    against the reference's captured loss / gradient, and its accumulate semantics."""
    from fsgs_amd import _lib

    lib = _lib.load()
    g = np.load(os.path.join(G, "flow_loss.npz"))
    for tag, rm in (("rigid", T(g["rigid"])), ("norigid", None)):
        w2c = pose.pose_to_w2c(T(g["q"]).reshape(1, 4, 1), T(g["t"]).reshape(3, 1), 0).detach().contiguous()
        tg = flow.FlowTargets(T(g["depth_prev"]), g["w2c_prev"], g["K"], T(g["flow"])[0], rm)
        M = int(tg.pts.shape[0])
        scratch = torch.empty((int(lib.fsgs_flow_scratch_bytes(M)),), dtype=torch.uint8, device=DEV)
        out = torch.empty((2,), device=DEV)
        base = torch.arange(16, dtype=torch.float32, device=DEV).reshape(4, 4) * 0.01
        for up, acc in ((1.0, 0.0), (0.1, 1.0), (2.0, -0.5)):
            dw = base.clone() if acc != 0.0 else torch.full((4, 4), float("nan"), device=DEV)  # acc = 0 never reads
            with torch.cuda.device(DEV):
                _lib.check(lib.fsgs_flow_pose_loss_fused(M, _lib.ptr(tg.pts), _lib.ptr(tg.vu), _lib.ptr(w2c), tg.K9,
                                                         _lib.ptr(tg.flow), tg.W, tg.H, 20.0, up, acc, _lib.ptr(scratch),
                                                         _lib.ptr(out), _lib.ptr(dw), _lib.current_stream()),
                           "fsgs_flow_pose_loss_fused")
            np.testing.assert_allclose(out[0].item(), g[f"loss_{tag}"], rtol=2e-5)
            want = up * g[f"dw2c_{tag}"]
            want[3] = 0.0
            if acc != 0.0:
                want = want + acc * base.cpu().numpy()
            np.testing.assert_allclose(dw.cpu().numpy(), want, rtol=2e-3, atol=2e-4 * max(1.0, abs(up)))
    # nothing valid: loss 0, gradient 0 (scene/pose_optimizer.py:196-214)
    dw = torch.ones((4, 4), device=DEV)
    with torch.cuda.device(DEV):
        _lib.check(lib.fsgs_flow_pose_loss_fused(0, None, None, _lib.ptr(w2c), tg.K9, _lib.ptr(tg.flow), tg.W, tg.H, 20.0,
                                                 1.0, 0.0, _lib.ptr(scratch), _lib.ptr(out), _lib.ptr(dw),
                                                 _lib.current_stream()), "fsgs_flow_pose_loss_fused")
    assert out[0].item() == 0.0 and float(dw.abs().max()) == 0.0


def _two_view_flow(H, W, K, w2c_1, w2c_2, depth, moving=None):
    """forward flow of a static scene seen from two poses (optionally with an independently moving block)."""
    Kt = T(np.asarray(K, np.float32))
    v, u = torch.meshgrid(torch.arange(H, device=DEV).float(), torch.arange(W, device=DEV).float(), indexing="ij")
    z = depth
    cam1 = torch.stack([(u - Kt[0, 2]) / Kt[0, 0] * z, (v - Kt[1, 2]) / Kt[1, 1] * z, z, torch.ones_like(z)], 0).reshape(4, -1)
    A, B = T(np.asarray(w2c_1, np.float32)), T(np.asarray(w2c_2, np.float32))
    world = torch.linalg.inv(A) @ cam1
    cam2 = (B @ world)[:3]
    p = Kt @ cam2
    fl = torch.stack([p[0] / p[2] - u.reshape(-1), p[1] / p[2] - v.reshape(-1)], 0).reshape(2, H, W)
    if moving is not None:
        y0, y1, x0, x1, du, dv = moving
        fl[0, y0:y1, x0:x1] += du
        fl[1, y0:y1, x0:x1] += dv
    return fl.contiguous()


def test_sampson_rigid_mask_matches_torch_statement_and_flags_the_moving_block():
    from fsgs_amd import epipolar, synth

    H, W = 256, 320
    K = synth.intrinsics(W, H)
    w1 = synth.pose_matrix((1, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    w2 = synth.pose_matrix((1, 0.004, -0.006, 0.002), (0.02, -0.01, 0.004))
    u = torch.arange(W, device=DEV).float()[None] / W
    v = torch.arange(H, device=DEV).float()[:, None] / H
    depth = 1.0 + 0.3 * torch.sin(6.28 * u) * torch.cos(6.28 * v)
    fl = _two_view_flow(H, W, K, w1, w2, depth, moving=(100, 140, 180, 240, 6.0, -4.0))
    fl = fl + 0.02 * torch.randn_like(fl)  # flow-estimator noise
    F = epipolar.fundamental_from_w2c(w1, w2, K)
    rigid, dist, stats = epipolar.rigid_mask(fl, F, 2.0)
    want = epipolar.sampson_distance_torch(fl, F)
    scale = want.abs().max().item()
    assert (dist - want).abs().max().item() <= 2e-4 * scale
    np.testing.assert_allclose(stats[0].item(), want.mean().item(), rtol=1e-4)
    np.testing.assert_allclose(stats[1].item(), want.std().item(), rtol=1e-4)
    wm = epipolar.rigid_mask_torch(want, 2.0)
    assert (rigid != wm).float().mean().item() <= 1e-4  # pixels within rounding of the threshold
    # epipolar-consistent pixels are rigid, the independently moving block mostly is not
    assert rigid[:90].float().mean().item() > 0.99
    assert rigid[100:140, 180:240].float().mean().item() < 0.2
    # exact two-view flow without the block and without noise: distances are rounding-level
    fl0 = _two_view_flow(H, W, K, w1, w2, depth)
    _, d0, _ = epipolar.rigid_mask(fl0, F, 2.0)
    assert d0.max().item() < 1e-3


@pytest.mark.parametrize("H,W", [(256, 320), (1024, 1280)])
def test_flow_targets_kernels_match_the_torch_statement(H, W):
    """csrc/flow.hip flow_targets_* (hash sort + neighbour test) against backproject_previous (torch.unique(dim=0) of
    the reference): same kept set and order.  Constructed duplicates are decided identically by both; a coordinate
    within an ulp of a 1e-4 rounding boundary may go either way (FMA chain vs 4x4 GEMM), so <= 1e-5 of the points
    may differ."""
    from fsgs_amd import synth

    K = synth.intrinsics(W, H).copy()
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0   # integer principal point: pixel u mirrors W - u exactly
    u = torch.arange(W, device=DEV).float()[None] / W
    v = torch.arange(H, device=DEV).float()[:, None] / H
    depth = (1.0 + 0.3 * torch.sin(6.28 * u) * torch.cos(6.28 * v)).reshape(1, H, W).contiguous()
    depth[0, :7, :9] = 0.0            # invalid pixels
    depth[0, 100:104, :] = 1.0        # a fronto-parallel strip: x <-> -x mirror pairs collide in |round(., 4)|
    rigid = torch.rand(H, W, device=DEV) > 0.1
    for w_prev in (np.eye(4, dtype=np.float32),
                   synth.pose_matrix((1, 0.01, -0.02, 0.005), (0.01, 0.02, -0.01)).astype(np.float32)):
        for rm in (rigid, None):
            pa, va = flow.backproject_previous(depth, K, w_prev, rm)
            pb, vb = flow.backproject_previous_hip(depth, K, w_prev, rm)
            ka = (va[:, 0] * W + va[:, 1]).cpu().numpy()
            kb = (vb[:, 0] * W + vb[:, 1]).cpu().numpy()
            assert np.all(np.diff(kb) > 0)  # pixel order, like boolean indexing
            sym = np.setxor1d(ka, kb)
            # (1e-5 of the points at 256x320; 2.4e-5 at 1280x1024, where the world coordinates span more 1e-4 boundaries)
            assert len(sym) <= max(2, int(5e-5 * len(ka))), (len(sym), len(ka))
            both = np.intersect1d(ka, kb)
            ia, ib = np.searchsorted(ka, both), np.searchsorted(kb, both)
            np.testing.assert_allclose(pb[ib].cpu().numpy(), pa[ia].cpu().numpy(), rtol=1e-5, atol=1e-6)
            if np.allclose(w_prev, np.eye(4)):
                assert len(ka) < int((depth[0] * (rm if rm is not None else 1) > 0).sum())  # duplicates were dropped
    # nothing valid
    pts, vu = flow.backproject_previous_hip(torch.zeros(1, H, W, device=DEV), K, np.eye(4, dtype=np.float32), None)
    assert pts.shape == (0, 3) and vu.shape == (0, 2)


@pytest.mark.parametrize("seed", range(10))
def test_randomised_flow_loss_sizes_masks_and_poses(seed):
    """image sizes down to the 20-pixel border the loss discards (nothing survives -> exactly zero), holes in the depth,
    sparse and empty rigid masks, poses that push part of the points behind the camera; the plain and the fused
    (forward + pose gradient in one pass) entry points against the torch statement."""
    from fsgs_amd import synth

    rng = np.random.default_rng(400 + seed)
    H, W = int(rng.integers(30, 200)), int(rng.integers(30, 260))
    if seed == 0:
        H, W = 40, 41  # the border band (20 < u < W - 20) leaves no pixel
    g = torch.Generator(device=DEV).manual_seed(seed)
    K = synth.intrinsics(W, H)
    depth = (0.5 + torch.rand((1, H, W), device=DEV, generator=g)).contiguous()
    depth[0][torch.rand((H, W), device=DEV, generator=g) < 0.2] = 0.0
    fl = 3.0 * torch.randn((2, H, W), device=DEV, generator=g)
    keep = float(rng.choice([0.0, 0.02, 0.5, 1.0]))
    rigid = torch.rand((H, W), device=DEV, generator=g) < keep if keep < 1.0 else None
    w_prev = synth.pose_matrix(np.array([1.0, 0, 0, 0]) + 0.01 * rng.standard_normal(4), 0.02 * rng.standard_normal(3))
    w_prev = w_prev.astype(np.float32)
    q = np.array([1.0, 0, 0, 0]) + float(rng.choice([0.01, 0.3])) * rng.standard_normal(4)
    t = float(rng.choice([0.02, 0.6])) * rng.standard_normal(3)
    w_cur = torch.tensor(synth.pose_matrix(q, t).astype(np.float32), device=DEV)
    targets = flow.FlowTargets(depth, w_prev, K, fl, rigid)
    a = w_cur.clone().requires_grad_(True)
    la = flow.flow_pose_loss(a, targets)
    (la + 0.0 * a.sum()).backward()  # an empty selection returns a constant 0 (scene/pose_optimizer.py:182-183,217)
    b = w_cur.clone().requires_grad_(True)
    lb = flow.projection_flow_loss_torch(depth, w_prev, b, K, fl, rigid)
    (lb + 0.0 * b.sum()).backward()
    assert abs(la.item() - lb.item()) <= 3e-5 * abs(lb.item()) + 1e-7, (H, W, keep)
    scale = max(b.grad[:3].abs().max().item(), 1e-12)
    assert (a.grad[:3] - b.grad[:3]).abs().max().item() <= 1e-3 * scale, (H, W, keep)
    if lb.item() == 0.0:
        assert la.item() == 0.0 and not bool(a.grad.abs().sum())


@pytest.mark.parametrize("seed", range(8))
def test_randomised_sampson_mask_sizes(seed):
    """pixel counts below, at and across the 4096-pixel chunks of the two-pass statistics; random relative poses and
    threshold factors; the `dist < (dist <= mean + factor std)` quirk of the reference decides the mask."""
    from fsgs_amd import epipolar, synth

    rng = np.random.default_rng(700 + seed)
    H, W = [(1, 1), (3, 5), (64, 64), (64, 65), (17, 241), (100, 123), (128, 160), (211, 97)][seed]
    K = synth.intrinsics(max(W, 8), max(H, 8))
    w1 = synth.pose_matrix(np.array([1.0, 0, 0, 0]) + 0.01 * rng.standard_normal(4), 0.02 * rng.standard_normal(3))
    w2 = synth.pose_matrix(np.array([1.0, 0, 0, 0]) + 0.02 * rng.standard_normal(4), 0.05 * rng.standard_normal(3))
    g = torch.Generator(device=DEV).manual_seed(seed)
    fl = 2.0 * torch.randn((2, H, W), device=DEV, generator=g)
    F = epipolar.fundamental_from_w2c(w1, w2, K)
    factor = float(rng.choice([0.5, 2.0, 3.0]))
    rigid, dist, stats = epipolar.rigid_mask(fl, F, factor)
    want = epipolar.sampson_distance_torch(fl, F)
    scale = max(want.abs().max().item(), 1e-30)
    assert (dist - want).abs().max().item() <= 2e-4 * scale
    np.testing.assert_allclose(stats[0].item(), want.mean().item(), rtol=2e-4)
    if H * W > 1:
        np.testing.assert_allclose(stats[1].item(), want.std().item(), rtol=2e-4)
    wm = epipolar.rigid_mask_torch(want, factor)
    assert (rigid != wm).float().mean().item() <= max(1e-4, 1.01 / (H * W)) if H * W > 1 else True


@pytest.mark.parametrize("H,W", [(1024, 1280), (1080, 1920)])
def test_sampson_rigid_mask_at_full_size(H, W):
    """the once-per-frame rigid mask (scene/pose_optimizer.py:700-746, train.py:158-162) at C2 / C4 image size: distances,
    mean / std partials over up to 2.07 M pixels and the thresholded mask against the torch statement."""
    from fsgs_amd import epipolar, synth

    torch.manual_seed(3)
    K = synth.intrinsics(W, H)
    w1 = synth.pose_matrix((1, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    w2 = synth.pose_matrix((1, 0.004, -0.006, 0.002), (0.02, -0.01, 0.004))
    u = torch.arange(W, device=DEV).float()[None] / W
    v = torch.arange(H, device=DEV).float()[:, None] / H
    depth = 1.0 + 0.3 * torch.sin(6.28 * u) * torch.cos(6.28 * v)
    fl = _two_view_flow(H, W, K, w1, w2, depth, moving=(H // 3, H // 2, W // 2, 3 * W // 4, 6.0, -4.0))
    fl = fl + 0.02 * torch.randn_like(fl)
    F = epipolar.fundamental_from_w2c(w1, w2, K)
    rigid, dist, stats = epipolar.rigid_mask(fl, F, 2.0)
    want = epipolar.sampson_distance_torch(fl, F)
    # (p2^T F p1 cancels between terms of size |F| * 1e3 at these pixel coordinates: two fp32 evaluations agree to a few
    # 1e-4 of the largest distance -- 2.4e-4 measured at 1280x1024 -- against 2e-4 at 320x256)
    assert (dist - want).abs().max().item() <= 1e-3 * want.abs().max().item()
    np.testing.assert_allclose(stats[0].item(), want.double().mean().item(), rtol=1e-4)
    np.testing.assert_allclose(stats[1].item(), want.double().std().item(), rtol=1e-4)
    assert (rigid != epipolar.rigid_mask_torch(want, 2.0)).float().mean().item() <= 1e-4
    assert rigid[H // 3:H // 2, W // 2:3 * W // 4].float().mean().item() < 0.2
