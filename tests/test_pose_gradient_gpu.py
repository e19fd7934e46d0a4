"""The camera-pose gradient as a CLOSED check at full size (against the CPU reference render the 12
entries of dL/dpose all ride on depth-order near-tie allowances, because one re-ordered pair moves every entry).

dL/dw2c = sum_i g_i [x_i; 1]^T with g_i = dL/d(camera-frame mean of Gaussian i) (scene/pose_optimizer.py:985-987 adjoint), and
with the SH degree at 0 the colours do not depend on the position, so the SAME launch hands out g_i through the position
gradient: dL/dxyz_i = W^T g_i (W = rotation of w2c).  The per-Gaussian gradients are held against the oracle element by element
elsewhere (tests/test_full_size_oracle_gpu.py); here the P-term reduction on top of them -- wave sums, per-workgroup partials,
atomics or the deterministic finish kernel -- is held against the fp64 contraction of the kernel's own per-Gaussian output,
with no allowance: what is left is fp32 summation error."""
import numpy as np
import pytest
import torch

from fsgs_amd import rasterizer, render_ops, synth
from fsgs_amd.model import GaussianCloud
from fsgs_amd.trainer import PoseTrack, settings_from_cam

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("W,H,P", [(640, 512, 20000), (1280, 1024, 300_000), (1920, 1080, 1_000_000)])
def test_pose_gradient_is_the_contraction_of_the_per_gaussian_mean_gradients(W, H, P, deterministic):
    from simple_knn._C import distCUDA2

    knn = lambda pts: distCUDA2(torch.tensor(pts, device=DEV)).cpu().numpy()
    sc = synth.trained_like_scene(W, H, P, seed=0, knn_fn=knn)
    cam = synth.make_camera(W, H)
    pc = GaussianCloud(dict(sc), sh_degree=3, device=DEV)
    pc.cam = settings_from_cam(cam, DEV)
    pc.active_sh_degree = 0
    poses = PoseTrack(2, DEV)
    poses.set_pose(1, q=synth.PERTURBED_POSE["q"], t=synth.PERTURBED_POSE["t"])
    with torch.no_grad():
        w2c0 = poses.get_pose(1).detach().clone()
    g = torch.Generator(device="cpu").manual_seed(11)
    wi = (torch.rand(3, H, W, generator=g) - 0.5).to(DEV) / (H * W)
    wd = (torch.rand(3, H, W, generator=g) - 0.5).to(DEV) / (H * W)
    prev = rasterizer.set_deterministic(deterministic)
    try:
        w2c = w2c0.clone().requires_grad_(True)
        for v in pc.params.values():
            v.grad = None
        means2D = torch.zeros_like(pc.params["_xyz"], requires_grad=True) + 0
        im, depth_sil, radii = render_ops.fused_render(pc, w2c, means2D, poses.cam_center, gs_grad=True, cam_grad=True)
        ((im * wi).sum() + (depth_sil * wd).sum()).backward()
        torch.cuda.synchronize()
    finally:
        rasterizer.set_deterministic(prev)
    dw = w2c.grad.detach().cpu().numpy().astype(np.float64)[:3]
    dxyz = pc.params["_xyz"].grad.detach().cpu().numpy().astype(np.float64)
    x = pc.params["_xyz"].detach().cpu().numpy().astype(np.float64)
    Wm = w2c0.cpu().numpy().astype(np.float64)[:3, :3]
    assert np.allclose(Wm @ Wm.T, np.eye(3), atol=1e-6)  # (LearnPose normalises the quaternion: W^-T = W)
    gcam = dxyz @ Wm.T  # rows g_i = W dxyz_i
    x4 = np.concatenate([x, np.ones((P, 1))], axis=1)
    want = gcam.T @ x4  # [3,4]: sum_i g_i [x_i;1]^T
    # the terms of an entry cancel: the yardstick is the sum of their magnitudes, of which fp32 accumulation in blocks of
    # 256 and ~P/256 partials loses a few 1e-7
    mag = np.abs(gcam).T @ np.abs(x4)
    err = np.abs(dw - want)
    assert (err <= 2e-6 * mag + 1e-12).all(), (err / mag).max()
    assert int((radii > 0).sum()) > 0.5 * P and float(np.abs(want).max()) > 0
