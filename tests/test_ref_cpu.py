"""The CPU reference render of tests/ref_cpu.py (tests/ref_glue.py -- an independent restatement of the reference's render()
glue -- around the oracle rasteriser: the thing the fused HIP render is compared with on the GPU box) checked on its own: its autograd wiring against fp64 central
differences along random directions of every parameter tensor and of the pose."""
import numpy as np
import torch

from fsgs_amd import synth
from fsgs_amd.model import PARAM_NAMES, GaussianCloud
from fsgs_amd.trainer import PoseTrack, settings_from_cam
from tests import ref_cpu


def test_reference_render_gradients_match_fp64_central_differences(oracle64):
    W, H, P = 40, 32, 14
    cam = synth.make_camera(W, H)
    xyz, col, op, s, rot = synth.random_small_scene(P, cam, seed=3, zmin=0.8, zmax=1.4, scale_px=(3.0, 7.0))
    rng = np.random.default_rng(0)
    params = {"_xyz": xyz, "_features_dc": rng.normal(0, 0.6, (P, 1, 3)), "_features_rest": rng.normal(0, 0.15, (P, 15, 3)),
              "_opacity": np.log(op / (1 - op)).reshape(P, 1), "_scaling": np.log(s), "_rotation": rot}
    pc = GaussianCloud(params, sh_degree=3, device="cpu")
    pc.cam = settings_from_cam(cam, "cpu")
    pc.active_sh_degree = 2
    poses = PoseTrack(2, "cpu")
    poses.set_pose(1, (1.0, 0.01, -0.02, 0.015), (0.02, -0.01, 0.03))
    pc, poses = ref_cpu._to_double(ref_cpu.cpu_cloud(pc), ref_cpu.cpu_poses(poses))
    render_two_pass = ref_cpu.render_reference
    g = torch.Generator().manual_seed(1)
    wi = (torch.rand(3, H, W, generator=g, dtype=torch.float64) - 0.5)
    wd = (torch.rand(H, W, generator=g, dtype=torch.float64) - 0.5)
    ws = torch.zeros(H, W, dtype=torch.float64)

    def loss_value():
        with ref_cpu.oracle_backend(oracle64):
            pkg = render_two_pass(poses, 1, pc, gs_grad=True, cam_grad=True)
            return float((pkg["render"].detach() * wi).sum() + (pkg["render_dep"].detach() * wd).sum())

    with ref_cpu.oracle_backend(oracle64):
        out, grads = ref_cpu.run_render(render_two_pass, pc, poses, 1, True, True, wi, wd, ws)
    assert out["radii"].min() > 0
    tensors = {k: pc.params[k] for k in PARAM_NAMES}
    tensors.update(r=poses.r, t=poses.t)
    eps = 1e-6
    for k, t in tensors.items():
        d = torch.tensor(rng.normal(0, 1, tuple(t.shape)))
        if k in ("r", "t"):  # only frame 1 is rendered
            m = torch.zeros_like(d)
            m[..., 1] = 1.0
            d = d * m
        base = t.detach().clone()
        with torch.no_grad():
            t.copy_(base + eps * d)
        lp = loss_value()
        with torch.no_grad():
            t.copy_(base - eps * d)
        lm = loss_value()
        with torch.no_grad():
            t.copy_(base)
        fd = (lp - lm) / (2 * eps)
        an = float((torch.tensor(grads[k]) * d).sum())
        assert abs(fd - an) <= 2e-5 * max(abs(fd), abs(an)) + 1e-9, (k, fd, an)
