"""The autograd-free step (fsgs_amd/fast_step.py: every stage one C-ABI call, chain rule by hand) must be
the same computation as the autograd path (fsgs_amd/trainer.py), which in turn is pinned against the
reference's call sequence by tests/test_render_gpu.py / test_loss_gpu.py."""
import numpy as np
import pytest
import torch

from fsgs_amd import losses, synth
from fsgs_amd.fast_step import FastStepper
from fsgs_amd.model import PARAM_NAMES, GaussianCloud
from fsgs_amd.trainer import FrameData, PoseTrack, mapping_step, settings_from_cam, tracking_step

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _world(seed=0, W=320, H=256, P=6000, n=3):
    cam = synth.make_camera(W, H)
    sc = synth.trained_like_scene(W, H, P, seed=seed)
    pc = GaussianCloud(sc, sh_degree=3, device=DEV)
    pc.cam = settings_from_cam(cam, DEV)
    pc.active_sh_degree = 2
    pc.training_setup()
    poses = PoseTrack(n, DEV)
    poses.set_pose(1, q=synth.PERTURBED_POSE["q"], t=synth.PERTURBED_POSE["t"])
    poses.set_pose(2, q=(1, -0.01, 0.005, 0.0), t=(-0.01, 0.02, 0.01))
    g = torch.Generator().manual_seed(seed)
    colors = [torch.rand(3, H, W, generator=g).to(DEV) for _ in range(n)]
    monos = [(torch.rand(H, W, generator=g) + 0.5).to(DEV) for _ in range(n)]
    flows = [torch.randn(2, H, W, generator=g).to(DEV) for _ in range(n - 1)]
    frames = FrameData(colors, monos, flows_fw=flows, K=cam["K"])
    return pc, poses, frames, cam


@pytest.mark.parametrize("views", [[1], [2, 1]])
def test_fast_mapping_step_equals_autograd_step(views):
    H, W = 256, 320
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    a = _world()
    b = _world()
    from fsgs_amd import optim, trainer

    fs = FastStepper(b[0], b[1], b[2])
    la = lb = None
    for _ in range(2):  # two consecutive steps: gradient buffers are reused, Adam state advances
        loss, first = 0, None
        for ts in views:  # autograd path, same patch corners for every view
            pkg = trainer.render(a[1], ts, a[0], gs_grad=True, cam_grad=False)
            loss = loss + trainer.mapping_loss(pkg, a[2].colors[ts], a[2].monodeps[ts], corners)
            first = first or pkg
        loss.backward()
        optim.densify_stats(first["radii"], first["viewspace_points"].grad, a[0].variables["max_radii2D"],
                            a[0].variables["xyz_gradient_accum"], a[0].variables["denom"])
        a[0].optimizer.step()
        a[0].optimizer.zero_grad(set_to_none=True)
        la = loss.detach()
        lb = fs.mapping_step(views, corners=corners)
    assert abs(la.item() - lb.item()) <= 1e-5 * abs(la.item())
    for k in PARAM_NAMES:
        pa, pb = a[0].params[k].detach(), b[0].params[k].detach()
        # Adam normalises the step: a few flipped alpha decisions move single Gaussians by O(lr)
        diff = (pa - pb).abs()
        assert (diff > 1e-5 * pa.abs().max()).float().mean().item() < 2e-3, k
    for k in ("max_radii2D", "denom"):
        assert torch.equal(a[0].variables[k], b[0].variables[k]), k
    assert torch.allclose(a[0].variables["xyz_gradient_accum"], b[0].variables["xyz_gradient_accum"], rtol=1e-3,
                          atol=1e-9)


def test_fast_tracking_step_equals_autograd_step():
    from fsgs_amd.flow import FlowTargets

    H, W = 256, 320
    a = _world(seed=1)
    b = _world(seed=1)
    rigid = torch.rand(H, W, device=DEV) > 0.1
    depth_prev = (torch.rand(1, H, W, device=DEV) + 0.5)
    out = []
    for w, fast in ((a, False), (b, True)):
        pc, poses, frames, cam = w
        poses.initialize_tracking_optimizer(50)
        targets = FlowTargets(depth_prev, np.eye(4, dtype=np.float32), frames.K, frames.flows_fw[0], rigid)
        if fast:
            fs = FastStepper(pc, poses, frames)
            for _ in range(3):
                l = fs.tracking_step(1, targets, rigid)
        else:
            for _ in range(3):
                l = tracking_step(pc, poses, frames, 1, targets, rigid)
        out.append((l[0].item(), poses.r.detach().clone(), poses.t.detach().clone()))
    assert abs(out[0][0] - out[1][0]) <= 1e-4 * abs(out[0][0])
    assert torch.allclose(out[0][1], out[1][1], rtol=0, atol=2e-4)  # Adam steps are +-lr sized: compare poses
    assert torch.allclose(out[0][2], out[1][2], rtol=0, atol=2e-4)


def test_adam_fused_into_the_backward_equals_backward_then_adam():
    """fsgs_render_backward_adam (single view, single rank: the gradient of every parameter is consumed by its
    Adam update inside the preprocess backward) against the two-launch form, over three consecutive steps."""
    H, W = 256, 320
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    a = _world()
    b = _world()
    fa, fb = FastStepper(a[0], a[1], a[2]), FastStepper(b[0], b[1], b[2])
    fa.fuse_adam = False
    assert fb.fuse_adam
    for step in range(3):
        a[0].update_learning_rate(step + 1)
        b[0].update_learning_rate(step + 1)
        la = fa.mapping_step([1], corners=corners)
        lb = fb.mapping_step([1], corners=corners)
        assert abs(la.item() - lb.item()) <= 1e-5 * abs(la.item())
    for k in PARAM_NAMES:
        pa, pb = a[0].params[k], b[0].params[k]
        sa, sb = a[0].optimizer.state[pa], b[0].optimizer.state[pb]
        assert int(sa["step"]) == int(sb["step"]) == 3
        # identical arithmetic; only the atomic summation order of the blend backward differs between two runs
        diff = (pa.detach() - pb.detach()).abs()
        assert (diff > 1e-5 * pa.detach().abs().max()).float().mean().item() < 2e-3, k
        for key in ("exp_avg", "exp_avg_sq"):
            scale = sa[key].abs().max().item()
            assert ((sa[key] - sb[key]).abs() > 1e-4 * scale).float().mean().item() < 2e-3, (k, key)
    assert torch.equal(a[0].variables["denom"], b[0].variables["denom"])


def test_mapping_step_without_statistics_is_the_same_update():
    """collect_stats=False (iterations past the last densification) drops the RGB-only mean2D terms from the backward
    blend and skips the statistics kernel; parameters and loss must not notice."""
    H, W = 256, 320
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    a = _world()
    b = _world()
    fa, fb = FastStepper(a[0], a[1], a[2]), FastStepper(b[0], b[1], b[2])
    for views in ([1], [2, 1]):
        la = fa.mapping_step(views, corners=corners)
        lb = fb.mapping_step(views, corners=corners, collect_stats=False)
        assert abs(la.item() - lb.item()) <= 1e-5 * abs(la.item())
    for k in PARAM_NAMES:
        pa, pb = a[0].params[k].detach(), b[0].params[k].detach()
        assert ((pa - pb).abs() > 1e-5 * pa.abs().max()).float().mean().item() < 2e-3, k
    assert float(a[0].variables["denom"].sum()) > 0 and float(b[0].variables["denom"].sum()) == 0


def test_full_size_c2_gradient_routes_agree():
    """BASELINE.json's C2 (1280x1024, 300 000 Gaussians): the three gradient routes of the step driver -- Adam inside the
    backward, the compact [P,14] gradient, full gradients + multi-tensor Adam -- must produce the same update at full
    size (size-dependent indexing: 300k x 45 SH floats, 5120 tiles, ~1.1 M pairs)."""
    import bench

    outs = []
    for route in ("fused", "compact", "full"):
        torch.manual_seed(0)
        pc, poses, frames, cam, sc = bench.build_problem("C2", DEV, 0, 1, n_frames=2)
        fs = FastStepper(pc, poses, frames)
        H, W = 1024, 1280
        g = torch.Generator().manual_seed(3)
        n = int(0.5 * (H // 128) * (W // 128))
        corners = (torch.randint(0, H - 128, (n,), generator=g).to(DEV), torch.randint(0, W - 128, (n,), generator=g).to(DEV))
        if route == "compact":
            fs.fuse_adam = False
        if route == "full":
            fs.fuse_adam = fs.compact = False
        loss = fs.mapping_step([1], corners=corners)
        torch.cuda.synchronize()
        assert torch.isfinite(loss)
        outs.append(({k: pc.params[k].detach().clone() for k in PARAM_NAMES}, loss.item(),
                     pc.variables["denom"].sum().item()))
        del pc, fs
    ref = outs[0]
    for other in outs[1:]:
        assert abs(other[1] - ref[1]) <= 1e-5 * abs(ref[1]) and other[2] == ref[2]
        for k in PARAM_NAMES:
            a, b = ref[0][k], other[0][k]
            assert torch.isfinite(b).all()
            # one Adam step of size lr: elements whose gradient is atomics-order noise may differ by ~lr
            assert ((a - b).abs() > 1e-5 * a.abs().max()).float().mean().item() < 2e-3, k


def test_four_million_gaussians_step_and_track():
    """4x the cloud of BASELINE.json's largest configuration (C4: 1920x1080, 1 M): every size-dependent index of the
    step -- 4 M x 45 SH floats, 8160 tiles x 8 sub-lists, the compact gradient, the fixed-capacity pair segments and
    their overflow retry -- in one mapping and one tracking iteration; finite results, every visible Gaussian updated,
    and the dense Adam moved the invisible ones' moments too."""
    import bench
    from fsgs_amd.flow import FlowTargets

    torch.manual_seed(0)
    pc, poses, frames, cam, sc = bench.build_problem("C4x4", DEV, 0, 1, n_frames=2)
    assert pc.num_points == 4_000_000
    fs = FastStepper(pc, poses, frames)
    before = {k: pc.params[k].detach().clone() for k in PARAM_NAMES}
    loss = fs.mapping_step([1])
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    vis = fs.last["radii"] > 0
    assert int(vis.sum()) > 2_000_000
    for k in PARAM_NAMES:
        assert torch.isfinite(pc.params[k]).all(), k
    moved = (pc.params["_opacity"].detach() != before["_opacity"]).reshape(-1)
    assert float(moved[vis].float().mean()) > 0.95  # (a visible Gaussian behind opaque ones may get an exact-zero gradient)
    assert int(pc.optimizer.state[pc.params["_xyz"]]["step"]) == 1
    # tracking on the same cloud
    H, W = 1080, 1920
    poses.initialize_tracking_optimizer(50)
    targets = FlowTargets(frames.monodeps[0].reshape(1, H, W), np.eye(4, dtype=np.float32), cam["K"],
                          torch.zeros((2, H, W), device=DEV), torch.ones((H, W), dtype=torch.bool, device=DEV))
    r0 = poses.r.detach().clone()
    total, rgb, flow = fs.tracking_step(1, targets, None)
    torch.cuda.synchronize()
    assert torch.isfinite(total) and torch.isfinite(poses.r).all() and not torch.equal(poses.r, r0)
