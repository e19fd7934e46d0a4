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


def test_two_view_step_overlapped_on_two_streams_equals_the_serial_step():
    """The progressive phase's two-view iteration (train.py:236-259): view 1's whole pipeline on its own stream and buffer
    set beside view 0's, the two compact gradients summed inside fsgs_adam_step_compact_sum, max_radii2D raised by BOTH
    views inside their backward launches (gaussian_renderer/__init__.py:79) -- against the same step with the views one
    after the other on one stream, and against the full-gradient route (gradients added by torch, torch.maximum for the
    radii); three views exercise the fold of the surplus views."""
    H, W = 256, 320
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    worlds = [_world() for _ in range(3)]
    steppers = [FastStepper(w[0], w[1], w[2]) for w in worlds]
    steppers[1].overlap_views = False
    steppers[2].fuse_adam = steppers[2].compact = False
    assert steppers[0].overlap_views
    for views in ([2, 1], [1, 2], [0, 1, 2], [1]):
        ls = [fs.mapping_step(views, corners=corners) for fs in steppers]
        torch.cuda.synchronize()
        assert all(abs(l.item() - ls[0].item()) <= 1e-5 * abs(ls[0].item()) for l in ls)
    ref = worlds[1][0]
    for w in (worlds[0], worlds[2]):
        pc = w[0]
        for k in PARAM_NAMES:
            pa, pb = ref.params[k].detach(), pc.params[k].detach()
            assert ((pa - pb).abs() > 1e-5 * pa.abs().max()).float().mean().item() < 2e-3, k
        for k in ("max_radii2D", "denom"):
            assert torch.equal(ref.variables[k], pc.variables[k]), k
        # four Adam steps apart by the arrival order of the backward's float atomics (twice as many, smaller ones with the
        # four-waves-per-tile backward this scene takes since round 5): the accumulated gradient norms agree to 1e-4 except
        # for a handful of Gaussians whose near-zero gradient moved them by O(lr) (1 run in 8 had one beyond 1e-4)
        a_, b_ = ref.variables["xyz_gradient_accum"], pc.variables["xyz_gradient_accum"]
        off = (a_ - b_).abs() > 1e-4 * a_.abs() + 1e-9
        assert off.float().mean().item() < 1e-3 and torch.allclose(a_, b_, rtol=1e-2, atol=1e-7)
    assert float(ref.variables["max_radii2D"].max()) > 0


def test_colour_reuse_check_switch_catches_a_write_without_a_version_bump(monkeypatch):
    """The per-Gaussian colours are copied from the previous forward's state while the parameters' version counters and
    the camera centre stand still (fast_step._render_forward).  FSGS_CHECK_REUSE=1 re-evaluates them: silent through
    legitimate sequences (tracking iterations, a mapping step in between), loud when a parameter is written behind the
    counters' back (`.data` writes do not bump them) -- and a bumped write simply ends the reuse."""
    from fsgs_amd.flow import FlowTargets

    monkeypatch.setenv("FSGS_CHECK_REUSE", "1")
    pc, poses, frames, cam = _world()
    H, W = 256, 320
    fs = FastStepper(pc, poses, frames)
    poses.initialize_tracking_optimizer(50)
    tg = FlowTargets(frames.monodeps[0].reshape(1, H, W), np.eye(4, dtype=np.float32), cam["K"], frames.flows_fw[0], None)
    for _ in range(3):
        fs.tracking_step(1, tg, None)
    fs.mapping_step([1, 2])
    for _ in range(2):
        fs.tracking_step(2, tg, None)
    torch.cuda.synchronize()
    pc.params["_features_dc"].data.add_(0.25)  # no version bump: the next forward would render stale colours
    with pytest.raises(RuntimeError, match="FSGS_CHECK_REUSE"):
        fs.tracking_step(2, tg, None)
    with torch.no_grad():
        pc.params["_features_dc"].add_(0.0)      # a counted write: the reuse ends, the colours are evaluated afresh
    fs.tracking_step(2, tg, None)
    torch.cuda.synchronize()


def test_colour_cache_left_by_the_adam_kernels_equals_fresh_evaluation(monkeypatch):
    """The Adam kernels (inside the backward for a single view, from the compact gradients otherwise) leave
    clamp_min(eval_sh + 0.5, 0) of the UPDATED parameters behind (FsgsFusedAdam.next_colors) and the next forward reads
    16 B per Gaussian instead of its SH block (fsgs_render_forward_cached_colors).  With FSGS_CHECK_REUSE=1 every such
    forward re-evaluates the colours and compares them bit for bit; the cache must actually be hit (single-view, two-view
    and tracking forwards behind a mapping step), must end with an SH-degree step or a foreign write to a parameter,
    and a run with the cache switched off must produce the same parameters."""
    from fsgs_amd.flow import FlowTargets

    monkeypatch.setenv("FSGS_CHECK_REUSE", "1")
    H, W = 256, 320
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    a, b = _world(), _world()
    fa, fb = FastStepper(a[0], a[1], a[2]), FastStepper(b[0], b[1], b[2])
    fb.cache_colors = False
    tg = FlowTargets(a[2].monodeps[0].reshape(1, H, W), np.eye(4, dtype=np.float32), a[3]["K"], a[2].flows_fw[0], None)
    a[1].initialize_tracking_optimizer(50)
    b[1].initialize_tracking_optimizer(50)
    hits = []
    for views in ([1], [2], [1, 2], [0], [2, 1]):
        fa.mapping_step(views, corners=corners)
        fb.mapping_step(views, corners=corners)
        hits.append(fa.cache_hits)
    assert hits == [0, 1, 3, 4, 6], hits          # every forward behind the first step's Adam reads the cache
    fa.tracking_step(1, tg, None)                   # ... and so does the first tracking forward behind a mapping step
    fb.tracking_step(1, tg, None)
    assert fa.cache_hits == 7 and fb.cache_hits == 0
    a[0].oneupSHdegree()                            # the colours depend on the active degree: the cache ends
    b[0].oneupSHdegree()
    fa.mapping_step([1], corners=corners)
    fb.mapping_step([1], corners=corners)
    assert fa.cache_hits == 7
    fa.mapping_step([2], corners=corners)
    fb.mapping_step([2], corners=corners)
    assert fa.cache_hits == 8
    with torch.no_grad():                           # a counted write by somebody else: evaluated afresh
        a[0].params["_features_dc"].add_(0.0)
        b[0].params["_features_dc"].add_(0.0)
    fa.mapping_step([1], corners=corners)
    fb.mapping_step([1], corners=corners)
    assert fa.cache_hits == 8
    torch.cuda.synchronize()
    for k in PARAM_NAMES:
        pa, pb = a[0].params[k].detach(), b[0].params[k].detach()
        assert ((pa - pb).abs() > 1e-5 * pa.abs().max()).float().mean().item() < 2e-3, k


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


@pytest.mark.parametrize("sched", [7, 8, 9])
def test_long_randomised_schedule_step_driver_tracks_the_autograd_path(sched):
    """ONE FastStepper (and one set of cached buffers, capacities and streams) driven through a seeded schedule of
    everything the harness does to it -- 1- and 2-view mapping iterations, tracking iterations with and without a
    rigid mask, densify + prune (P changes, up and down), opacity reset, SH degree steps -- each event replayed from
    the SAME state on the torch-autograd path.  Catches state carried from one call shape to the next."""
    from fsgs_amd import checkpoint, optim, trainer
    from fsgs_amd.flow import FlowTargets

    H, W = 256, 320
    a, b = _world(seed=5, P=5000), _world(seed=5, P=5000)
    for w in (a, b):
        w[0].active_sh_degree = 0
    fs = FastStepper(b[0], b[1], b[2])
    rng = np.random.default_rng(sched)

    def copy_state():  # b := a (parameters, Adam moments and step counts, statistics, SH degree, poses)
        checkpoint.restore_gaussians(b[0], checkpoint.capture_gaussians(a[0]))
        for k in ("max_radii2D", "xyz_gradient_accum", "denom"):
            b[0].variables[k] = a[0].variables[k].clone()
        for ga, gb in zip(a[0].optimizer.param_groups, b[0].optimizer.param_groups):
            gb["lr"] = ga["lr"]
        with torch.no_grad():
            b[1].r.copy_(a[1].r)
            b[1].t.copy_(a[1].t)
        optim.mark_updated([b[1].r, b[1].t])

    def close(pa, pb, name, frac=2e-3):
        if pa.numel() == 0:  # the schedule may prune the whole cloud away; the step must still run
            assert pb.numel() == 0, name
            return
        diff = (pa - pb).abs()
        assert (diff > 1e-5 * pa.abs().max() + 1e-12).float().mean().item() < frac, name

    log = []
    for step in range(28):
        ev = rng.choice(["M1", "M2", "T", "Tm", "D", "R", "S"], p=[0.3, 0.25, 0.1, 0.1, 0.15, 0.05, 0.05])
        copy_state()
        P = a[0].num_points
        log.append((ev, P))
        if ev in ("M1", "M2"):
            views = [int(rng.integers(0, 3))] if ev == "M1" else [int(rng.integers(0, 3)), int(rng.integers(0, 3))]
            corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
            loss, first = 0, None
            for ts in views:
                pkg = trainer.render(a[1], ts, a[0], gs_grad=True, cam_grad=False)
                loss = loss + trainer.mapping_loss(pkg, a[2].colors[ts], a[2].monodeps[ts], corners)
                first = first or pkg
            loss.backward()
            optim.densify_stats(first["radii"], first["viewspace_points"].grad, a[0].variables["max_radii2D"],
                                a[0].variables["xyz_gradient_accum"], a[0].variables["denom"])
            a[0].optimizer.step()
            a[0].optimizer.zero_grad(set_to_none=True)
            lb = fs.mapping_step(views, corners=corners)
            assert abs(loss.item() - lb.item()) <= 1e-5 * abs(loss.item()), (step, ev, log)
            for k in PARAM_NAMES:
                close(a[0].params[k].detach(), b[0].params[k].detach(), (step, ev, k, log))
            assert torch.equal(a[0].variables["denom"], b[0].variables["denom"]), (step, ev, log)
            assert torch.equal(a[0].variables["max_radii2D"], b[0].variables["max_radii2D"]), (step, ev, log)
        elif ev in ("T", "Tm"):
            rigid = (torch.rand(H, W, device=DEV) > 0.15) if ev == "Tm" else None
            depth_prev = torch.rand(1, H, W, device=DEV) + 0.5
            t = int(rng.integers(1, 3))
            res = []
            for w, fast in ((a, False), (b, True)):
                w[1].initialize_tracking_optimizer(50)
                targets = FlowTargets(depth_prev, np.eye(4, dtype=np.float32), w[2].K, w[2].flows_fw[0],
                                      rigid if rigid is not None else torch.ones(H, W, dtype=torch.bool, device=DEV))
                for _ in range(2):
                    l = fs.tracking_step(t, targets, rigid) if fast else tracking_step(w[0], w[1], w[2], t, targets, rigid)
                res.append(l[0].item())
            assert abs(res[0] - res[1]) <= 1e-4 * abs(res[0]), (step, ev, P, res, log)
            assert torch.allclose(a[1].r, b[1].r, rtol=0, atol=2e-4) and torch.allclose(a[1].t, b[1].t, rtol=0, atol=2e-4), (
                step, ev, P, (a[1].r - b[1].r).abs().max().item(), (a[1].t - b[1].t).abs().max().item())
        elif ev == "D":
            # make the statistics select something: ~10 % above the gradient threshold, a few huge / transparent ones
            with torch.no_grad():
                sel = torch.rand(P, 1, device=DEV) < 0.1
                a[0].variables["denom"] = torch.ones(P, 1, device=DEV)
                a[0].variables["xyz_gradient_accum"] = torch.where(sel, torch.full_like(sel, 1e-3, dtype=torch.float32),
                                                                   torch.zeros(P, 1, device=DEV))
                a[0].params["_opacity"][torch.rand(P, 1, device=DEV) < 0.03] = -6.0
                a[0].variables["max_radii2D"] = torch.rand(P, device=DEV) * 24.0  # ~1/6 above the screen-size limit
            copy_state()
            torch.manual_seed(100 + step)
            a[0].densify_and_prune(2e-4, 0.05, 20 if step % 2 else None)
            torch.manual_seed(100 + step)
            b[0].densify_and_prune_device(2e-4, 0.05, 20 if step % 2 else None)
            assert a[0].num_points == b[0].num_points and (a[0].num_points != P or P == 0), (step, log)
            for k in PARAM_NAMES:
                if k == "_xyz":  # the children's positions: batched GEMM there, FMAs here (tests/test_optim_gpu.py)
                    assert a[0].num_points == 0 or (
                        (a[0].params[k] - b[0].params[k]).abs().max() <= 1e-6 * a[0].params[k].abs().max()), (step, log)
                else:
                    assert torch.equal(a[0].params[k], b[0].params[k]), (step, k, log)
                sa, sb = a[0].optimizer.state[a[0].params[k]], b[0].optimizer.state[b[0].params[k]]
                assert ("exp_avg" in sa) == ("exp_avg" in sb)
                if "exp_avg" in sa:  # (no Adam state yet when the schedule densifies before the first step)
                    assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        elif ev == "R":
            a[0].reset_opacity()
            b[0].reset_opacity()
            assert torch.equal(a[0].params["_opacity"], b[0].params["_opacity"])
        else:
            a[0].oneupSHdegree()
            b[0].oneupSHdegree()
    kinds = {e for e, _ in log}
    print("schedule:", " ".join("%s@%d" % (e, p) for e, p in log))
    assert {"M1", "M2"} <= kinds and (("D" not in kinds) or len({p for _, p in log}) >= 2), log


@pytest.mark.parametrize("own_streams", [True, False])
def test_trainer_thread_and_viewer_thread_share_the_library(own_streams):
    """train.py runs the viewer's render_custom from a second Python thread while the training thread iterates
    (train.py:150,227-231; tracking iterations are not even locked): one thread steps a model through the step
    driver, another renders a different model in a loop -- on their own streams, and (what the reference does) both on the default
    stream, where the two threads' launches interleave.  The stepped model must end
    where a solitary run ends, and every concurrent render must equal the solitary render bit for bit."""
    import threading

    from fsgs_amd.render import render

    torch.manual_seed(0)
    H, W = 256, 320
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    solo, busy = _world(seed=2, P=5000), _world(seed=2, P=5000)
    view = _world(seed=9, P=3000)
    with torch.no_grad():
        want = render(view[1], 1, view[0], gs_grad=False, cam_grad=False)["render"].clone()
    fs_solo = FastStepper(solo[0], solo[1], solo[2])
    STEPS = 12
    for it in range(STEPS):
        fs_solo.mapping_step([it % 3], corners=corners)
    torch.cuda.synchronize()
    stop, bad, count, started = threading.Event(), [], [0], threading.Event()

    def viewer():
        s = torch.cuda.Stream() if own_streams else torch.cuda.current_stream()
        with torch.cuda.stream(s), torch.no_grad():
            while not stop.is_set():
                img = render(view[1], 1, view[0], gs_grad=False, cam_grad=False)["render"]
                if not torch.equal(img, want):
                    bad.append(count[0])
                count[0] += 1
                started.set()
        s.synchronize()

    def trainer():
        s = torch.cuda.Stream() if own_streams else torch.cuda.current_stream()
        started.wait(60)  # the viewer's loop is running (the step driver got fast enough to finish before its first render)
        with torch.cuda.stream(s):
            fs = FastStepper(busy[0], busy[1], busy[2])
            for it in range(STEPS):
                fs.mapping_step([it % 3], corners=corners)
        s.synchronize()

    tv, tt = threading.Thread(target=viewer), threading.Thread(target=trainer)
    tv.start()
    tt.start()
    tt.join()
    stop.set()
    tv.join()
    torch.cuda.synchronize()
    assert count[0] >= 2 and not bad, (count[0], bad[:5])  # (one render before the trainer starts + those beside its 12 steps)
    for k in PARAM_NAMES:
        a, b = solo[0].params[k].detach(), busy[0].params[k].detach()
        assert ((a - b).abs() > 1e-5 * a.abs().max()).float().mean().item() < 2e-3, k
    assert torch.equal(solo[0].variables["denom"], busy[0].variables["denom"])


@pytest.mark.parametrize("seed", range(6))
def test_randomised_image_sizes_step_driver_equals_autograd(seed):
    """one mapping and two tracking iterations at image sizes that are multiples of nothing (tile 16, loss tile 32,
    streaming segment 64, Pearson patch 128) and cloud sizes off the 256-Gaussian workgroup: step driver against the
    torch-autograd statement of the same iteration."""
    from fsgs_amd import optim, trainer
    from fsgs_amd.flow import FlowTargets

    rng = np.random.default_rng(900 + seed)
    W, H = int(rng.integers(131, 420)), int(rng.integers(130, 300))
    P = int(rng.integers(300, 6000))
    a, b = _world(seed=seed, W=W, H=H, P=P), _world(seed=seed, W=W, H=H, P=P)
    assert a[0].num_points == b[0].num_points
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    views = [1] if seed % 2 else [2, 0]
    fs = FastStepper(b[0], b[1], b[2])
    loss, first = 0, None
    for ts in views:
        pkg = trainer.render(a[1], ts, a[0], gs_grad=True, cam_grad=False)
        loss = loss + trainer.mapping_loss(pkg, a[2].colors[ts], a[2].monodeps[ts], corners)
        first = first or pkg
    loss.backward()
    optim.densify_stats(first["radii"], first["viewspace_points"].grad, a[0].variables["max_radii2D"],
                        a[0].variables["xyz_gradient_accum"], a[0].variables["denom"])
    a[0].optimizer.step()
    a[0].optimizer.zero_grad(set_to_none=True)
    lb = fs.mapping_step(views, corners=corners)
    assert abs(loss.item() - lb.item()) <= 1e-5 * abs(loss.item()), (W, H, P)
    for k in PARAM_NAMES:
        pa, pb = a[0].params[k].detach(), b[0].params[k].detach()
        assert ((pa - pb).abs() > 1e-5 * pa.abs().max()).float().mean().item() < 4e-3, (k, W, H, P)
    assert torch.equal(a[0].variables["denom"], b[0].variables["denom"])
    assert torch.equal(a[0].variables["max_radii2D"], b[0].variables["max_radii2D"])
    # tracking from the (slightly different) post-step clouds: losses agree to the clouds' agreement
    rigid = torch.rand(H, W, device=DEV) > 0.1
    depth_prev = torch.rand(1, H, W, device=DEV) + 0.5
    out = []
    for w, fast in ((a, False), (b, True)):
        w[1].initialize_tracking_optimizer(50)
        targets = FlowTargets(depth_prev, np.eye(4, dtype=np.float32), w[2].K, w[2].flows_fw[0], rigid)
        for _ in range(2):
            l = fs.tracking_step(1, targets, rigid) if fast else tracking_step(w[0], w[1], w[2], 1, targets, rigid)
        out.append(l[0].item())
    assert abs(out[0] - out[1]) <= 2e-3 * abs(out[0]), (W, H, P, out)
    assert torch.allclose(a[1].r, b[1].r, rtol=0, atol=4e-4) and torch.allclose(a[1].t, b[1].t, rtol=0, atol=4e-4)


def test_every_backward_of_the_step_driver_leaves_its_accumulators_zero():
    """FSGS_FLAG_SCRATCH_SELF_CLEAN: the per-Gaussian backward stores zeros over the accumulator rows it has read, so the
    driver never clears its scratch (allocated zero) -- after every kind of step (Adam inside the backward, compact gradient
    of one and of two views, full gradients, the pose-only tracking backward, the row-chunked producer route) the first
    P * 64 bytes must be zero again, on every buffer set that was used; and the C-ABI flag itself: the same backward with
    and without it gives the same gradients."""
    import ctypes as C

    from fsgs_amd import _lib
    from fsgs_amd.flow import FlowTargets

    pc, poses, frames, cam = _world()
    H, W = 256, 320
    fs = FastStepper(pc, poses, frames)
    assert fs._cfg().flags & _lib.FSGS_FLAG_SCRATCH_SELF_CLEAN
    poses.initialize_tracking_optimizer(50)
    tg = FlowTargets(frames.monodeps[0].reshape(1, H, W), np.eye(4, dtype=np.float32), cam["K"], frames.flows_fw[0], None)

    def clean():
        torch.cuda.synchronize()
        bufs = [fs.buf] + list(fs.__dict__.get("_view_bufs", {}).values())
        P = pc.num_points
        return all(int(b.bwd_scratch[:P * 64].count_nonzero()) == 0 for b in bufs if b is not None)

    fs.mapping_step([1]); assert clean()
    fs.mapping_step([2, 1]); assert clean()
    fs.tracking_step(1, tg, None); assert clean()
    fs.mapping_step([0, 1, 2]); assert clean()
    fs.mapping_step([1], step_optimizer=False); assert clean()
    fs.mapping_step([2], reduce_compact=lambda t: None); assert clean()

    class Chunked:  # the producer-side route of N > 1 (dist.ProducerPipelinedReducer), two row chunks, no exchange
        producer = True

        def bounds(self, P):
            mid = (P // 2) // 256 * 256
            return [(0, mid), (mid, P)]

        def produced(self, gc, lo, hi): pass
        def __call__(self, t): pass
        def finish(self, *a, **k): pass

    try:
        fs.mapping_step([1], reduce_compact=Chunked())
    except (AttributeError, TypeError):  # (the stub does not carry the whole reducer protocol: the backward ran, which is the point)
        pass
    assert clean()
    fs.tracking_step(2, tg, None); assert clean()
    # the flag at the C ABI: same gradients with and without it; without it the rows keep their sums
    args, state, sbytes, cap, nr = fs._render_forward(poses.get_pose_detached(1), fs.buf)
    b = fs.buf
    b.d_image.normal_(); b.d_depth_sil.zero_(); b.d_depth_sil[0].normal_()
    outs = []
    for flags_off in (0, _lib.FSGS_FLAG_SCRATCH_SELF_CLEAN):
        cfg = _lib.FsgsRasterCfg()
        C.memmove(C.byref(cfg), C.byref(fs._cfg()), C.sizeof(cfg))
        cfg.flags &= ~flags_off
        gc = torch.zeros((pc.num_points, 14), device=DEV)
        tail = _lib.FsgsStepTail()
        b.bwd_scratch.zero_()
        _lib.check(fs.lib.fsgs_render_backward_compact(C.byref(cfg), pc.num_points, C.byref(args), _lib.ptr(b.radii), _lib.ptr(state),
                                                       sbytes, cap, nr, _lib.ptr(b.d_image), _lib.ptr(b.d_depth_sil), _lib.ptr(gc), None,
                                                       C.byref(tail), _lib.ptr(b.bwd_scratch), b.bwd_scratch.numel(),
                                                       _lib.current_stream()), "fsgs_render_backward_compact")
        torch.cuda.synchronize()
        outs.append((gc.clone(), int(b.bwd_scratch[:pc.num_points * 64].count_nonzero())))
    (g_clean, nz_clean), (g_plain, nz_plain) = outs
    assert nz_clean == 0 and nz_plain > 0
    # (two runs of the same backward differ by the arrival order of the blend's float atomics only)
    assert ((g_clean - g_plain).abs() > 1e-5 * g_plain.abs().max()).float().mean().item() < 2e-3
    b.bwd_scratch.zero_()


def test_an_aborted_backward_does_not_poison_the_following_steps():
    """the driver runs its backwards with FSGS_FLAG_SCRATCH_ZEROED and trusts each one to leave the accumulator rows
    zero.  (a) A call the library REJECTS (here: an inconsistent FsgsStepTail) must not have launched the blend -- the rows
    stay zero; (b) a row-chunked backward that stops between two chunks (the exchange of dist.ProducerPipelinedReducer
    raising) leaves rows dirty: the driver notices and the next step clears them itself, so its update is the update of an
    undisturbed driver."""
    import ctypes as C

    from fsgs_amd import _lib

    pc, poses, frames, cam = _world()
    fs = FastStepper(pc, poses, frames)
    fs.mapping_step([1])
    b = fs.buf
    P = pc.num_points
    # (a) rejected before anything is launched
    args, state, sbytes, cap, nr = fs._render_forward(poses.get_pose_detached(1), b)
    b.d_image.normal_(); b.d_depth_sil.zero_(); b.d_depth_sil[0].normal_()
    tail = _lib.FsgsStepTail()
    tail.xyz_gradient_accum = pc.variables["xyz_gradient_accum"].data_ptr()  # ... without denom / max_radii2D: invalid
    gc = torch.zeros((P, 14), device=DEV)
    rc = fs.lib.fsgs_render_backward_compact(C.byref(fs._cfg_zeroed()), P, C.byref(args), _lib.ptr(b.radii), _lib.ptr(state),
                                             sbytes, cap, nr, _lib.ptr(b.d_image), _lib.ptr(b.d_depth_sil), _lib.ptr(gc), None,
                                             C.byref(tail), _lib.ptr(b.bwd_scratch), b.bwd_scratch.numel(), _lib.current_stream())
    torch.cuda.synchronize()
    assert rc == _lib.FSGS_ERR_INVALID
    assert int(b.bwd_scratch[:P * 64].count_nonzero()) == 0 and int(gc.count_nonzero()) == 0

    # (b) a producer route whose exchange dies after the first row chunk
    class Dies:
        producer = True

        def bounds(self, P_):
            mid = (P_ // 2) // 256 * 256
            return [(0, mid), (mid, P_)]

        def produced(self, gc_, lo, hi):
            raise RuntimeError("exchange failed")

    snap = {n: pc.params[n].detach().clone() for n in PARAM_NAMES}
    with pytest.raises(RuntimeError, match="exchange failed"):
        fs.mapping_step([1], reduce_compact=Dies())
    torch.cuda.synchronize()
    assert b.scratch_dirty and int(b.bwd_scratch[:P * 64].count_nonzero()) > 0  # the second half's rows hold the blend's sums
    assert all(torch.equal(snap[n], pc.params[n].detach()) for n in PARAM_NAMES)  # no Adam ran
    # an undisturbed twin in the same state (same parameters, fresh moments on both sides)
    pc2, poses2, frames2, _ = _world()
    for n in PARAM_NAMES:
        pc2.params[n].data.copy_(snap[n])
    pc.training_setup(); pc2.training_setup()
    fs2 = FastStepper(pc2, poses2, frames2)
    cr = losses.draw_patch_corners(256, 320, 128, 0.5, DEV)
    fs.mapping_step([2], reduce_compact=lambda t: None, corners=cr)
    fs2.mapping_step([2], reduce_compact=lambda t: None, corners=cr)
    torch.cuda.synchronize()
    assert not b.scratch_dirty and int(b.bwd_scratch[:P * 64].count_nonzero()) == 0
    for n in PARAM_NAMES:  # first Adam step = lr * sign-like: compare the updates, atomics-order tolerance
        da, db = pc.params[n].detach() - snap[n], pc2.params[n].detach() - snap[n]
        bad = ((da - db).abs() > 0.05 * db.abs().max()).float().mean().item()
        assert bad < 1e-2, (n, bad)


def test_begin_frame_is_initialize_pose_plus_a_fresh_optimizer():
    """PoseTrack.begin_frame (one launch: fsgs_pose_frame_begin, on an optimizer / schedule built once and reset) against
    what train.py:322-331 does per frame -- initialize_pose + a NEW Adam + MultiStepLR: same initial pose, same first
    iterations, also when the reused optimizer comes back dirty from the previous frame."""
    from fsgs_amd.flow import FlowTargets

    pc, poses, frames, cam = _world(n=4)
    H, W = 256, 320
    twin = PoseTrack(4, DEV)
    with torch.no_grad():
        twin.r.copy_(poses.r)
        twin.t.copy_(poses.t)
    tg = FlowTargets(frames.monodeps[2].reshape(1, H, W), np.eye(4, dtype=np.float32), cam["K"], frames.flows_fw[2], None)

    def track(pt, n=6):
        fs = FastStepper(pc, pt, frames)
        for _ in range(n):
            fs.tracking_step(3, tg, None, want_losses=False)
        torch.cuda.synchronize()
        return pt.r.detach()[0, :, 3].clone(), pt.t.detach()[:, 3].clone()

    # the reference's sequence
    twin.initialize_pose(3)
    twin.initialize_tracking_optimizer(50)
    # the one-launch form
    poses.begin_frame(3, 50)
    assert torch.allclose(poses.r, twin.r, atol=1e-7, rtol=0) and torch.allclose(poses.t, twin.t, atol=1e-7, rtol=0)
    assert not torch.equal(poses.r[0, :, 3], poses.r[0, :, 2])  # (extrapolated, not copied)
    ra, ta = track(twin)
    rb, tb = track(poses)
    # six Adam steps of lr 5e-3 move the pose by ~1e-2; the two runs differ by the order of the blend's float atomics
    assert (ra - rb).abs().max().item() < 2e-5 and (ta - tb).abs().max().item() < 2e-5, ((ra - rb).abs().max(), (ta - tb).abs().max())
    assert [g["lr"] for g in poses.optimizer.param_groups] == [g["lr"] for g in twin.optimizer.param_groups]
    assert poses.optimizer.state[poses.r]["step"] == 6
    # the same frame again on the now dirty optimizer: moments, step counters and schedule start over
    poses.begin_frame(3, 50)
    assert poses.optimizer.state[poses.r]["step"] == 0 and float(poses.optimizer.state[poses.t]["exp_avg"].abs().sum()) == 0.0
    rc, tc = track(poses)
    assert (rc - rb).abs().max().item() < 2e-5 and (tc - tb).abs().max().item() < 2e-5
    # frame 1 copies frame 0 (scene/pose_optimizer.py:513-516)
    poses.begin_frame(1, 50)
    assert torch.equal(poses.r[0, :, 1], poses.r[0, :, 0]) and torch.equal(poses.t[:, 1], poses.t[:, 0])


def test_forward_done_event_orders_another_stream_behind_the_forward_blend():
    """fsgs_forward_done_event: the next fused forward signals the event with its blend launch (no marker packet behind
    it); a second stream that waits for it must see the finished image -- also when the forward is repeated on the same
    buffers with other poses -- and a forward that fails before its blend records the event the plain way (the wait
    returns).  The event API rejects null handles."""
    import ctypes as C

    from fsgs_amd import _lib

    pc, poses, frames, cam = _world()
    fs = FastStepper(pc, poses, frames)
    lib = fs.lib
    ev = C.c_void_p()
    _lib.check(lib.fsgs_event_create(C.byref(ev)), "fsgs_event_create")
    assert ev.value
    assert lib.fsgs_stream_wait_event(_lib.current_stream(), None) == _lib.FSGS_ERR_INVALID
    assert lib.fsgs_event_create(None) == _lib.FSGS_ERR_INVALID
    other = torch.cuda.Stream()
    H, W = 256, 320
    b = fs._buffers(pc.num_points, H, W, 2, torch.device(DEV, torch.cuda.current_device()))
    for t in (1, 2, 0, 1):
        w2c = poses.get_pose_detached(t)
        b.image.fill_(-5.0)  # (on the current stream, in front of the forward)
        fs._render_forward(w2c, b, allow_reuse=False, done_event=ev)
        _lib.check(lib.fsgs_stream_wait_event(C.c_void_p(other.cuda_stream), ev), "fsgs_stream_wait_event")
        with torch.cuda.stream(other):
            seen = b.image.clone()  # reads behind the event only
        torch.cuda.synchronize()
        ref = b.image.clone()
        assert torch.equal(seen, ref) and float(ref.min()) >= 0.0 and float(ref.max()) > 0.0, t
    # a forward that leaves before its blend: the event is recorded anyway and a later waiter does not hang
    _lib.check(lib.fsgs_forward_done_event(ev), "fsgs_forward_done_event")
    cfg = fs._cfg()
    nr = C.c_int64(0)
    rc = lib.fsgs_render_forward(C.byref(cfg), -1, None, None, None, None, None, 0, None, 0, 0, C.byref(nr), _lib.current_stream())
    assert rc == _lib.FSGS_ERR_INVALID
    _lib.check(lib.fsgs_stream_wait_event(C.c_void_p(other.cuda_stream), ev), "fsgs_stream_wait_event")
    with torch.cuda.stream(other):
        torch.zeros(1, device=DEV).add_(1)
    other.synchronize()
    # the one-shot is spent: an ordinary forward afterwards does not touch the event
    fs._render_forward(poses.get_pose_detached(1), b, allow_reuse=False)
    torch.cuda.synchronize()
    _lib.check(lib.fsgs_event_destroy(ev), "fsgs_event_destroy")
    assert lib.fsgs_event_destroy(None) == _lib.FSGS_OK



@pytest.mark.parametrize("given_corners", [True, False])
def test_one_stream_loss_stage_is_the_same_step(given_corners):
    """FSGS_LOSS_STREAMS=1 (`one_stream_losses`: the view's loss stage as the two launches of
    fsgs_view_losses_forward_backward on the step's own stream, patch corners drawn on the side stream and joined by an
    event that has fired long before) against the default two-stream layout: the first step's loss terms and loss
    gradients are the same bits (the forward and the loss kernels are deterministic); further one- and two-view steps stay
    within the run-to-run noise of the backward's float atomics; both consume the RNG alike (train.py:236-258)."""
    H, W = 256, 320
    wa, wb = _world(), _world()
    fa, fb = FastStepper(wa[0], wa[1], wa[2]), FastStepper(wb[0], wb[1], wb[2])
    fa.one_stream_losses, fb.one_stream_losses = False, True
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV) if given_corners else None
    seeds = []
    for fs in (fa, fb):
        torch.manual_seed(5)
        fs.mapping_step([1], corners=corners)
        torch.cuda.synchronize()
        seeds.append(torch.cuda.get_rng_state().clone())
    assert torch.equal(seeds[0], seeds[1])
    ba, bb = fa.buf, fb.buf  # view 0's buffer set
    assert torch.equal(ba.terms[:5], bb.terms[:5]), (ba.terms, bb.terms)
    assert torch.equal(ba.d_image, bb.d_image) and torch.equal(ba.d_depth_sil[0], bb.d_depth_sil[0])
    assert float(bb.d_depth_sil[0].abs().max()) > 0 and float(bb.terms[4]) > 0
    for views in ([2, 1], [0], [1, 2, 0], [2]):
        ls = []
        for fs in (fa, fb):
            torch.manual_seed(11 + len(views))
            ls.append(fs.mapping_step(views, corners=corners))
        torch.cuda.synchronize()
        assert abs(ls[0].item() - ls[1].item()) <= 1e-5 * abs(ls[0].item())
    for k in PARAM_NAMES:
        pa, pb = wa[0].params[k].detach(), wb[0].params[k].detach()
        assert ((pa - pb).abs() > 1e-5 * pa.abs().max()).float().mean().item() < 2e-3, k
    for k in ("max_radii2D", "denom"):
        assert torch.equal(wa[0].variables[k], wb[0].variables[k]), k
