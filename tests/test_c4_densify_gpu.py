"""BASELINE.json configs[3] as it is stated -- "1M-Gaussian synthetic scene, 1920x1080, densification ON" -- as a test
(VERDICT r2 missing #5): the bench's C4 problem, densification statistics accumulated by real mapping steps of the HIP
step driver, then `densify_and_prune_device` (csrc/densify.hip) against the reference's own clone / cat / split / cat /
prune / prune sequence (scene/gaussian_model.py:523-676, train.py:297-316) run with torch on the same cloud and seed,
with a threshold under which the cloud GROWS (clone + split, 1 M -> >= 1.3 M) -- count, order, parameters, Adam moments
-- and one more mapping + one tracking iteration on the grown cloud."""
import numpy as np
import pytest
import torch

from fsgs_amd.model import PARAM_NAMES, GaussianCloud

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _clone_cloud(pc):
    """an independent copy: parameters, Adam moments / step counters, densification statistics, camera."""
    c = GaussianCloud({k: pc.params[k].detach().clone() for k in PARAM_NAMES}, sh_degree=pc.max_sh_degree, device=DEV,
                      spatial_lr_scale=pc.spatial_lr_scale, scene_radius=float(pc.variables["scene_radius"]))
    c.active_sh_degree, c.cam = pc.active_sh_degree, pc.cam
    c.training_setup(eps=pc.optimizer.param_groups[0]["eps"])
    by_name = {g["name"]: g for g in pc.optimizer.param_groups}
    for g in c.optimizer.param_groups:
        src = pc.optimizer.state[by_name[g["name"]]["params"][0]]
        c.optimizer.state[g["params"][0]] = {"step": int(src["step"]), "exp_avg": src["exp_avg"].clone(),
                                             "exp_avg_sq": src["exp_avg_sq"].clone()}
        g["lr"] = by_name[g["name"]]["lr"]
    for k in ("max_radii2D", "xyz_gradient_accum", "denom"):
        c.variables[k] = pc.variables[k].clone()
    return c


def test_c4_cloud_grows_like_the_reference_sequence_and_keeps_training():
    import bench
    from fsgs_amd.fast_step import FastStepper
    from fsgs_amd.flow import FlowTargets

    pc, poses, frames, cam, sc = bench.build_problem("C4", DEV, 0, 1)
    W, H, P0 = 1920, 1080, pc.num_points
    assert P0 == 1_000_000
    fs = FastStepper(pc, poses, frames)
    for it in range(6):  # train.py:260-263,298-303: statistics from view 0 of every mapping iteration
        loss = fs.mapping_step([it % 8])
    assert torch.isfinite(loss)
    den = pc.variables["denom"].reshape(-1)
    acc = pc.variables["xyz_gradient_accum"].reshape(-1)
    seen = den > 0
    assert int(seen.sum()) > P0 // 2 and float(den.max()) == 6.0
    grads = (acc / den)[seen]
    # the reference's threshold is 2e-4 (train.py:307) on gradients of ITS loss scale; here: the value above which 45 % of
    # the seen Gaussians lie, so that clone (small ones) + split (large ones) add >= 30 % whatever the prune removes
    thr = float(torch.quantile(grads[torch.randperm(len(grads), device=DEV)[:200_000]], 0.55))
    assert thr > 0
    a, b = _clone_cloud(pc), pc
    torch.manual_seed(11)
    # (size_threshold None: train.py:308 applies the 20-pixel screen-size prune only past iteration 4000)
    a.densify_and_prune(thr, 0.05, None)            # the reference's sequence, torch ops
    torch.manual_seed(11)
    info = b.densify_and_prune_device(thr, 0.05, None)
    print("C4 densify: %d -> %d (kept %d, cloned %d, split %d -> %d children kept)" % (
        P0, b.num_points, info["kept"], info["cloned"], info["split"], info["children_kept"]))
    assert a.num_points == b.num_points >= int(1.3 * P0), (a.num_points, b.num_points)
    assert info["cloned"] > 10_000 and info["split"] > 1_000 and info["kept"] < P0
    n_fix = info["kept"] + info["cloned"]
    for k in PARAM_NAMES:
        pa, pb = a.params[k].detach(), b.params[k].detach()
        if k == "_xyz":  # the children's positions: the reference leaves their 3x3 product to a batched GEMM
            assert torch.equal(pa[:n_fix], pb[:n_fix])
            assert (pa - pb).abs().max().item() <= 1e-6 * pa.abs().max().item()
        else:
            assert torch.equal(pa, pb), k
        sa, sb = a.optimizer.state[a.params[k]], b.optimizer.state[b.params[k]]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), k
        assert int(sa["step"]) == int(sb["step"]) and b.params[k].requires_grad
    for k in ("max_radii2D", "xyz_gradient_accum", "denom"):  # densification_postfix zeroes them (:615-617)
        assert torch.equal(a.variables[k], b.variables[k]) and not bool(b.variables[k].any())
        assert b.variables[k].shape[0] == b.num_points
    del a
    # the grown cloud keeps training: one mapping iteration (statistics restart) and one tracking iteration
    before = b.params["_xyz"].detach().clone()
    loss = fs.mapping_step([1])
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and not torch.equal(before, b.params["_xyz"].detach())
    assert float(b.variables["denom"].max()) == 1.0 and fs.last["P"] == b.num_points
    poses.initialize_tracking_optimizer(50)
    dep = frames.monodeps[0].reshape(1, H, W)
    tg = FlowTargets(dep, np.eye(4, dtype=np.float32), cam["K"], torch.zeros(2, H, W, device=DEV), None)
    r0 = poses.r.detach().clone()
    l_trk = fs.tracking_step(1, tg, None)[0]
    torch.cuda.synchronize()
    assert torch.isfinite(l_trk) and not torch.equal(r0, poses.r.detach())
    assert all(torch.isfinite(b.params[k]).all() for k in PARAM_NAMES)
