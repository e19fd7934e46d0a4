"""FSGS_AUTOBIND end to end on the GPU: the SAME stand-in checkout files (scripts/standin_checkout.py: the reference's module
and function names, bodies = this repository's restatement of the reference's own sequence) run a mapping iteration
(train.py:236-272,297-303) once as they are -- two drop-in rasteriser calls + torch glue, torch losses, torch.optim.Adam --
and once with fsgs_amd.autobind installed, which binds the fused render op, the HIP loss kernels and FusedAdam BY NAME.
The render dict, the loss values and the updated cloud must agree."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _run(tree, bind, steps=3):
    import standin_checkout
    from fsgs_amd import autobind, synth
    from fsgs_amd.model import PARAM_NAMES
    from fsgs_amd.trainer import PoseTrack, settings_from_cam

    standin_checkout.forget()
    if bind:
        autobind.install()
    sys.path.insert(0, tree)
    try:
        import standin_train
        from scene import GaussianModel

        W, H, P = 320, 256, 6000
        cam = synth.make_camera(W, H)
        sc = synth.trained_like_scene(W, H, P, seed=5)
        pc = GaussianModel(dict(sc), sh_degree=3, device=DEV)
        pc.cam = settings_from_cam(cam, DEV)
        pc.active_sh_degree = 2
        opt = pc.training_setup(eps=1e-15)
        poses = PoseTrack(3, DEV)
        poses.set_pose(1, q=synth.PERTURBED_POSE["q"], t=synth.PERTURBED_POSE["t"])
        g = torch.Generator().manual_seed(3)
        colors = [torch.rand(3, H, W, generator=g).to(DEV) for _ in range(3)]
        monos = [(torch.rand(H, W, generator=g) + 0.5).to(DEV) for _ in range(3)]
        names = {"render": standin_train.render.__module__, "rgb": standin_train.rgb_loss_func.__module__,
                 "pearson": standin_train.pearson_depth_loss.__module__, "local": standin_train.local_pearson_loss.__module__,
                 "adam": type(opt).__module__ + "." + type(opt).__name__}
        torch.manual_seed(11)  # local_pearson_loss draws its patch corners on the device (utils/loss_utils.py:114-121)
        losses, first = [], None
        for it in range(steps):
            loss, pkg = standin_train.mapping_iteration(poses, pc, colors, monos, 1)
            losses.append(float(loss))
            if first is None:
                first = {k: v.detach().clone() for k, v in pkg.items() if torch.is_tensor(v)}
        torch.cuda.synchronize()
        params = {k: pc.params[k].detach().clone() for k in PARAM_NAMES}
        stats = {k: pc.variables[k].detach().clone() for k in ("max_radii2D", "xyz_gradient_accum", "denom")}
        return names, losses, first, params, stats
    finally:
        sys.path.remove(tree)
        standin_checkout.forget()
        if bind:
            autobind.uninstall()


def test_autobound_standin_checkout_equals_the_unpatched_route(tmp_path):
    import standin_checkout

    tree = standin_checkout.write_tree(str(tmp_path))
    plain = _run(tree, bind=False)
    bound = _run(tree, bind=True)
    assert plain[0] == {"render": "gaussian_renderer", "rgb": "utils.loss_utils", "pearson": "utils.loss_utils",
                        "local": "utils.loss_utils", "adam": "torch.optim.adam.Adam"}
    assert bound[0] == {"render": "fsgs_amd.render", "rgb": "fsgs_amd.losses", "pearson": "fsgs_amd.losses",
                        "local": "fsgs_amd.losses", "adam": "fsgs_amd.optim.FusedAdam"}
    # the 10-key dict of the first iteration (same parameters on both sides): images to 1e-5 of scale, masks / radii exact
    assert set(plain[2]) == set(bound[2])
    for k in ("render", "render_dep", "render_opacity", "uncertainty", "render_w2c"):
        a, b = plain[2][k], bound[2][k]
        assert float((a - b).abs().max()) <= 1e-5 * (float(a.abs().max()) + 1.0), k
    for k in ("radii", "visibility_filter", "nan_mask"):
        assert torch.equal(plain[2][k], bound[2][k]), k
    assert float((plain[2]["presence_mask"] != bound[2]["presence_mask"]).float().mean()) < 1e-5  # silhouette at the 0.3 edge
    # the loss of every iteration (the later ones on parameters the two Adams have moved)
    for a, b in zip(plain[1], bound[1]):
        assert abs(a - b) <= 2e-4 * abs(a), (plain[1], bound[1])
    # the cloud after three Adam steps: Adam normalises the update, a few near-zero gradients move single elements by O(lr)
    for k, pa in plain[3].items():
        diff = (pa - bound[3][k]).abs()
        assert (diff > 1e-5 * pa.abs().max()).float().mean().item() < 2e-3, k
    assert torch.equal(plain[4]["max_radii2D"], bound[4]["max_radii2D"]) and torch.equal(plain[4]["denom"], bound[4]["denom"])
    # the accumulated viewspace-gradient norms: iterations 2 and 3 run on parameters the two Adams have moved apart by
    # O(1e-5 lr), so single Gaussians differ in the third digit; the bulk agrees to 1e-3
    a, b = plain[4]["xyz_gradient_accum"], bound[4]["xyz_gradient_accum"]
    off = (a - b).abs() > 1e-3 * a.abs() + 1e-9
    assert off.float().mean().item() < 0.01 and float((a - b).abs().max()) <= 0.05 * float(a.abs().max())
