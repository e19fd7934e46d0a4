"""FusedAdam / densify_stats (csrc/optim.hip) against torch.optim.Adam and the reference's
boolean-index statistics (scene/gaussian_model.py:678-681) -- the plain PyTorch fp32 statement."""
import numpy as np
import pytest
import torch

from fsgs_amd.optim import FusedAdam, densify_stats

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("eps", [1e-15, 1e-8])
def test_fused_adam_tracks_torch_adam(eps):
    torch.manual_seed(0)
    P = 10007  # ragged on purpose: exercises the unaligned tail path
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)]
    lrs = [8e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3]
    a = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    oa = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], lr=0.0, eps=eps)
    ob = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=eps)
    for step in range(25):
        for p, q in zip(a, b):
            g = torch.randn_like(p) * (10.0 ** np.random.default_rng(step).uniform(-6, 0))
            p.grad = g.clone()
            q.grad = g.clone()
        oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 8e-4 * 0.99 ** step  # xyz lr schedule changes per step
        oa.step()
        ob.step()
    for p, q in zip(a, b):
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item() + 1e-7
    for p, q in zip(a, b):
        sa, sb = oa.state[p], ob.state[q]
        assert int(sa["step"]) == int(sb["step"])
        # moments cancel (m ~ sum of +- gradients): compare against the tensor's scale, not element-wise
        for key in ("exp_avg", "exp_avg_sq"):
            scale = sb[key].abs().max().item()
            assert (sa[key] - sb[key]).abs().max().item() <= 2e-6 * scale


def test_fused_adam_at_c4_size_tracks_torch_adam():
    """the same comparison at BASELINE.json's largest cloud (1 M Gaussians x 59 parameters = 236 MB per tensor set):
    64-bit element offsets, the float4 body and every group's tail at a size where 32-bit indices would wrap."""
    torch.manual_seed(1)
    P = 1_000_003
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)]
    lrs = [8e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3]
    a = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    oa = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    for step in range(4):
        for p, q in zip(a, b):
            p.grad = torch.randn_like(p) * 10.0 ** (-step)
            q.grad = p.grad.clone()
        oa.step()
        ob.step()
    for p, q in zip(a, b):
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item() + 1e-7
        sa, sb = oa.state[p], ob.state[q]
        for key in ("exp_avg", "exp_avg_sq"):
            assert (sa[key] - sb[key]).abs().max().item() <= 2e-6 * sb[key].abs().max().item()
        # the last elements of every tensor were reached
        assert not torch.equal(p.detach()[-1], torch.zeros_like(p.detach()[-1]))


def test_fused_adam_state_survives_densification_surgery():
    """cat_tensors_to_optimizer / prune replace params and slice the moments (scene/gaussian_model.py:523-580)."""
    p = torch.randn(100, 3, device=DEV).requires_grad_(True)
    opt = FusedAdam([{"params": [p], "lr": 1e-2, "name": "_xyz"}], eps=1e-15)
    p.grad = torch.randn_like(p)
    opt.step()
    st = opt.state.pop(p)
    new = torch.cat([p.detach()[:50], torch.zeros(10, 3, device=DEV)]).requires_grad_(True)
    st["exp_avg"] = torch.cat([st["exp_avg"][:50], torch.zeros(10, 3, device=DEV)])
    st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"][:50], torch.zeros(10, 3, device=DEV)])
    opt.param_groups[0]["params"][0] = new
    opt.state[new] = st
    new.grad = torch.ones_like(new)
    opt.step()
    assert int(opt.state[new]["step"]) == 2 and torch.isfinite(new).all()


def test_densify_stats_matches_boolean_index_statement():
    torch.manual_seed(1)
    P = 5003
    radii = (torch.rand(P, device=DEV) * 30).int() * (torch.rand(P, device=DEV) > 0.3)
    radii = radii.int()
    grad = torch.randn(P, 3, device=DEV)
    mr, acc, den = torch.rand(P, device=DEV) * 20, torch.rand(P, 1, device=DEV), torch.rand(P, 1, device=DEV).round()
    mr2, acc2, den2 = mr.clone(), acc.clone(), den.clone()
    densify_stats(radii, grad, mr, acc, den)
    vis = radii > 0
    mr2[vis] = torch.max(mr2[vis], radii[vis].float())
    acc2[vis] += torch.norm(grad[vis], dim=-1, keepdim=True)
    den2[vis] += 1
    assert torch.equal(mr, mr2) and torch.equal(den, den2)
    assert torch.allclose(acc, acc2, rtol=1e-6)


def _cloud_from_golden(g, fused):
    from fsgs_amd.model import PARAM_NAMES, GaussianCloud

    T = lambda a: torch.tensor(np.asarray(a), device=DEV)
    pc = GaussianCloud({k: g["p_" + k] for k in PARAM_NAMES}, device=DEV, scene_radius=float(g["var_scene_radius"]))
    pc.training_setup(fused=fused)
    for grp in pc.optimizer.param_groups:  # install the recorded Adam moments
        p = grp["params"][0]
        pc.optimizer.state[p] = {"step": 1, "exp_avg": T(g["m_" + grp["name"]]).clone(),
                                 "exp_avg_sq": T(g["v_" + grp["name"]]).clone()}
    pc.variables["max_radii2D"] = T(g["var_max_radii2D"]).clone()
    pc.variables["xyz_gradient_accum"] = T(g["var_xyz_gradient_accum"]).clone()
    pc.variables["denom"] = T(g["var_denom"]).clone()
    return pc


@pytest.mark.parametrize("max_screen", [20, None])
def test_device_densify_matches_the_reference_sequence_and_its_golden(max_screen):
    """csrc/densify.hip (plan + one gather) against (a) the reference's clone/cat/split/cat/prune/prune sequence
    run with torch on the same device and RNG seed -- bit-identical except the children's positions, whose 3x3
    product the reference leaves to a batched GEMM -- and (b) the vectors captured from the reference itself for
    everything that does not depend on the random draws."""
    import os

    from fsgs_amd.model import PARAM_NAMES

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify.npz"))
    a, b = _cloud_from_golden(g, True), _cloud_from_golden(g, True)
    torch.manual_seed(11)
    a.densify_and_prune(2e-4, 0.05, max_screen)
    torch.manual_seed(11)
    info = b.densify_and_prune_device(2e-4, 0.05, max_screen)
    assert a.num_points == b.num_points and b.num_points != int(g["P"])
    assert info["cloned"] > 0 and info["split"] > 0 and info["kept"] < int(g["P"])
    for k in PARAM_NAMES:
        pa, pb = a.params[k].detach(), b.params[k].detach()
        if k == "_xyz":
            assert torch.equal(pa[: info["kept"] + info["cloned"]], pb[: info["kept"] + info["cloned"]])
            assert (pa - pb).abs().max().item() <= 1e-6 * pa.abs().max().item()
        else:
            assert torch.equal(pa, pb), k
        sa, sb = a.optimizer.state[a.params[k]], b.optimizer.state[b.params[k]]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), k
        assert int(sa["step"]) == int(sb["step"]) and b.params[k].requires_grad
    for k in ("max_radii2D", "xyz_gradient_accum", "denom"):
        assert torch.equal(a.variables[k], b.variables[k]) and not bool(b.variables[k].any())
    if max_screen == 20:  # the configuration the golden vectors were captured with
        assert b.num_points == g["d__xyz"].shape[0]
        n_fix = info["kept"] + info["cloned"]
        for k in PARAM_NAMES:
            rows = slice(0, n_fix) if k == "_xyz" else slice(None)
            np.testing.assert_allclose(b.params[k].detach().cpu().numpy()[rows], g["d_" + k][rows], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(b.optimizer.state[b.params[k]]["exp_avg"].cpu().numpy(), g["dm_" + k], rtol=1e-6,
                                       atol=1e-12)
    # the optimizer keeps working on the new tensors
    for k in PARAM_NAMES:
        b.params[k].grad = torch.ones_like(b.params[k])
    b.optimizer.step()
    assert all(torch.isfinite(b.params[k]).all() for k in PARAM_NAMES)


def test_device_densify_nothing_selected_and_everything_pruned():
    from fsgs_amd import synth
    from fsgs_amd.model import PARAM_NAMES, GaussianCloud

    cam = synth.make_camera(64, 48)
    sc = synth.init_scene(64, 48, 200, seed=0)
    pc = GaussianCloud(sc, sh_degree=3, device=DEV, scene_radius=1.0)
    pc.training_setup(fused=True)
    P = pc.num_points
    # never seen (denom = 0 -> NaN gradients): nothing is cloned or split; opacity 0.1 >= 0.05: nothing pruned
    before = {k: pc.params[k].detach().clone() for k in PARAM_NAMES}
    info = pc.densify_and_prune_device(2e-4, 0.05, None)
    assert info == {"kept": P, "cloned": 0, "split": 0, "children_kept": 0}
    assert all(torch.equal(before[k], pc.params[k].detach()) for k in PARAM_NAMES)
    # min_opacity above every opacity: the cloud becomes empty without crashing
    info = pc.densify_and_prune_device(2e-4, 0.5, None)
    assert pc.num_points == 0 and info["kept"] == 0
