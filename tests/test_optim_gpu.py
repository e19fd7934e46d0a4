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
