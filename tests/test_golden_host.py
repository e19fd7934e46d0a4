"""The host-side restatements (torch statements of the reference's glue maths) against golden
vectors captured from the reference's own Python (tests/golden/make_golden.py).
These statements are the numerics references of the fused HIP kernels' GPU tests."""
import os

import numpy as np
import pytest
import torch

from fsgs_amd import flow, losses, model, pose, sh

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = lambda a: torch.tensor(np.asarray(a))


def _load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_eval_sh_and_its_gradients(deg):
    g = _load("eval_sh.npz")
    s = T(g["sh"]).requires_grad_(True)
    d = T(g["dirs"]).requires_grad_(True)
    rgb = torch.clamp_min(sh.eval_sh(deg, s, d) + 0.5, 0.0)
    (rgb * T(g["w"])).sum().backward()
    np.testing.assert_allclose(rgb.detach().numpy(), g[f"rgb{deg}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s.grad.numpy(), g[f"dsh{deg}"], rtol=1e-5, atol=1e-6)
    if deg > 0:
        np.testing.assert_allclose(d.grad.numpy(), g[f"ddir{deg}"], rtol=1e-4, atol=2e-5)


def test_rgb_loss_ssim_l1():
    g = _load("rgb_loss.npz")
    gt = T(g["gt"])
    for tag, m in (("nomask", None), ("mask", T(g["mask"]))):
        x = T(g["img"]).requires_grad_(True)
        l = losses.rgb_loss_torch(x, gt, mask=m)
        l.backward()
        np.testing.assert_allclose(l.item(), g[f"loss_{tag}"], rtol=1e-5)
        np.testing.assert_allclose(x.grad.numpy(), g[f"grad_{tag}"], rtol=1e-4, atol=1e-8)
    x = T(g["img"]).requires_grad_(True)
    s = losses.ssim_torch(x, gt)
    s.backward()
    np.testing.assert_allclose(s.item(), g["ssim"], rtol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["ssim_grad"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(losses.l1_loss(T(g["img"]), gt).item(), g["l1"], rtol=1e-6)


def test_pearson_and_local_pearson():
    g = _load("pearson.npz")
    src = T(g["src"])
    x = T(g["tgt"]).requires_grad_(True)
    l = losses.pearson_torch(src, x)
    l.backward()
    np.testing.assert_allclose(l.item(), g["pearson"], rtol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["pearson_grad_tgt"], rtol=1e-4, atol=1e-9)
    x = T(g["src"]).requires_grad_(True)
    losses.pearson_torch(x, T(g["tgt"])).backward()
    np.testing.assert_allclose(x.grad.numpy(), g["pearson_grad_src"], rtol=1e-4, atol=1e-9)
    # local: same corners the reference drew; also our draw consumes the RNG identically
    x = T(g["tgt"]).requires_grad_(True)
    corners = (T(g["lp_x0"]), T(g["lp_y0"]))
    l = losses.local_pearson_torch(src, x, int(g["lp_box"]), float(g["lp_p"]), corners)
    l.backward()
    np.testing.assert_allclose(l.item(), g["lp_loss"], rtol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["lp_grad_tgt"], rtol=1e-4, atol=1e-9)
    torch.manual_seed(3)
    x0, y0 = losses.draw_patch_corners(src.shape[0], src.shape[1], int(g["lp_box"]), float(g["lp_p"]), "cpu")
    np.testing.assert_array_equal(x0.numpy(), g["lp_x0"])
    np.testing.assert_array_equal(y0.numpy(), g["lp_y0"])


def test_learnpose_forward_and_gradients():
    g = _load("pose_glue.npz")
    for cam in range(g["r"].shape[2]):
        r = T(g["r"]).requires_grad_(True)
        t = T(g["t"]).requires_grad_(True)
        w2c = pose.pose_to_w2c(r, t, cam)
        (w2c * T(g["wsum"])).sum().backward()
        np.testing.assert_allclose(w2c.detach().numpy(), g[f"w2c_{cam}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r.grad.numpy(), g[f"dr_{cam}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(t.grad.numpy(), g[f"dt_{cam}"], rtol=1e-5, atol=1e-7)


def test_transform_to_frame_detach_semantics():
    g = _load("pose_glue.npz")
    for gg, cg in ((True, True), (True, False), (False, True)):
        a = T(g["ttf_xyz"]).requires_grad_(True)
        m = T(g["ttf_w2c"]).requires_grad_(True)
        y = pose.transform_to_frame(a, m, gg, cg)
        (y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
        key = f"{int(gg)}{int(cg)}"
        np.testing.assert_allclose(y.detach().numpy(), g["ttf_" + key], rtol=1e-5, atol=1e-6)
        dx = a.grad.numpy() if a.grad is not None else np.zeros((64, 3), np.float32)
        dm = m.grad.numpy() if m.grad is not None else np.zeros((4, 4), np.float32)
        np.testing.assert_allclose(dx, g["ttf_dx_" + key], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dm[:3], g["ttf_dm_" + key][:3], rtol=1e-4, atol=1e-5)


def test_small_helpers():
    g = _load("pose_glue.npz")
    from fsgs_amd import synth

    for q, R in zip(g["br_q"], g["br_R"]):
        np.testing.assert_allclose(synth.quat_to_rot(q), R, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(model.inverse_sigmoid(T(g["inv_sigmoid_in"])).numpy(), g["inv_sigmoid_out"], rtol=1e-6)
    lr = [model.expon_lr(int(s), 1.6e-4 * 5, 1.6e-6 * 5, 30000) for s in g["lr_steps"]]
    np.testing.assert_allclose(lr, g["lr_vals"], rtol=1e-12)


def test_projection_flow_loss():
    g = _load("flow_loss.npz")
    for tag, rm in (("rigid", T(g["rigid"])), ("norigid", None)):
        r = T(g["q"]).reshape(1, 4, 1).clone().requires_grad_(True)
        t = T(g["t"]).reshape(3, 1).clone().requires_grad_(True)
        w2c = pose.pose_to_w2c(r, t, 0)
        w2c.retain_grad()
        l = flow.projection_flow_loss_torch(T(g["depth_prev"]), g["w2c_prev"], w2c, g["K"], T(g["flow"])[0], rm)
        l.backward()
        np.testing.assert_allclose(l.item(), g[f"loss_{tag}"], rtol=1e-5)
        np.testing.assert_allclose(w2c.grad.numpy()[:3], g[f"dw2c_{tag}"][:3], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(r.grad.numpy(), g[f"dr_{tag}"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(t.grad.numpy(), g[f"dt_{tag}"], rtol=1e-3, atol=1e-4)


def test_depth_silhouette_pseudo_colours_use_the_stored_matrix_rows():
    g = _load("depth_sil.npz")
    pts = T(g["pts"])
    V = T(g["viewmatrix_stored"])
    z = (pts @ V[2, :3].reshape(3, 1) + V[2, 3]).reshape(-1)
    np.testing.assert_allclose(z.numpy(), g["ds_stored"][:, 0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose((z * z).numpy(), g["ds_stored"][:, 2], rtol=1e-5, atol=1e-6)
    assert (g["ds_stored"][:, 1] == 1).all()
    np.testing.assert_allclose(pts[:, 2].numpy(), g["ds_identity"][:, 0], rtol=1e-6)


def test_pose_metrics_match_reference_align_pose():
    from fsgs_amd import metrics

    g = _load("pose_metrics.npz")
    for case in range(3):
        aligned, m = metrics.pose_metrics(g[f"pred_{case}"], g[f"gt_{case}"])
        np.testing.assert_allclose(aligned, g[f"aligned_{case}"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(m, g[f"metrics_{case}"], rtol=2e-3, atol=1e-5)


def test_densify_prune_and_opacity_reset_match_reference():
    """same inputs, same torch RNG seed -> identical clone / split / prune result and Adam-state surgery
    (scene/gaussian_model.py:501-676)."""
    from fsgs_amd.model import PARAM_NAMES, GaussianCloud

    g = _load("densify.npz")
    pc = GaussianCloud({k: g["p_" + k] for k in PARAM_NAMES}, device="cpu", scene_radius=float(g["var_scene_radius"]))
    pc.training_setup(fused=False)
    for grp in pc.optimizer.param_groups:  # install the recorded Adam moments
        p = grp["params"][0]
        pc.optimizer.state[p] = {"step": torch.tensor(1.0), "exp_avg": T(g["m_" + grp["name"]]).clone(),
                                 "exp_avg_sq": T(g["v_" + grp["name"]]).clone()}
    pc.variables["max_radii2D"] = T(g["var_max_radii2D"]).clone()
    pc.variables["xyz_gradient_accum"] = T(g["var_xyz_gradient_accum"]).clone()
    pc.variables["denom"] = T(g["var_denom"]).clone()
    torch.manual_seed(11)
    pc.densify_and_prune(2e-4, 0.05, 20)
    assert pc.num_points == g["d__xyz"].shape[0] and pc.num_points != int(g["P"])
    for k in PARAM_NAMES:
        np.testing.assert_allclose(pc.params[k].detach().numpy(), g["d_" + k], rtol=1e-6, atol=1e-7)
        st = pc.optimizer.state[pc.params[k]]
        np.testing.assert_allclose(st["exp_avg"].numpy(), g["dm_" + k], rtol=1e-6, atol=1e-12)
    for k in ("max_radii2D", "xyz_gradient_accum", "denom"):
        np.testing.assert_array_equal(pc.variables[k].numpy(), g["dvar_" + k])
    pc.reset_opacity()
    np.testing.assert_allclose(pc.params["_opacity"].detach().numpy(), g["r_opacity"], rtol=1e-6)
    assert not np.any(pc.optimizer.state[pc.params["_opacity"]]["exp_avg"].numpy())


def test_fundamental_matrix_and_sampson_statement_known_answers():
    """fsgs_amd/epipolar.py restates kornia's essential_from_Rt / fundamental_from_essential /
    sampson_epipolar_distance (kornia is not in the reference tree).  Known answers: exact two-view correspondences
    satisfy x2^T F x1 = 0 and have zero Sampson distance; a pixel displaced by d perpendicular to its epipolar
    line has squared distance ~ d^2 / 2 (both images share the error); the reference's `dist < mask` quirk."""
    from fsgs_amd import epipolar, synth

    H, W = 48, 64
    K = synth.intrinsics(W, H).astype(np.float64)
    w1 = synth.pose_matrix((1, 0.01, -0.02, 0.005), (0.01, 0.02, -0.01)).astype(np.float64)
    w2 = synth.pose_matrix((1, -0.02, 0.01, 0.02), (0.05, -0.03, 0.02)).astype(np.float64)
    F = epipolar.fundamental_from_w2c(w1, w2, K).astype(np.float64)
    rng = np.random.default_rng(0)
    Xw = np.concatenate([rng.uniform(-0.3, 0.3, (50, 2)), rng.uniform(0.8, 1.5, (50, 1)), np.ones((50, 1))], 1)
    p1 = (K @ (w1 @ Xw.T)[:3]).T
    p2 = (K @ (w2 @ Xw.T)[:3]).T
    x1 = p1 / p1[:, 2:]
    x2 = p2 / p2[:, 2:]
    resid = np.einsum("ni,ij,nj->n", x2, F, x1)
    scale = np.linalg.norm(F) * np.linalg.norm(x1, axis=1) * np.linalg.norm(x2, axis=1)
    assert np.max(np.abs(resid) / scale) < 1e-5
    # dense statement on a pixel grid: flow from exact geometry -> ~0; plus a perpendicular offset d -> d^2/2
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    z = 1.0 + 0.2 * np.sin(u / 9.0) * np.cos(v / 7.0)
    cam1 = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z, np.ones_like(z)], 0).reshape(4, -1)
    q = K @ (w2 @ np.linalg.inv(w1) @ cam1)[:3]
    flow = np.stack([q[0] / q[2] - u.reshape(-1), q[1] / q[2] - v.reshape(-1)], 0).reshape(2, H, W)
    d0 = epipolar.sampson_distance_torch(T(flow.astype(np.float32)), F).numpy()
    assert d0.max() < 1e-3
    l = (F @ np.stack([u.reshape(-1), v.reshape(-1), np.ones(H * W)], 0))[:2]  # epipolar line normals in image 2
    n = l / np.linalg.norm(l, axis=0, keepdims=True)
    d = 0.8
    flow_off = flow + (d * n).reshape(2, H, W)
    d1 = epipolar.sampson_distance_torch(T(flow_off.astype(np.float32)), F).numpy()
    np.testing.assert_allclose(d1, np.full_like(d1, d * d / 2), rtol=0.15)
    # dist < (dist <= thr): rigid needs BOTH dist <= thr and dist < 1
    dist = torch.tensor([[0.1, 0.5, 0.99, 1.0, 1.5, 30.0]])
    thr = dist.mean().item() + 2.0 * dist.std().item()
    m = epipolar.rigid_mask_torch(dist, 2.0)
    assert thr > 1.5 and m.tolist() == [[True, True, True, False, False, False]]
