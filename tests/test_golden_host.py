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


def _write_golden_dataset(root, d):
    from fsgs_amd import dataset

    dataset.write_sequence(root, d["colors_u8"], d["disparity"], d["flows_fw_in"], d["flows_bw_in"], d["cam_poses"],
                           d["KL"], scene="1", data=[str(r) for r in d["runs"]])


@pytest.mark.parametrize("tag,fs,fe", [("all", 0, -1), ("slice", 2, 9)])
def test_sequence_reader_matches_reference_record_data(tmp_path, tag, fs, fe):
    """the on-disk layout: what PoseModel.__init__ made of the same files (scene/pose_optimizer.py:355-460)."""
    from fsgs_amd import dataset

    d = _load("dataset.npz")
    _write_golden_dataset(str(tmp_path), d)
    fr = dataset.read_sequence(str(tmp_path), frame_start=fs, frame_end=fe, device="cpu")
    n = int(d[tag + "_num_cams"])
    assert len(fr.colors) == n == len(fr.monodeps) and len(fr.flows_fw) == n - 1
    np.testing.assert_array_equal(torch.stack(fr.colors).numpy(), d[tag + "_colors"])       # u8 / 255 in fp32
    np.testing.assert_array_equal(torch.stack(fr.flows_fw).numpy(), d[tag + "_flows_fw"])
    np.testing.assert_array_equal(torch.stack(fr.flows_bw).numpy(), d[tag + "_flows_bw"])
    np.testing.assert_array_equal(torch.stack(fr.monodeps).numpy(), d[tag + "_monodeps"])   # float64 maths, then fp32
    np.testing.assert_allclose(fr.K, d[tag + "_intrinsic"], rtol=1e-7)
    np.testing.assert_array_equal(fr.i_test, d[tag + "_i_test"])
    np.testing.assert_array_equal(fr.i_train, d[tag + "_i_train"])
    np.testing.assert_array_equal(fr.data_ind, d[tag + "_data_ind"])
    np.testing.assert_allclose(fr.weights, d[tag + "_weights"])
    for k, v in fr.gt_poses.items():
        np.testing.assert_array_equal(v, d["%s_gt_%s" % (tag, k)])
    cam = dataset.camera_from_frames(fr)
    fov = d[tag + "_fov"]
    np.testing.assert_allclose([cam["tanfovx"], cam["tanfovy"]], np.tan(fov * 0.5), rtol=1e-6)
    assert fr.monodeps[0].min() == 0.5 and fr.monodeps[0].max() == 1.5


def test_sequence_reader_errors(tmp_path):
    from fsgs_amd import dataset

    with pytest.raises(FileNotFoundError):
        dataset.read_sequence(str(tmp_path), device="cpu")
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / "badname.png").write_bytes(b"")
    with pytest.raises(ValueError):
        dataset.read_sequence(str(tmp_path), device="cpu")


def test_checkpoints_written_by_the_reference_restore(tmp_path):
    """chkpnt7.pth / poses7.pth as GaussianModel.capture() / PoseModel.capture() wrote them (train.py:371-376)."""
    import shutil

    from fsgs_amd import checkpoint
    from fsgs_amd.trainer import PoseTrack

    for f in ("ref_chkpnt7.pth", "ref_poses7.pth"):
        shutil.copy(os.path.join(G, f), tmp_path / f.replace("ref_", ""))
    ref, it = torch.load(tmp_path / "chkpnt7.pth", weights_only=False)
    P = ref[1].shape[0]
    pc = model.GaussianCloud({k: np.zeros((1,) + s, np.float32) for k, s in
                              (("_xyz", (3,)), ("_features_dc", (1, 3)), ("_features_rest", (15, 3)), ("_opacity", (1,)),
                               ("_scaling", (3,)), ("_rotation", (4,)))}, device="cpu")
    poses = PoseTrack(2, device="cpu")
    it2, intrinsic = checkpoint.load(str(tmp_path / "chkpnt7.pth"), pc, poses, fused=False)
    assert it == it2 == 7 and pc.num_points == P and pc.active_sh_degree == 2 and pc.spatial_lr_scale == 5.0
    order = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
    for k, t in zip(order, ref[1:7]):
        assert torch.equal(pc.params[k], t.detach()) and pc.params[k].requires_grad and pc.params[k].is_leaf
    assert torch.equal(pc.variables["max_radii2D"], ref[7])
    assert not pc.variables["denom"].any() and not pc.variables["xyz_gradient_accum"].any()  # upstream's restore quirk
    sd = ref[10]
    for gi, g in enumerate(pc.optimizer.param_groups):
        assert g["name"] == sd["param_groups"][gi]["name"] and g["lr"] == pytest.approx(sd["param_groups"][gi]["lr"])
        st = pc.optimizer.state[g["params"][0]]
        assert int(st["step"]) == 3
        assert torch.equal(st["exp_avg"], sd["state"][gi]["exp_avg"])
        assert torch.equal(st["exp_avg_sq"], sd["state"][gi]["exp_avg_sq"])
    # poses: 11 cameras, every second pred_w2c filled by LearnPose.forward
    pref, _ = torch.load(tmp_path / "poses7.pth", weights_only=False)
    assert torch.equal(poses.r, pref[1].detach()) and torch.equal(poses.t, pref[2].detach())
    assert poses.optimizer is None and intrinsic.shape == (3, 3)
    for i in range(poses.r.shape[-1]):
        if i % 2 == 0:
            np.testing.assert_allclose(poses.pred_w2c[i].numpy(), pref[3][i], rtol=1e-6)
            np.testing.assert_allclose(pose.pose_to_w2c(poses.r, poses.t, i).detach().numpy(), pref[3][i], atol=1e-6)
        else:
            assert poses.pred_w2c[i] is None
    # and the step after a restore continues the reference's Adam trajectory
    torch.manual_seed(0)
    grads = {k: torch.randn_like(pc.params[k]) * 1e-2 for k in order}
    ref_opt = torch.optim.Adam([{"params": [ref[1 + order.index(g["name"])].detach().clone().requires_grad_(True)],
                                 "lr": g["lr"], "name": g["name"]} for g in sd["param_groups"]], lr=0.0, eps=1e-15)
    ref_opt.load_state_dict(sd)
    for g in ref_opt.param_groups:
        g["params"][0].grad = grads[g["name"]].clone()
    ref_opt.step()
    for k in order:
        pc.params[k].grad = grads[k].clone()
    pc.optimizer.step()
    for g in ref_opt.param_groups:
        assert torch.equal(pc.params[g["name"]], g["params"][0])


def test_checkpoint_round_trip_and_bad_tuples(tmp_path):
    from fsgs_amd import checkpoint
    from fsgs_amd.trainer import PoseTrack

    torch.manual_seed(3)
    P = 17
    mk = lambda: model.GaussianCloud({"_xyz": torch.randn(P, 3), "_features_dc": torch.randn(P, 1, 3),
                                      "_features_rest": torch.randn(P, 15, 3), "_opacity": torch.randn(P, 1),
                                      "_scaling": torch.randn(P, 3), "_rotation": torch.randn(P, 4)}, device="cpu")
    pc = mk()
    with pytest.raises(RuntimeError):
        checkpoint.capture_gaussians(pc)
    pc.training_setup(fused=False)
    pc.active_sh_degree = 1
    for k in pc.params:
        pc.params[k].grad = torch.randn_like(pc.params[k])
    pc.optimizer.step()
    # a FusedAdam-style python-int step must come out as the tensor torch's Adam expects
    pc.optimizer.state[pc.params["_xyz"]]["step"] = 1
    poses = PoseTrack(4, device="cpu")
    poses.set_pose(2, [0.9, 0.1, -0.2, 0.05], [0.1, 0.2, 0.3])
    with torch.no_grad():
        poses.get_pose(2)
    checkpoint.save(str(tmp_path), 42, pc, poses, np.eye(3) * 2)
    sd = torch.load(tmp_path / "chkpnt42.pth", weights_only=False)[0][10]
    assert all(torch.is_tensor(s["step"]) and s["step"].dtype == torch.float32 for s in sd["state"].values())
    pc2, poses2 = mk(), PoseTrack(1, device="cpu")
    it, K = checkpoint.load(str(tmp_path / "chkpnt42.pth"), pc2, poses2, fused=False)
    assert it == 42 and np.array_equal(K, np.eye(3) * 2) and pc2.active_sh_degree == 1
    for k in pc.params:
        assert torch.equal(pc.params[k], pc2.params[k])
        assert torch.equal(pc.optimizer.state[pc.params[k]]["exp_avg"], pc2.optimizer.state[pc2.params[k]]["exp_avg"])
    assert torch.equal(poses.r, poses2.r) and torch.equal(poses2.pred_w2c[2], poses.pred_w2c[2])
    assert poses2.pred_w2c[0] is None
    # an in-memory capture restored into another model is a snapshot, not a view of the live tensors
    pc3 = mk()
    checkpoint.restore_gaussians(pc3, checkpoint.capture_gaussians(pc), fused=False)
    assert pc3.params["_xyz"].data_ptr() != pc.params["_xyz"].data_ptr()
    m1, m3 = pc.optimizer.state[pc.params["_xyz"]]["exp_avg"], pc3.optimizer.state[pc3.params["_xyz"]]["exp_avg"]
    assert m1.data_ptr() != m3.data_ptr() and torch.equal(m1, m3)
    with pytest.raises(ValueError):
        checkpoint.restore_gaussians(pc2, (1, 2, 3))
    with pytest.raises(ValueError):
        checkpoint.restore_poses(poses2, (None, torch.zeros(4, 3), torch.zeros(3, 3), np.zeros((3, 4, 4)), np.eye(3)))


def test_fused_adam_checkpoint_steps_under_torch_adam():
    """A capture taken from the HIP FusedAdam (param_groups hold lr / betas / eps only, `step` is a python int) must be
    steppable by torch.optim.Adam after restore(fused=False): torch adopts the saved groups verbatim and reads
    weight_decay, amsgrad, maximize ... from them."""
    from fsgs_amd import checkpoint, optim

    torch.manual_seed(5)
    P = 9
    mk = lambda: model.GaussianCloud({"_xyz": torch.randn(P, 3), "_features_dc": torch.randn(P, 1, 3),
                                      "_features_rest": torch.randn(P, 15, 3), "_opacity": torch.randn(P, 1),
                                      "_scaling": torch.randn(P, 3), "_rotation": torch.randn(P, 4)}, device="cpu")
    pc = mk()
    pc.training_setup(fused=True)
    assert isinstance(pc.optimizer, optim.FusedAdam)
    for g in pc.optimizer.param_groups:  # the state FusedAdam.step() would have left (the kernel itself needs the GPU)
        p = g["params"][0]
        pc.optimizer.state[p] = {"step": 3, "exp_avg": torch.randn_like(p), "exp_avg_sq": torch.rand_like(p)}
    cap = checkpoint.capture_gaussians(pc)
    for g in cap[10]["param_groups"]:
        assert g["weight_decay"] == 0 and g["amsgrad"] is False and g["maximize"] is False
    pc2 = mk()
    checkpoint.restore_gaussians(pc2, cap, fused=False)
    assert isinstance(pc2.optimizer, torch.optim.Adam)
    before = {k: pc2.params[k].detach().clone() for k in pc2.params}
    for k in pc2.params:
        pc2.params[k].grad = torch.randn_like(pc2.params[k])
    pc2.optimizer.step()  # KeyError: 'weight_decay' before the fix
    assert all(not torch.equal(before[k], pc2.params[k]) for k in pc2.params)
    st = pc2.optimizer.state[pc2.params["_xyz"]]
    assert float(st["step"]) == 4.0
    # and the reverse direction keeps working: torch's groups (with the extra keys) load into FusedAdam
    pc3 = mk()
    checkpoint.restore_gaussians(pc3, checkpoint.capture_gaussians(pc2), fused=True)
    assert isinstance(pc3.optimizer, optim.FusedAdam)


def test_eval_pose_aligns_every_run_on_its_own_and_global_run_starts_at_iteration_zero():
    """train.py:492-506: one Sim(3) alignment per <data> run, metrics summed with the runs' weights -- two runs that are
    each a perfect copy of their ground truth up to a DIFFERENT similarity must score ~0 (one joint alignment cannot).
    train.py:381: global_run covers range(first_iter = 0, iterations + 1)."""
    from fsgs_amd import synth
    from fsgs_amd.sequence import _rot_to_quat, gt_trajectory
    from fsgs_amd.trainer import FrameData, PoseTrack, Runner

    gt = [np.asarray(m, np.float32) for m in gt_trajectory(8, step_t=0.02, step_r=0.01, seed=3)]
    runs = {"1_5": np.stack(gt[:4]), "1_6": np.stack(gt[4:])}
    frames = FrameData([torch.zeros(3, 4, 4)] * 8, [torch.zeros(4, 4)] * 8, gt_w2c=gt)
    frames.gt_poses, frames.data_ind, frames.weights = runs, [0, 4, 8], [0.5, 0.5]
    poses = PoseTrack(8, "cpu")
    sims = [(1.0, synth.pose_matrix((1, 0, 0, 0), (0, 0, 0))), (2.5, synth.pose_matrix((0.9, 0.1, -0.2, 0.3), (0.4, -0.2, 0.1)))]
    for i, m in enumerate(gt):
        s, S = sims[i // 4]
        p = np.eye(4)
        p[:3, :3] = S[:3, :3] @ m[:3, :3]
        p[:3, 3] = s * (S[:3, :3] @ m[:3, 3]) + S[:3, 3]  # a similarity of the whole run
        poses.set_pose(i, _rot_to_quat(p[:3, :3]), p[:3, 3])
    pc = model.GaussianCloud({"_xyz": torch.randn(5, 3), "_features_dc": torch.randn(5, 1, 3),
                              "_features_rest": torch.randn(5, 15, 3), "_opacity": torch.randn(5, 1),
                              "_scaling": torch.randn(5, 3), "_rotation": torch.randn(5, 4)}, device="cpu")
    run = Runner(pc, poses, frames, fused=False)
    for i in range(8):  # what a run leaves behind: every frame's last get_pose recorded (scene/pose_optimizer.py:635-638)
        poses.get_pose(i)
    rpe_t, rpe_r, ate = run.eval_pose()
    assert ate < 1e-4 and rpe_r < 0.3, (rpe_t, rpe_r, ate)
    # train.py:499-500 evaluates the RECORD, not the parameters: a pose moved after its last get_pose (a test frame's last
    # tracking step) does not show, and the evaluation itself records nothing; test_frame_quirks=False reads the
    # parameters as they stand -- without touching the record either
    recorded = [w.clone() for w in poses.pred_w2c]
    poses.set_pose(4, (1, 0, 0, 0), (5.0, 5.0, 5.0))
    assert run.eval_pose()[2] == ate
    run.test_frame_quirks = False
    assert run.eval_pose()[2] > 10 * max(ate, 1e-6)
    assert all(torch.equal(a, b) for a, b in zip(recorded, poses.pred_w2c))
    run.test_frame_quirks = True
    del frames.gt_poses  # one joint alignment of the two differently-scaled runs cannot fit both
    assert run.eval_pose()[2] > 1e-2
    # the iterations of the global phase
    seen = []
    run.mapping = lambda ts, n, progressive, want_pkg=True: seen.append(ts)
    pc.training_setup(fused=False)
    pc.initialize_optimizer = lambda fused=True: None
    deg0 = pc.active_sh_degree
    run.global_run(3, eval_every=0)  # (the periodic test-frame evaluation renders: GPU only, tests/test_harness_gpu.py)
    assert len(seen) == 4 and pc.active_sh_degree == deg0 + 1  # iterations 0..3; 0 % 1000 == 0 raises the SH degree


def test_validation_ssim_is_the_uniform_window_definition():
    """metrics.ssim = the metric rgb_evaluation prints (skimage defaults: 7x7 uniform window, sample covariance, cropped
    border), pinned by closed forms: identical images -> 1; a flat image against the same image plus a constant d ->
    luminance term only; and against a direct per-window evaluation at one pixel of a random pair."""
    from fsgs_amd import metrics

    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (2, 3, 24, 31))
    assert abs(metrics.ssim(a, a) - 1.0) < 1e-12
    m, d = 0.4, 0.25
    flat = np.full((1, 3, 20, 20), m)
    C1 = 0.01 ** 2
    want = (2 * m * (m + d) + C1) / (m * m + (m + d) ** 2 + C1)  # variances and covariance are zero: the C2 factors cancel
    assert abs(metrics.ssim(flat, flat + d) - want) < 1e-12
    b = np.clip(a + rng.normal(0, 0.1, a.shape), 0, 1)
    # one image, one channel, window centred at (10, 12): direct evaluation of the definition
    x, y = a[0, 1, 7:14, 9:16], b[0, 1, 7:14, 9:16]
    ux, uy = x.mean(), y.mean()
    vx, vy, vxy = x.var(ddof=1), y.var(ddof=1), ((x - ux) * (y - uy)).sum() / 48.0
    s_direct = ((2 * ux * uy + C1) * (2 * vxy + 0.03 ** 2)) / ((ux * ux + uy * uy + C1) * (vx + vy + 0.03 ** 2))
    single = np.zeros((1, 1, 7, 7))
    got = metrics.ssim(a[:1, 1:2, 7:14, 9:16], b[:1, 1:2, 7:14, 9:16])  # a 7x7 image: exactly one uncropped window
    assert abs(got - s_direct) < 1e-9, (got, s_direct)
    assert 0.0 < metrics.ssim(a, b) < 1.0


@pytest.mark.parametrize("tag", ["c2_identity", "c1_posed", "vis_posed"])
def test_setup_camera_settings_match_the_reference(tag):
    """PoseModel.setup_camera (scene/pose_optimizer.py:600-633) -> the 12 GaussianRasterizationSettings fields, captured
    from the imported reference (tests/golden/camera.npz), against the three places the build states them:
    synth.make_camera, dataset.camera_from_frames and trainer.settings_from_cam.  tanfov, the transposed view matrix
    and projmatrix = w2c^T . opengl_proj^T (an fp32 product in the reference) bit for bit; campos = inverse(w2c)[:3,3]
    (an fp32 LAPACK / cuSOLVER inverse there, never read by the rasteriser when colours are precomputed) to an ulp."""
    from fsgs_amd import dataset, synth
    from fsgs_amd.trainer import settings_from_cam

    g = _load("camera.npz")
    f = lambda k: g["%s_%s" % (tag, k)]
    W, H = int(f("image_width")), int(f("image_height"))
    near, far = f("near_far")
    cams = [synth.make_camera(W, H, w2c=f("w2c"), K=f("K"), near=near, far=far)]
    if tag == "c2_identity":  # the route a loaded sequence takes (identity raster camera, default planes)
        class Frames:
            pass

        fr = Frames()
        fr.W, fr.H, fr.K = W, H, f("K")
        cams.append(dataset.camera_from_frames(fr))
    for cam in cams:
        assert cam["image_height"] == H and cam["image_width"] == W
        assert float(cam["tanfovx"]) == float(f("tanfovx")) and float(cam["tanfovy"]) == float(f("tanfovy"))
        assert np.array_equal(cam["viewmatrix"], f("viewmatrix").reshape(4, 4))
        assert np.array_equal(cam["projmatrix"], f("projmatrix").reshape(4, 4))
        assert cam["viewmatrix"].dtype == np.float32 and cam["projmatrix"].dtype == np.float32
        np.testing.assert_allclose(cam["campos"], f("campos"), rtol=0, atol=2e-7)
        np.testing.assert_allclose(cam["campos"], f("cam_center"), rtol=0, atol=2e-7)
        assert np.array_equal(cam["bg"], f("bg")) and float(cam["scale_modifier"]) == float(f("scale_modifier"))
        assert int(cam["sh_degree"]) == int(f("sh_degree")) and bool(cam["prefiltered"]) == bool(f("prefiltered"))
        assert bool(cam["debug"]) == bool(f("debug"))
        s = settings_from_cam(cam, "cpu")
        assert s.viewmatrix.shape == (1, 4, 4) == tuple(f("viewmatrix").shape) and s.projmatrix.shape == (1, 4, 4)
        assert np.array_equal(s.viewmatrix.numpy(), f("viewmatrix")) and np.array_equal(s.projmatrix.numpy(), f("projmatrix"))
        assert s.tanfovx == float(f("tanfovx")) and s.tanfovy == float(f("tanfovy")) and s.image_height == H
        assert np.array_equal(s.bg.numpy(), f("bg")) and s.scale_modifier == 1.0 and s.sh_degree == 0
        assert s.prefiltered is False and s.debug is False
    if tag == "c2_identity":
        assert np.array_equal(cams[0]["campos"], np.zeros(3, np.float32))


def test_covariance_matches_the_reference_and_the_oracle_follows_it(oracle32):
    """Sigma = R S S^T R^T as the reference's own Python states it (build_covariance_from_scaling_rotation,
    scene/gaussian_model.py:32-36; strip_symmetric order xx, xy, xz, yy, yz, zz) -- the one in-tree statement of the
    rasteriser's cov3D -- against the oracle's preprocess (raster_oracle.c cov3d_from_scale_rot; fed the
    normalised quaternion, as get_rotation hands it over) for scale_modifier 1 and 1.7."""
    from fsgs_amd import synth

    g = _load("covariance.npz")
    n = len(g["scaling"])
    cam = synth.make_camera(64, 48)
    xyz = np.zeros((n, 3), np.float32)
    xyz[:, 2] = 1.0
    col = np.zeros((n, 3), np.float32)
    for key, mod in (("cov6", 1.0), ("cov6_modifier_1p7", 1.7)):
        c = dict(cam, scale_modifier=mod)
        st = oracle32.raster_forward(c, xyz, col, np.full(n, 0.5, np.float32), g["scaling"], g["rotation_normalised"])[3]
        want = g[key]
        np.testing.assert_allclose(st.cov3D(), want, rtol=2e-5, atol=2e-6 * float(np.abs(want).max()))
    # the reference normalises inside build_rotation: raw and normalised quaternions give the same Sigma there
    q = g["rotation_raw"].astype(np.float64)
    assert np.allclose(q / np.linalg.norm(q, axis=1, keepdims=True), g["rotation_normalised"], atol=1e-6)


def test_pose_lr_table_is_the_multisteplr_sequence_of_the_reference():
    """PoseTrack.begin_frame replays the learning rates of Adam(lr .01) + MultiStepLR(milestones 0, s, 2s, ..; gamma .5) from a
    table recorded once (scene/pose_optimizer.py:489-496): the table must be what a freshly built pair shows at construction
    and after every scheduler.step() -- for the reference's 50 iterations and for the pinned test's 5."""
    import warnings

    from fsgs_amd.trainer import PoseTrack

    for n_it in (50, 5, 30):
        table = PoseTrack._lr_table(n_it)
        p = PoseTrack(2, "cpu")
        p.initialize_tracking_optimizer(n_it)  # the reference's construction, on torch.optim.Adam here (CPU)
        seq = [tuple(float(g["lr"]) for g in p.optimizer.param_groups)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(n_it):
                p.scheduler.step()
                seq.append(tuple(float(g["lr"]) for g in p.optimizer.param_groups))
        assert table[:n_it + 1] == seq, n_it
        # milestone 0 halves the rate at construction: the first iteration already runs at 0.005 (train.py:189,194)
        assert table[0] == (0.005, 0.005)
    sch = PoseTrack._TableScheduler(torch.optim.SGD([{"params": [torch.zeros(1, requires_grad=True)], "lr": 1.0},
                                                     {"params": [torch.zeros(1, requires_grad=True)], "lr": 1.0}], lr=1.0),
                                    PoseTrack._lr_table(5))
    lrs = [sch.optimizer.param_groups[0]["lr"]]
    for _ in range(7):  # past the end of the table: the last rate stays
        sch.step()
        lrs.append(sch.optimizer.param_groups[0]["lr"])
    assert lrs[-1] == lrs[5] and lrs[0] == 0.005
    sch.rewind()
    assert sch.optimizer.param_groups[1]["lr"] == 0.005 and sch.pos == 0


def test_comm_model_of_the_bench_line():
    """bench.comm_model (the prediction a scaling record is held against, DESIGN s6): same wire term for both routes, the
    ring pays 2 (N - 1) dependent hops, the direct form two launches; efficiency falls with the exchange time."""
    import bench

    nbytes = 300_000 * 14 * 4
    prev = None
    for n in (2, 4, 8):
        m = bench.comm_model(nbytes, n, 0.62)  # on top of a measured one-rank step (here: round 5's C2 step, ms)
        wire = 2.0 * (n - 1) / n * nbytes / ((n - 1) * 64e9) * 1e3
        assert abs(m["wire_ms"] - wire) < 1e-12 and m["rccl_ms"] > wire and m["direct_ms"] > wire
        assert abs((m["rccl_ms"] - m["wire_ms"]) - (0.020 + 2 * (n - 1) * 0.005)) < 1e-12
        assert 0.5 < m["weak_scaling_efficiency"] < 1.0
        assert abs(m["weak_scaling_efficiency"] - 0.62 / (0.62 + min(m["rccl_ms"], m["direct_ms"]))) < 1e-12
        if prev is not None:
            assert m["wire_ms"] < prev["wire_ms"] and m["weak_scaling_efficiency"] > prev["weak_scaling_efficiency"]
        prev = m
    assert bench.comm_model(nbytes, 8, 0.62)["direct_ms"] < bench.comm_model(nbytes, 8, 0.62)["rccl_ms"]  # 14 hops against 2
