"""tests/ref_glue.py -- the independent restatement of the reference's render() glue that the GPU parity tests use as their
render() oracle -- pinned on CPU (VERDICT r5 weak #1a: the oracle must not share code with the product):
  * piece by piece against fixtures generated from the imported reference (tests/golden/make_golden.py):
    eval_sh.npz, pose_glue.npz, depth_sil.npz;
  * as a composition against render_composition.npz: the reference's OWN render() run around the same C oracle rasteriser
    (tests/golden/make_render_golden.py), all three (gs_grad, cam_grad) modes, every output, every gradient, the side effects."""
import os

import numpy as np
import pytest
import torch

from tests import ref_cpu, ref_glue

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = lambda a: torch.tensor(np.asarray(a))


def test_ref_glue_imports_no_product_code():
    src = open(ref_glue.__file__).read()
    assert "import fsgs_amd" not in src and "from fsgs_amd" not in src
    src = open(ref_cpu.__file__).read().split("def oracle_backend")[0]  # (oracle_backend serves the CPU harness, a different check)
    assert "import fsgs_amd" not in src and "from fsgs_amd" not in src


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_and_gradients_match_the_reference(deg):
    g = np.load(os.path.join(G, "eval_sh.npz"))
    s = T(g["sh"]).requires_grad_(True)
    d = T(g["dirs"]).requires_grad_(True)
    rgb = torch.clamp_min(ref_glue.sh_to_colour(deg, s, d) + 0.5, 0.0)
    (rgb * T(g["w"])).sum().backward()
    np.testing.assert_allclose(rgb.detach().numpy(), g[f"rgb{deg}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s.grad.numpy(), g[f"dsh{deg}"], rtol=1e-5, atol=1e-6)
    if deg > 0:
        np.testing.assert_allclose(d.grad.numpy(), g[f"ddir{deg}"], rtol=1e-4, atol=2e-5)


def test_learn_pose_and_transform_to_frame_match_the_reference():
    g = np.load(os.path.join(G, "pose_glue.npz"))
    for cam in range(g["r"].shape[2]):
        r = T(g["r"]).requires_grad_(True)
        t = T(g["t"]).requires_grad_(True)
        w2c = ref_glue.learn_pose(r, t, cam)
        (w2c * T(g["wsum"])).sum().backward()
        np.testing.assert_allclose(w2c.detach().numpy(), g[f"w2c_{cam}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r.grad.numpy(), g[f"dr_{cam}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(t.grad.numpy(), g[f"dt_{cam}"], rtol=1e-5, atol=1e-7)
    for gg, cg in ((True, True), (True, False), (False, True)):
        a = T(g["ttf_xyz"]).requires_grad_(True)
        m = T(g["ttf_w2c"]).requires_grad_(True)
        y = ref_glue.to_frame(a, m, gg, cg)
        (y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
        key = f"{int(gg)}{int(cg)}"
        np.testing.assert_allclose(y.detach().numpy(), g["ttf_" + key], rtol=1e-5, atol=1e-6)
        dx = a.grad.numpy() if a.grad is not None else np.zeros((64, 3), np.float32)
        dm = m.grad.numpy() if m.grad is not None else np.zeros((4, 4), np.float32)
        np.testing.assert_allclose(dx, g["ttf_dx_" + key], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dm[:3], g["ttf_dm_" + key][:3], rtol=1e-4, atol=1e-5)


def test_depth_silhouette_colours_match_the_reference():
    g = np.load(os.path.join(G, "depth_sil.npz"))
    np.testing.assert_allclose(ref_glue.depth_silhouette_colours(T(g["pts"]), T(g["viewmatrix_stored"])).numpy(), g["ds_stored"],
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ref_glue.depth_silhouette_colours(T(g["pts"]), torch.eye(4)).numpy(), g["ds_identity"], rtol=1e-6)


@pytest.mark.parametrize("deg,index,gs_grad,cam_grad", [(2, 1, True, True), (3, 2, True, False), (1, 1, False, True)])
def test_composition_equals_the_references_own_render(oracle32, deg, index, gs_grad, cam_grad):
    g = np.load(os.path.join(G, "render_composition.npz"))
    tag = "_d%d_i%d_g%d_c%d" % (deg, index, int(gs_grad), int(cam_grad))
    W, H = int(g["W"]), int(g["H"])
    cam = ref_glue.Settings(image_height=H, image_width=W, tanfovx=float(g["tanfovx"]), tanfovy=float(g["tanfovy"]),
                            bg=torch.ones(3), scale_modifier=1.0, viewmatrix=T(g["viewmatrix"]), projmatrix=T(g["projmatrix"]),
                            sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False)
    pc = ref_glue.Cloud({k: T(g["p" + k]) for k in ref_glue.Cloud.NAMES}, cam, deg)
    pc.variables["max_radii2D"] = T(g["max_radii2D_before"]).clone()
    poses = ref_glue.Poses(T(g["r"]), T(g["t"]), torch.zeros(3))
    with ref_cpu.oracle_backend(oracle32):
        pkg = ref_cpu.render_reference(poses, index, pc, gs_grad=gs_grad, cam_grad=cam_grad)
    loss = (pkg["render"] * T(g["wi"])).sum() + (pkg["render_dep"] * T(g["wd"])).sum() + (pkg["render_opacity"] * T(g["ws"])).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss" + tag])) <= 1e-6 * abs(float(g["loss" + tag]))
    for k in ("render", "render_dep", "render_w2c", "render_opacity", "uncertainty"):
        want = g[k + tag]
        np.testing.assert_allclose(pkg[k].detach().numpy(), want, rtol=0, atol=2e-6 * np.abs(want).max(), err_msg=k)
    for k in ("nan_mask", "presence_mask", "visibility_filter", "radii"):
        assert np.array_equal(pkg[k].numpy(), g[k + tag]), k
    assert not pkg["uncertainty"].requires_grad
    np.testing.assert_array_equal(pc.variables["max_radii2D"].numpy(), g["max_radii2D" + tag])
    assert np.array_equal(pc.variables["seen"].numpy(), g["seen" + tag])
    assert pc.variables["means2D"] is pkg["viewspace_points"]

    def close(got, want, name):
        got = np.zeros_like(want) if got is None else got.numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=5e-6 * max(np.abs(want).max(), 1e-12), err_msg=name)

    for k in ref_glue.Cloud.NAMES:
        close(pc.params[k].grad, g["d" + k + tag], k)
    close(poses.r.grad, g["dr" + tag], "r")
    close(poses.t.grad, g["dt" + tag], "t")
    close(pkg["viewspace_points"].grad if gs_grad else None, g["dviewspace" + tag], "viewspace")
    if not gs_grad:  # gs_grad=False detaches ONLY means3D: the SH view direction still reaches _xyz (SURVEY a1 note v)
        assert np.abs(g["d_xyz" + tag]).max() > 0 and np.abs(g["d_opacity" + tag]).max() > 0
    if not cam_grad:
        assert not np.any(g["dr" + tag]) and not np.any(g["dt" + tag])
