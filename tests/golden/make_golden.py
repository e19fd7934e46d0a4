"""Generates tests/golden/*.npz by IMPORTING the reference's Python (CPU) in the dev container.

    python tests/golden/make_golden.py            # needs /root/reference; never runs on the GPU box

Only data (inputs + the reference's outputs/gradients) is written; no reference source or
bytecode is copied.  Recipe of SURVEY.md s8c: (1) MagicMock stubs for the 16 third-party packages
missing here, (2) Tensor.cuda = identity, (3) a TorchFunctionMode that rewrites device='cuda'.
"""
import os
import random
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

MISSING = ["diff_gaussian_rasterization", "simple_knn", "kornia", "plyfile", "cv2", "torchvision", "lpips",
           "skimage", "imageio", "viser", "nerfview", "jaxtyping", "loguru", "open3d", "torchviz", "wandb",
           "roma", "trimesh"]


class _StubFinder:
    """Any import below a missing third-party package resolves to a MagicMock package."""

    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery

        if fullname.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock()
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


for _n in list(MISSING):
    try:
        __import__(_n)
        MISSING.remove(_n)
    except Exception:
        pass
sys.meta_path.insert(0, _StubFinder())

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        d = kwargs.get("device")
        if d is not None and "cuda" in str(d):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


sys.path.insert(0, REF)


def seed_all(s=0):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def main():
    with CudaToCpu():
        from utils import loss_utils, sh_utils  # noqa: E402
        from utils import general_utils  # noqa: E402
        from scene import pose_optimizer  # noqa: E402

        # ---------------- eval_sh (utils/sh_utils.py:57-112) ----------------
        seed_all(0)
        P = 257
        sh = torch.randn(P, 3, 16, dtype=torch.float32)
        dirs = torch.nn.functional.normalize(torch.randn(P, 3), dim=1)
        out = {"sh": npy(sh), "dirs": npy(dirs)}
        for deg in range(4):
            s = sh.clone().requires_grad_(True)
            d = dirs.clone().requires_grad_(True)
            rgb = torch.clamp_min(sh_utils.eval_sh(deg, s, d) + 0.5, 0.0)  # scene/gaussian_model.py:319-320
            w = torch.linspace(-1, 1, P * 3).reshape(P, 3)
            (rgb * w).sum().backward()
            out[f"rgb{deg}"] = npy(rgb)
            out[f"dsh{deg}"] = npy(s.grad)
            out[f"ddir{deg}"] = npy(d.grad)
        out["w"] = npy(torch.linspace(-1, 1, P * 3).reshape(P, 3))
        np.savez_compressed(os.path.join(OUT, "eval_sh.npz"), **out)

        # ---------------- rgb_loss_func / ssim / l1 (utils/loss_utils.py:41-96) ----------------
        seed_all(1)
        H, W = 70, 93
        img = torch.rand(3, H, W)
        gt = (img + 0.1 * torch.randn(3, H, W)).clamp(0, 1)
        mask = (torch.rand(1, H, W) > 0.3)
        out = {"img": npy(img), "gt": npy(gt), "mask": npy(mask)}
        for tag, m in (("nomask", None), ("mask", mask)):
            x = img.clone().requires_grad_(True)
            loss = loss_utils.rgb_loss_func(x, gt, mask=m)
            loss.backward()
            out[f"loss_{tag}"] = npy(loss)
            out[f"grad_{tag}"] = npy(x.grad)
        x = img.clone().requires_grad_(True)
        s = loss_utils.ssim(x, gt)
        s.backward()
        out["ssim"] = npy(s)
        out["ssim_grad"] = npy(x.grad)
        out["l1"] = npy(loss_utils.l1_loss(img, gt))
        np.savez_compressed(os.path.join(OUT, "rgb_loss.npz"), **out)

        # ---------------- pearson / local pearson (utils/loss_utils.py:98-127) ----------------
        seed_all(2)
        H, W = 300, 420
        tgt = torch.rand(H, W) + 0.5
        src = (tgt * 0.7 + 0.2 * torch.rand(H, W)).contiguous()
        x = tgt.clone().requires_grad_(True)
        l = loss_utils.pearson_depth_loss(src, x)
        l.backward()
        out = {"src": npy(src), "tgt": npy(tgt), "pearson": npy(l), "pearson_grad_tgt": npy(x.grad)}
        x = src.clone().requires_grad_(True)
        l = loss_utils.pearson_depth_loss(x, tgt)
        l.backward()
        out["pearson_grad_src"] = npy(x.grad)
        # local: record the corners the reference draws (same RNG consumption: two randint calls)
        seed_all(3)
        box, pc = 128, 0.5
        gstate = torch.get_rng_state()
        nh, nw = H // box, W // box
        n_corr = int(pc * nh * nw)
        x0 = torch.randint(0, H - box, size=(n_corr,))
        y0 = torch.randint(0, W - box, size=(n_corr,))
        torch.set_rng_state(gstate)
        x = tgt.clone().requires_grad_(True)
        l = loss_utils.local_pearson_loss(src, x, box, pc)
        l.backward()
        out.update({"lp_x0": npy(x0), "lp_y0": npy(y0), "lp_loss": npy(l), "lp_grad_tgt": npy(x.grad),
                    "lp_box": box, "lp_p": pc})
        np.savez_compressed(os.path.join(OUT, "pearson.npz"), **out)

        # ---------------- pose glue (scene/pose_optimizer.py:822-877, 960-989) ----------------
        seed_all(4)
        N = 5
        lp = pose_optimizer.LearnPose(N, True, True, None, 0.0, 0.0, 100, 100, None, None, 1.0, "cpu") \
            if False else None
        # LearnPose's constructor signature is long and irrelevant; build the two tensors it owns and call
        # its methods unbound.
        LP = pose_optimizer.LearnPose
        obj = LP.__new__(LP)
        torch.nn.Module.__init__(obj)
        obj.r = torch.nn.Parameter(torch.randn(1, 4, N) * 0.2 + torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1))
        obj.t = torch.nn.Parameter(torch.randn(3, N) * 0.1)
        out = {"r": npy(obj.r), "t": npy(obj.t)}
        wsum = torch.arange(16, dtype=torch.float32).reshape(4, 4) / 16.0 - 0.4
        for cam in range(N):
            obj.zero_grad()
            w2c = obj.forward(cam)
            (w2c * wsum).sum().backward()
            out[f"w2c_{cam}"] = npy(w2c)
            out[f"dr_{cam}"] = npy(obj.r.grad)
            out[f"dt_{cam}"] = npy(obj.t.grad)
        out["wsum"] = npy(wsum)
        xyz = torch.randn(64, 3)
        w2c = obj.forward(2).detach()
        for gg, cg in ((True, True), (True, False), (False, True)):
            a = xyz.clone().requires_grad_(True)
            m = w2c.clone().requires_grad_(True)
            y = pose_optimizer.transform_to_frame(a, m, gg, cg)
            (y * torch.linspace(-1, 1, y.numel()).reshape(y.shape)).sum().backward()
            out[f"ttf_{int(gg)}{int(cg)}"] = npy(y)
            out[f"ttf_dx_{int(gg)}{int(cg)}"] = npy(a.grad) if a.grad is not None else np.zeros((64, 3), np.float32)
            out[f"ttf_dm_{int(gg)}{int(cg)}"] = npy(m.grad) if m.grad is not None else np.zeros((4, 4), np.float32)
        out["ttf_xyz"] = npy(xyz)
        out["ttf_w2c"] = npy(w2c)
        # build_rotation (utils/general_utils.py:204-226)
        q = torch.randn(33, 4)
        out["br_q"] = npy(q)
        out["br_R"] = npy(general_utils.build_rotation(q))
        out["inv_sigmoid_in"] = np.linspace(0.01, 0.99, 17).astype(np.float32)
        out["inv_sigmoid_out"] = npy(general_utils.inverse_sigmoid(torch.tensor(out["inv_sigmoid_in"])))
        f = general_utils.get_expon_lr_func(1.6e-4 * 5, 1.6e-6 * 5, max_steps=30000)
        steps = np.array([0, 1, 10, 1000, 15000, 30000, 40000])
        out["lr_steps"] = steps
        out["lr_vals"] = np.array([f(int(s)) for s in steps])
        np.savez_compressed(os.path.join(OUT, "pose_glue.npz"), **out)

        # ---------------- projection_flow_loss (scene/pose_optimizer.py:164-218) ----------------
        seed_all(5)
        H, W = 96, 128
        K = np.array([[1035.0 * W / 1280, 0, 596.5 * W / 1280], [0, 1035.0 * H / 1024, 520.5 * H / 1024], [0, 0, 1]],
                     dtype=np.float64)
        u = torch.arange(W).float()[None, :] / W
        v = torch.arange(H).float()[:, None] / H
        depth_prev = (1.0 + 0.3 * torch.sin(6.28 * u) * torch.cos(6.28 * v)).reshape(1, H, W).contiguous()
        depth_prev[0, :5, :7] = 0.0  # invalid-depth pixels are dropped
        flow = torch.randn(2, 2, H, W) * 1.5  # flows_fw[index-1]: [2,H,W]
        rigid = (torch.rand(H, W) > 0.2)
        w2c_prev = np.eye(4, dtype=np.float32)
        w2c_prev[:3, 3] = [0.01, -0.02, 0.005]
        q = torch.tensor([1.0, 0.01, -0.02, 0.015])
        t = torch.tensor([0.02, -0.01, 0.03])
        obj2 = LP.__new__(LP)
        torch.nn.Module.__init__(obj2)
        obj2.r = torch.nn.Parameter(q.reshape(1, 4, 1).clone())
        obj2.t = torch.nn.Parameter(t.reshape(3, 1).clone())
        w2c_cur = obj2.forward(0)
        rec = {"flows_fw": flow, "intrinsic": K.astype(np.float64)}
        out = {"depth_prev": npy(depth_prev), "flow": npy(flow), "rigid": npy(rigid), "w2c_prev": w2c_prev, "K": K,
               "q": npy(q), "t": npy(t)}
        for tag, rm in (("rigid", rigid), ("norigid", None)):
            obj2.zero_grad()
            w2c_cur = obj2.forward(0)
            w2c_cur.retain_grad()
            l = pose_optimizer.projection_flow_loss(1, depth_prev, w2c_prev, w2c_cur, rec, rm)
            l.backward()
            out[f"loss_{tag}"] = npy(l)
            out[f"dw2c_{tag}"] = npy(w2c_cur.grad)
            out[f"dr_{tag}"] = npy(obj2.r.grad)
            out[f"dt_{tag}"] = npy(obj2.t.grad)
        out["w2c_cur"] = npy(w2c_cur)
        np.savez_compressed(os.path.join(OUT, "flow_loss.npz"), **out)

        # ---------------- depth/silhouette pseudo colours (scene/gaussian_model.py:260-275) ----------------
        from scene import gaussian_model  # noqa: E402

        GM = gaussian_model.GaussianModel
        gm = GM.__new__(GM)
        pts = torch.randn(40, 3) + torch.tensor([0, 0, 2.0])
        V = torch.eye(4).unsqueeze(0)
        ds = GM.get_depth_and_silhouette(gm, pts, V)
        Vt = torch.tensor(np.linalg.inv(np.array([[0.99, 0.01, 0.02, 0.1], [-0.01, 0.98, 0.03, -0.2],
                                                  [0.0, -0.02, 1.0, 0.3], [0, 0, 0, 1.0]])), dtype=torch.float32).T
        ds2 = GM.get_depth_and_silhouette(gm, pts, Vt.unsqueeze(0))
        np.savez_compressed(os.path.join(OUT, "depth_sil.npz"), pts=npy(pts), ds_identity=npy(ds),
                            viewmatrix_stored=npy(Vt), ds_stored=npy(ds2))
        # ---------------- pose metrics (utils/geometry_utils.py:18-29, utils/utils_poses/*) ----------------
        from utils import geometry_utils  # noqa: E402

        seed_all(6)
        out = {}
        for case in range(3):
            n = 12 + 5 * case
            ang = np.cumsum(np.random.randn(n, 3) * 0.03, axis=0)
            pos = np.cumsum(np.random.randn(n, 3) * 0.05, axis=0)
            gt = np.tile(np.eye(4), (n, 1, 1))
            from scipy.spatial.transform import Rotation as Rot

            gt[:, :3, :3] = Rot.from_rotvec(ang).as_matrix()
            gt[:, :3, 3] = pos
            sim_R = Rot.from_rotvec([0.3, -0.2, 0.5]).as_matrix()
            pred = gt.copy()
            pred[:, :3, :3] = sim_R @ gt[:, :3, :3] @ Rot.from_rotvec(np.random.randn(n, 3) * 0.01).as_matrix()
            pred[:, :3, 3] = (gt[:, :3, 3] @ sim_R.T) * 1.7 + np.array([0.3, -1.0, 2.0]) + np.random.randn(n, 3) * 0.01
            import contextlib, io

            with contextlib.redirect_stdout(io.StringIO()):
                aligned, metrics = geometry_utils.align_pose(torch.tensor(pred).float(), torch.tensor(gt).float())
            out[f"gt_{case}"] = gt.astype(np.float32)
            out[f"pred_{case}"] = pred.astype(np.float32)
            out[f"aligned_{case}"] = npy(aligned)
            out[f"metrics_{case}"] = np.array(metrics, dtype=np.float64)  # rpe_trans, rpe_rot(deg), ate
        np.savez_compressed(os.path.join(OUT, "pose_metrics.npz"), **out)

        # ---------------- densify / prune / opacity reset (scene/gaussian_model.py:501-681) ----------------
        from scene import gaussian_model as gmod  # noqa: E402

        seed_all(7)
        GM = gmod.GaussianModel
        P = 400
        gm = GM.__new__(GM)
        gm.setup_functions()
        gm.max_sh_degree, gm.active_sh_degree = 3, 0
        raw = {"_xyz": torch.randn(P, 3), "_features_dc": torch.randn(P, 1, 3), "_features_rest": torch.randn(P, 15, 3),
               "_opacity": torch.randn(P, 1) * 2.0, "_scaling": torch.randn(P, 3) * 0.8 - 4.0,
               "_rotation": torch.randn(P, 4)}
        gm.params = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
        gm.variables = {"max_radii2D": torch.rand(P) * 40, "xyz_gradient_accum": torch.rand(P, 1) * 1e-3,
                        "denom": torch.randint(0, 3, (P, 1)).float(), "scene_radius": torch.tensor(0.75)}
        groups = [{"params": [gm.params[k]], "lr": 1e-3, "name": k} for k in raw]
        gm.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        for k in raw:  # one Adam step so the moments are non-trivial
            gm.params[k].grad = torch.randn_like(gm.params[k]) * 1e-2
        gm.optimizer.step()
        before = {k: npy(gm.params[k]) for k in raw}
        mom = {k: npy(gm.optimizer.state[gm.params[k]]["exp_avg"]) for k in raw}
        mom2 = {k: npy(gm.optimizer.state[gm.params[k]]["exp_avg_sq"]) for k in raw}
        var = {k: npy(v) for k, v in gm.variables.items()}
        out = {"P": P}
        out.update({"p_" + k: v for k, v in before.items()})
        out.update({"m_" + k: v for k, v in mom.items()})
        out.update({"v_" + k: v for k, v in mom2.items()})
        out.update({"var_" + k: v for k, v in var.items()})
        import contextlib, io

        torch.manual_seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            gm.densify_and_prune(2e-4, 0.05, 20)
        out.update({"d_" + k: npy(gm.params[k]) for k in raw})
        out.update({"dm_" + k: npy(gm.optimizer.state[gm.params[k]]["exp_avg"]) for k in raw})
        out.update({"dvar_" + k: npy(v) for k, v in gm.variables.items()})
        with contextlib.redirect_stdout(io.StringIO()):
            gm.reset_opacity()
        out["r_opacity"] = npy(gm.params["_opacity"])
        out["r_m_opacity"] = npy(gm.optimizer.state[gm.params["_opacity"]]["exp_avg"])
        np.savez_compressed(os.path.join(OUT, "densify.npz"), **out)

        # ---------------- on-disk sequence layout -> record_data (scene/pose_optimizer.py:355-460) ----------------
        # inputs are laid down by the BUILD's writer (fsgs_amd.dataset.write_sequence), read by the reference's
        # PoseModel.__init__; the fixture keeps the raw arrays (inputs) and what the reference made of them (outputs)
        import argparse
        import contextlib
        import io
        import tempfile

        sys.path.insert(0, os.path.join(os.path.dirname(OUT), "..", "free-surgs_amd"))
        from fsgs_amd import checkpoint as ckpt  # noqa: E402
        from fsgs_amd import dataset as fds  # noqa: E402

        rng = np.random.default_rng(21)
        n, H, W = 11, 20, 36
        colors_u8 = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
        disparity = rng.uniform(0.2, 3.0, (n, H, W)).astype(np.float32)
        flows_fw = rng.normal(0, 2.0, (n - 1, 2, H, W)).astype(np.float32)
        flows_bw = rng.normal(0, 2.0, (n - 1, 2, H, W)).astype(np.float32)
        cam_poses = np.tile(np.eye(4), (n, 1, 1))
        cam_poses[:, :3, 3] = rng.normal(0, 0.1, (n, 3))
        KL = np.array([[1035.3, 0.0, 596.5], [0.0, 1035.1, 520.4], [0.0, 0.0, 1.0]])
        runs = ["5"] * 7 + ["6"] * 4
        with tempfile.TemporaryDirectory() as tmp:
            fds.write_sequence(tmp, colors_u8, disparity, flows_fw, flows_bw, cam_poses, KL, scene="1", data=runs)
            out = {"colors_u8": colors_u8, "disparity": disparity, "flows_fw_in": flows_fw, "flows_bw_in": flows_bw,
                   "cam_poses": cam_poses, "KL": KL, "runs": np.array(runs)}
            for tag, (fs, fe) in (("all", (0, -1)), ("slice", (2, 9))):
                a = argparse.Namespace(source_path=tmp, data_type="scared", frame_start=fs, frame_end=fe)
                with contextlib.redirect_stdout(io.StringIO()):
                    pm = pose_optimizer.PoseModel(a, device="cpu")
                rd = pm.record_data
                out.update({tag + "_colors": npy(rd["colors"]), tag + "_flows_fw": npy(rd["flows_fw"]),
                            tag + "_flows_bw": npy(rd["flows_bw"]), tag + "_monodeps": npy(rd["monodeps"]),
                            tag + "_intrinsic": np.asarray(rd["intrinsic"]), tag + "_data_ind": np.asarray(rd["data_ind"]),
                            tag + "_weights": np.asarray(rd["weights"]), tag + "_i_test": np.asarray(pm.i_test),
                            tag + "_i_train": np.asarray(pm.i_train), tag + "_fov": np.array([pm.FovX, pm.FovY]),
                            tag + "_proj": npy(pm.projection_matrix), tag + "_num_cams": pm.num_cams})
                for k, v in rd["gt_poses"].items():
                    out["%s_gt_%s" % (tag, k)] = npy(v)
            np.savez_compressed(os.path.join(OUT, "dataset.npz"), **out)

            # ---------------- checkpoints (scene/gaussian_model.py:86-116, scene/pose_optimizer.py:472-487) -----------
            # a chkpnt / poses pair WRITTEN BY THE REFERENCE classes (tensors + Adam state_dicts: data only) ...
            seed_all(23)
            P = 53
            gm = GM.__new__(GM)
            gm.setup_functions()
            gm.max_sh_degree, gm.active_sh_degree, gm.spatial_lr_scale = 3, 2, 5.0
            gm.params = {"_xyz": torch.randn(P, 3), "_features_dc": torch.randn(P, 1, 3),
                         "_features_rest": torch.randn(P, 15, 3), "_opacity": torch.randn(P, 1),
                         "_scaling": torch.randn(P, 3) - 4.0, "_rotation": torch.randn(P, 4)}
            gm.params = {k: torch.nn.Parameter(v) for k, v in gm.params.items()}
            gm.variables = {"max_radii2D": torch.rand(P) * 30, "xyz_gradient_accum": torch.rand(P, 1),
                            "denom": torch.randint(0, 5, (P, 1)).float()}
            targs = argparse.Namespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                                       position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                       opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
            gm.training_setup(targs)
            gm.variables["xyz_gradient_accum"] = torch.rand(P, 1)
            gm.variables["denom"] = torch.randint(1, 5, (P, 1)).float()
            for it in range(3):
                for k in gm.params:
                    gm.params[k].grad = torch.randn_like(gm.params[k]) * 1e-2
                gm.optimizer.step()
            a = argparse.Namespace(source_path=tmp, data_type="scared", frame_start=0, frame_end=-1)
            with contextlib.redirect_stdout(io.StringIO()):
                pm = pose_optimizer.PoseModel(a, device="cpu")
            with torch.no_grad():
                pm.pose_param_net.r.add_(0.05 * torch.randn_like(pm.pose_param_net.r))
                pm.pose_param_net.t.add_(0.05 * torch.randn_like(pm.pose_param_net.t))
            pm.initialize_tracking_optimizer(50)
            for it in range(2):
                pm.pose_param_net.r.grad = torch.randn_like(pm.pose_param_net.r) * 1e-2
                pm.pose_param_net.t.grad = torch.randn_like(pm.pose_param_net.t) * 1e-2
                pm.optimizer.step()
            for i in range(0, pm.num_cams, 2):
                pm.record_data["pred_w2c"][i] = npy(pm.pose_param_net(i))
            torch.save((gm.capture(), 7), os.path.join(OUT, "ref_chkpnt7.pth"))
            torch.save((pm.capture(), 7), os.path.join(OUT, "ref_poses7.pth"))

            # ... and the other direction checked right here, where the reference can run: a pair written by the
            # build (fsgs_amd.checkpoint.save) restores into the reference's classes and its Adam keeps stepping
            from fsgs_amd.model import GaussianCloud  # noqa: E402
            from fsgs_amd.trainer import PoseTrack  # noqa: E402

            pc = GaussianCloud({k: npy(v) for k, v in gm.params.items()}, sh_degree=3, device="cpu")
            pc.training_setup(fused=False)
            for k in pc.params:
                pc.params[k].grad = torch.randn_like(pc.params[k]) * 1e-2
            pc.optimizer.step()
            pt = PoseTrack(pm.num_cams, device="cpu")
            pt.optimizer = torch.optim.Adam([{"params": pt.r, "lr": 0.01}, {"params": pt.t, "lr": 0.01}], lr=0.001,
                                            eps=1e-15)
            pt.r.grad, pt.t.grad = torch.randn_like(pt.r), torch.randn_like(pt.t)
            pt.optimizer.step()
            with torch.no_grad():
                pt.get_pose(3)
            ckpt.save(tmp, 9, pc, pt, np.eye(3))
            mp, it9 = torch.load(os.path.join(tmp, "chkpnt9.pth"), weights_only=False)
            g2 = GM.__new__(GM)
            g2.setup_functions()
            g2.params, g2.variables = {}, {}
            g2.restore(mp, targs)
            assert it9 == 9 and all(torch.equal(g2.params[k], pc.params[k]) for k in pc.params)
            for k in g2.params:
                g2.params[k].grad = torch.ones_like(g2.params[k])
            g2.optimizer.step()  # torch's Adam accepts the state the build wrote
            assert float(g2.optimizer.state[g2.params["_xyz"]]["step"]) == 2.0
            pp, _ = torch.load(os.path.join(tmp, "poses9.pth"), weights_only=False)
            pm.restore(pp)
            assert torch.equal(pm.pose_param_net.r, pt.r) and torch.equal(pm.pose_param_net.t, pt.t)
            assert np.allclose(pm.record_data["pred_w2c"][3], npy(pt.pred_w2c[3])) and not pm.record_data["pred_w2c"][4].any()
        # ---------------- PoseModel.setup_camera (scene/pose_optimizer.py:600-633) ----------------
        # the 12 keyword arguments it hands to GaussianRasterizationSettings (stubbed here: a MagicMock records them),
        # for the record_data route (intrinsics as the loader rescales them, :413-414) and the visualize_data route
        PM = pose_optimizer.PoseModel
        out = {}
        cases = [
            ("c2_identity", 1280, 1024, np.eye(4), None),
            ("c1_posed", 640, 512, None, None),
            ("vis_posed", 1920, 1080, None, "vis"),
        ]
        seed_all(31)
        for tag, W_, H_, w2c_, route in cases:
            KL_ = np.array([[1035.3, 0.0, 596.5], [0.0, 1035.1, 520.4], [0.0, 0.0, 1.0]])
            K_ = KL_.copy()
            K_[0, :] = K_[0, :] * W_ / 1280
            K_[1, :] = K_[1, :] * H_ / 1024
            if w2c_ is None:
                from scipy.spatial.transform import Rotation as Rot

                w2c_ = np.eye(4)
                w2c_[:3, :3] = Rot.from_rotvec(np.random.randn(3) * 0.05).as_matrix()
                w2c_[:3, 3] = np.random.randn(3) * 0.05
            w2c_ = w2c_.astype(np.float32)
            pm_ = PM.__new__(PM)
            pm_.record_data = {"intrinsic": K_, "image_width": W_, "image_height": H_}
            pose_optimizer.Camera.reset_mock()
            if route == "vis":
                pm_.record_data = {}
                pm_.setup_camera(w2c_, visualize_data={"K": K_, "W": W_, "H": H_}, near=0.05, far=20)
            else:
                pm_.setup_camera(w2c_)
            kw = pose_optimizer.Camera.call_args.kwargs
            assert sorted(kw) == sorted(["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                         "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug"])
            out[tag + "_K"] = K_
            out[tag + "_w2c"] = w2c_
            out[tag + "_near_far"] = np.array([0.05, 20.0] if route == "vis" else [0.01, 100.0])
            for k, v in kw.items():
                out["%s_%s" % (tag, k)] = npy(v) if torch.is_tensor(v) else np.asarray(v)
            out[tag + "_cam_center"] = npy(pm_.cam_center)
        out["cases"] = np.array([c[0] for c in cases])
        np.savez_compressed(os.path.join(OUT, "camera.npz"), **out)

        # ---------------- Sigma = R S S^T R^T (scene/gaussian_model.py:32-36, utils/general_utils.py:191-236) --------
        # GaussianModel.get_covariance's activation on raw (un-normalised) quaternions and activated scales: the six
        # upper-triangular entries (xx, xy, xz, yy, yz, zz) strip_symmetric keeps
        seed_all(32)
        GM = gaussian_model.GaussianModel
        gmc = GM.__new__(GM)
        gmc.setup_functions()
        n = 48
        scaling = torch.exp(torch.randn(n, 3) * 0.7 - 3.0)
        rotq = torch.randn(n, 4)
        rotq[0] = torch.tensor([1.0, 0, 0, 0])
        rotq[1] = torch.tensor([0.0, 0, 0, 2.0])      # 180 deg about z, un-normalised
        cov6 = gmc.covariance_activation(scaling, 1.0, rotq)
        cov6_m = gmc.covariance_activation(scaling, 1.7, rotq)
        np.savez_compressed(os.path.join(OUT, "covariance.npz"), scaling=npy(scaling), rotation_raw=npy(rotq),
                            rotation_normalised=npy(torch.nn.functional.normalize(rotq)), cov6=npy(cov6),
                            cov6_modifier_1p7=npy(cov6_m))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
