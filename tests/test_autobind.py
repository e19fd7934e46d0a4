"""FSGS_AUTOBIND=1 (fsgs_amd/autobind.py): an UNCHANGED checkout's driver gets the fused path by name.  The checkout here is
a builder-written STAND-IN tree with the reference's module and function names (train.py:5,30;
gaussian_renderer/__init__.py:15,49; utils/loss_utils.py:47,98,112; scene/gaussian_model.py:12,18,378;
scene/pose_optimizer.py:1,3,5,490) -- never the reference's files -- run in a subprocess with the two PYTHONPATH entries
a user would set."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "free-surgs_amd")

STANDIN = {
    "gaussian_renderer/__init__.py": """
        import torch
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        from scene.gaussian_model import GaussianModel
        def render(viewpoint_camera, index, pc, gs_grad=True, cam_grad=True):
            return "standin render"
        def render_custom(*a):
            return render.__module__       # what the module's OWN global `render` is
        def inference(*a):
            return "standin inference"
    """,
    "utils/__init__.py": "",
    "utils/loss_utils.py": """
        import torch
        def l1_loss(a, b): return "standin l1"
        def rgb_loss_func(img, gt, lambda_dssim=0.2, mask=None): return "standin rgb"
        def pearson_depth_loss(src, tgt): return "standin pearson"
        def local_pearson_loss(src, tgt, box, p_corr): return "standin local"
    """,
    "scene/__init__.py": "from scene.gaussian_model import GaussianModel\n",
    "scene/gaussian_model.py": """
        import torch
        from torch import nn
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        from simple_knn._C import distCUDA2
        class GaussianModel:
            def training_setup(self, params):
                self.optimizer = torch.optim.Adam(params, lr=0.0, eps=1e-15)
                self.sched = torch.optim.lr_scheduler.ExponentialLR      # everything else of torch.optim is torch's own
                return self.optimizer
            def zeros(self):
                return torch.zeros(3)                                     # and so is the rest of torch
            def add_densification_stats(self, viewspace_point_tensor, update_filter):
                return "standin stats"
    """,
    "scene/pose_optimizer.py": """
        import torch
        import torch.nn as nn
        import torch.optim as optim
        from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
        from utils.loss_utils import l1_loss
        class PoseModel:
            def initialize_tracking_optimizer(self, params):
                self.optimizer = optim.Adam([{"params": params, "lr": 0.01}], lr=0.001, eps=1e-15)
                return self.optimizer
            def other(self, params):
                return torch.optim.Adam(params, lr=0.0, eps=1e-15), optim.SGD(params, lr=0.1)
    """,
    "train.py": """
        import json, sys, torch
        from gaussian_renderer import render, render_custom, inference
        from scene import GaussianModel
        from scene.pose_optimizer import PoseModel
        from utils.loss_utils import rgb_loss_func, pearson_depth_loss, local_pearson_loss
        import gaussian_renderer, utils.loss_utils, scene.gaussian_model
        p = [torch.zeros(4, requires_grad=True)]
        q = [torch.zeros(4, requires_grad=True)]
        adam_a, sgd = PoseModel().other(q)
        gm = GaussianModel()
        gm.training_setup(p)
        out = {
            "render": "%s.%s" % (render.__module__, render.__name__),
            "render_attr": "%s.%s" % (gaussian_renderer.render.__module__, gaussian_renderer.render.__name__),
            "render_custom": render_custom(),
            "inference": inference(),
            "losses": ["%s.%s" % (f.__module__, f.__name__) for f in (rgb_loss_func, pearson_depth_loss, local_pearson_loss)],
            "l1": utils.loss_utils.l1_loss(0, 0),
            "adam_model": type(GaussianModel().training_setup(p)).__module__ + "." + type(GaussianModel().training_setup(p)).__name__,
            "adam_pose": type(PoseModel().initialize_tracking_optimizer(p)).__name__,
            "adam_pose_torch_spelling": type(adam_a).__name__, "sgd": type(sgd).__module__,
            "sched": gm.sched.__module__,
            "zeros": GaussianModel().zeros().tolist(),
            "stats": GaussianModel.add_densification_stats.__module__,
        }
        try:
            from fsgs_amd import autobind
            out["bound"] = autobind.bound()
        except Exception as e:
            out["bound"] = repr(e)
        print("RESULT " + json.dumps(out))
    """,
}


def _run(tmp_path, env_extra, scene_first=False):
    for rel, src in STANDIN.items():
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        src = textwrap.dedent(src)
        if scene_first and rel == "train.py":  # a driver that imports the model before the renderer
            src = src.replace("import json, sys, torch\n", "import json, sys, torch\nfrom scene import GaussianModel\n", 1)
            assert src.count("from scene import GaussianModel") == 2
        f.write_text(src)
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([PKG, str(tmp_path)])
    env.pop("FSGS_AUTOBIND", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, str(tmp_path / "train.py")], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_without_the_variable_nothing_is_rebound(tmp_path):
    out = _run(tmp_path, {})
    assert out["render"] == "gaussian_renderer.render" and out["render_attr"] == "gaussian_renderer.render"
    assert out["losses"] == ["utils.loss_utils.rgb_loss_func", "utils.loss_utils.pearson_depth_loss",
                             "utils.loss_utils.local_pearson_loss"]
    assert out["adam_model"] == "torch.optim.adam.Adam" and out["adam_pose"] == "Adam"
    assert out["bound"] == {} and out["stats"] == "scene.gaussian_model"


def test_one_environment_variable_binds_render_losses_and_adam_by_name(tmp_path):
    out = _run(tmp_path, {"FSGS_AUTOBIND": "1"})
    # train.py:5 -- gaussian_renderer was IN FLIGHT when the rasteriser shim installed the binding (class swap)
    assert out["render"] == "fsgs_amd.render.render" and out["render_attr"] == "fsgs_amd.render.render"
    # (class swap: the module's own global keeps the original; a module that came through the finder is patched in place)
    assert out["render_custom"] == "gaussian_renderer" and out["inference"] == "standin inference"
    # train.py:30 -- utils.loss_utils was imported through the finder and patched right after its body ran
    assert out["losses"] == ["fsgs_amd.losses.rgb_loss_func", "fsgs_amd.losses.pearson_depth_loss",
                             "fsgs_amd.losses.local_pearson_loss"]
    assert out["l1"] == "standin l1"
    # scene/gaussian_model.py:378,405 and scene/pose_optimizer.py:490,888: both spellings construct FusedAdam ...
    assert out["adam_model"] == "fsgs_amd.optim.FusedAdam"
    assert out["adam_pose"] == "FusedAdam" and out["adam_pose_torch_spelling"] == "FusedAdam"
    # ... and nothing else of torch changed for those modules
    assert out["sgd"].startswith("torch.optim") and out["sched"].startswith("torch.optim") and out["zeros"] == [0.0, 0.0, 0.0]
    assert set(out["bound"]) == {"gaussian_renderer", "utils.loss_utils", "scene.gaussian_model", "scene.pose_optimizer"}
    # scene.gaussian_model came through the finder (patched when its body had run): Adam and the sync-free statistics method
    assert out["stats"] == "fsgs_amd.autobind" and "GaussianModel.add_densification_stats" in out["bound"]["scene.gaussian_model"]
    assert sorted(out["bound"]["utils.loss_utils"]) == ["local_pearson_loss", "pearson_depth_loss", "rgb_loss_func"]


def test_install_is_idempotent_and_patches_modules_imported_before_it(tmp_path):
    """the explicit route (no shim involved): modules imported BEFORE install() are patched in place"""
    (tmp_path / "utils").mkdir()
    (tmp_path / "utils" / "__init__.py").write_text("")
    (tmp_path / "utils" / "loss_utils.py").write_text("def rgb_loss_func(*a):\n    return 'standin'\n")
    code = textwrap.dedent("""
        import sys
        sys.path[:0] = [%r, %r]
        import utils.loss_utils
        from fsgs_amd import autobind
        r1 = autobind.install(); r2 = autobind.install()
        assert utils.loss_utils.rgb_loss_func.__module__ == "fsgs_amd.losses", utils.loss_utils.rgb_loss_func
        assert utils.loss_utils._fsgs_original_rgb_loss_func() == "standin"
        assert "pearson_depth_loss" not in vars(utils.loss_utils)        # a name the module never had is not invented
        assert sum(1 for f in sys.meta_path if type(f).__name__ == "_Finder") == 1
        autobind.uninstall()
        assert sum(1 for f in sys.meta_path if type(f).__name__ == "_Finder") == 0
        print("OK", r1)
    """ % (PKG, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


def test_a_driver_that_imports_the_model_first_is_bound_too(tmp_path):
    """`from scene import GaussianModel` before the renderer: scene.gaussian_model is IN FLIGHT when its rasteriser import
    (scene/gaussian_model.py:18) installs the binding -- `torch` is already a global of it and is proxied at once, the class
    does not exist yet and gets its method when the body has finished; gaussian_renderer then comes through the finder"""
    out = _run(tmp_path, {"FSGS_AUTOBIND": "1"}, scene_first=True)
    assert out["render"] == "fsgs_amd.render.render" and out["adam_model"] == "fsgs_amd.optim.FusedAdam"
    assert out["adam_pose"] == "FusedAdam" and out["losses"][0] == "fsgs_amd.losses.rgb_loss_func"
    assert out["stats"] == "fsgs_amd.autobind", out["bound"]
    assert out["render_custom"] == "fsgs_amd.render"  # gaussian_renderer came through the finder: patched in place
    assert out["zeros"] == [0.0, 0.0, 0.0] and out["sched"].startswith("torch.optim")
