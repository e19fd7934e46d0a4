"""A builder-written STAND-IN for an unchanged Free-SurGS checkout: the module tree and the names its driver binds
(train.py:5,30; gaussian_renderer/__init__.py:15,49-92; utils/loss_utils.py:47-54,98-127; scene/gaussian_model.py:12,18,378-409),
with bodies that state what the reference's code does THROUGH this repository's own restatements (the two-pass render
sequence on the rasteriser drop-in, the plain-torch losses, torch.optim.Adam) -- no line of the reference is in it.
tests/test_autobind_gpu.py and bench.py's `drop_in_step.autobind` import it twice: plainly (= what an unchanged checkout
runs on the drop-in) and with fsgs_amd.autobind installed (= the same files, the fused path bound by name).

    tree = write_tree(tmp_dir); sys.path.insert(0, tree); import standin_train
"""
import os
import textwrap

FILES = {
    "gaussian_renderer/__init__.py": '''
        """stand-in for gaussian_renderer/__init__.py: `render` = the reference's sequence of torch ops around TWO rasteriser
        calls (restated by fsgs_amd.render.render_two_pass), reached through the drop-in import name like the original"""
        import torch
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401  (:15)
        from scene.gaussian_model import GaussianModel  # noqa: F401  (:16)


        def render(viewpoint_camera, index, pc, gs_grad=True, cam_grad=True):
            from fsgs_amd.render import render_two_pass

            return render_two_pass(viewpoint_camera, index, pc, gs_grad=gs_grad, cam_grad=cam_grad)
    ''',
    "utils/__init__.py": "",
    "utils/loss_utils.py": '''
        """stand-in for utils/loss_utils.py: the three mapping losses as plain torch (fsgs_amd.losses.*_torch restate them)"""
        import torch  # noqa: F401
        from fsgs_amd import losses as _l


        def l1_loss(a, b):
            return _l.l1_loss(a, b)


        def rgb_loss_func(img, gt, lambda_dssim=0.2, mask=None):
            return _l.rgb_loss_torch(img, gt, lambda_dssim, mask)


        def pearson_depth_loss(src, tgt):
            return _l.pearson_torch(src, tgt)


        def local_pearson_loss(src, tgt, box, p_corr):
            return _l.local_pearson_torch(src, tgt, box, p_corr)
    ''',
    "scene/__init__.py": "from scene.gaussian_model import GaussianModel  # noqa: F401\n",
    "scene/gaussian_model.py": '''
        """stand-in for scene/gaussian_model.py: the optimizer construction spelled `torch.optim.Adam(...)` on the module's
        own global `torch` (:12, :378, :405); the cloud itself is fsgs_amd.model.GaussianCloud (same attributes)"""
        import torch
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401  (:18)
        from simple_knn._C import distCUDA2  # noqa: F401  (:23)
        from fsgs_amd.model import PARAM_NAMES, GaussianCloud


        class GaussianModel(GaussianCloud):
            def training_setup(self, opt=None, eps=1e-15):
                from fsgs_amd.model import OptimizationParams

                opt = opt or OptimizationParams
                lr = {"_xyz": opt.position_lr_init * self.spatial_lr_scale, "_features_dc": opt.feature_lr,
                      "_features_rest": opt.feature_lr / 20.0, "_opacity": opt.opacity_lr, "_scaling": opt.scaling_lr,
                      "_rotation": opt.rotation_lr}
                groups = [{"params": [self.params[k]], "lr": lr[k], "name": k} for k in PARAM_NAMES]
                self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=eps)
                self.opt = opt
                return self.optimizer

            def add_densification_stats(self, viewspace_point_tensor, update_filter):
                # (:678-681: boolean-mask indexing, i.e. a host synchronisation per statement)
                self.variables["xyz_gradient_accum"][update_filter] += torch.norm(
                    viewspace_point_tensor.grad[update_filter], dim=-1, keepdim=True)
                self.variables["denom"][update_filter] += 1
    ''',
    "standin_train.py": '''
        """stand-in for the mapping iteration of train.py:236-272,297-303: the names are bound at import (train.py:5,30)"""
        import torch
        from gaussian_renderer import render
        from scene import GaussianModel  # noqa: F401
        from utils.loss_utils import rgb_loss_func, pearson_depth_loss, local_pearson_loss

        LOSS_W_RGB = 5.0


        def mapping_iteration(poses, gaussians, colors, monodeps, timestep, statistics=True):
            render_pkg = render(poses, timestep, gaussians, gs_grad=True, cam_grad=False)
            image = render_pkg["render"]
            rgb_loss = rgb_loss_func(image, colors[timestep]) * LOSS_W_RGB
            mono_dep = monodeps[timestep]
            pearson_dep_loss = pearson_depth_loss(mono_dep, render_pkg["render_dep"])
            lp_loss = local_pearson_loss(mono_dep, render_pkg["render_dep"], 128, 0.5)
            loss = rgb_loss + (pearson_dep_loss * 0.05 + lp_loss * 0.15)
            loss.backward()
            with torch.no_grad():
                if statistics:  # FreeSurGS.densification, train.py:297-303 (statements of the driver itself)
                    vis, radii = render_pkg["visibility_filter"], render_pkg["radii"]
                    gaussians.variables["max_radii2D"][vis] = torch.max(gaussians.variables["max_radii2D"][vis], radii[vis].float())
                    gaussians.add_densification_stats(render_pkg["viewspace_points"], vis)
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
            return loss.detach(), render_pkg
    ''',
}
MODULES = ("standin_train", "gaussian_renderer", "utils.loss_utils", "utils", "scene.gaussian_model", "scene")


def write_tree(root):
    for rel, src in FILES.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(textwrap.dedent(src).lstrip("\n"))
    return root


def forget():
    """drop the stand-in's modules from sys.modules (to import the same tree again with / without the binding)"""
    import sys

    for m in MODULES:
        sys.modules.pop(m, None)
