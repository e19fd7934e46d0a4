"""Gaussian cloud state with the reference's tensor layout (scene/gaussian_model.py:53-60,350-357):
params = {_xyz[P,3], _features_dc[P,1,3], _features_rest[P,15,3], _opacity[P,1], _scaling[P,3],
_rotation[P,4]} as leaf tensors, one Adam group each (scene/gaussian_model.py:382-409)."""
import math

import numpy as np
import torch

PARAM_NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def expon_lr(step, lr_init, lr_final, max_steps):
    """get_expon_lr_func without delay (utils/general_utils.py:155-188)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class OptimizationParams:
    """Defaults of arguments/__init__.py:111-130 that the hot path reads."""
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_max_steps = 30_000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.005
    rotation_lr = 0.001
    percent_dense = 0.01
    densify_grad_threshold = 0.0002


class GaussianCloud:
    def __init__(self, params, sh_degree=3, device="cuda", spatial_lr_scale=5.0, scene_radius=0.75):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = 0
        self.spatial_lr_scale = spatial_lr_scale  # scene/gaussian_model.py:257
        self.params = {
            k: torch.as_tensor(np.asarray(params[k]) if not torch.is_tensor(params[k]) else params[k],
                               dtype=torch.float32).to(device).contiguous().requires_grad_(True)
            for k in PARAM_NAMES
        }
        P = self.num_points
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.variables = {"max_radii2D": z(P), "xyz_gradient_accum": z(P, 1), "denom": z(P, 1),
                          "scene_radius": torch.tensor(float(scene_radius), device=device)}
        self.optimizer = None
        self.cam = None

    @property
    def num_points(self):
        return int(self.params["_xyz"].shape[0])

    # activations (scene/gaussian_model.py:118-138)
    @property
    def get_scaling(self):
        return torch.exp(self.params["_scaling"])

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self.params["_rotation"])

    @property
    def get_opacity(self):
        return torch.sigmoid(self.params["_opacity"])

    @property
    def get_xyz(self):
        return self.params["_xyz"]

    @property
    def get_features(self):
        return torch.cat((self.params["_features_dc"], self.params["_features_rest"]), dim=1)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def training_setup(self, opt=OptimizationParams, eps=1e-15, fused=True):
        """Adam with the reference's per-group learning rates (scene/gaussian_model.py:382-409).
        fused=True: one HIP launch per step (fsgs_amd.optim.FusedAdam); False: torch.optim.Adam."""
        lr = {"_xyz": opt.position_lr_init * self.spatial_lr_scale, "_features_dc": opt.feature_lr,
              "_features_rest": opt.feature_lr / 20.0, "_opacity": opt.opacity_lr, "_scaling": opt.scaling_lr,
              "_rotation": opt.rotation_lr}
        groups = [{"params": [self.params[k]], "lr": lr[k], "name": k} for k in PARAM_NAMES]
        if fused:
            from .optim import FusedAdam

            self.optimizer = FusedAdam(groups, lr=0.0, eps=eps)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=eps)
        self.opt = opt
        return self.optimizer

    def update_learning_rate(self, iteration):
        lr = expon_lr(iteration, self.opt.position_lr_init * self.spatial_lr_scale,
                      self.opt.position_lr_final * self.spatial_lr_scale, self.opt.position_lr_max_steps)
        for g in self.optimizer.param_groups:
            if g["name"] == "_xyz":
                g["lr"] = lr
        return lr

    def add_densification_stats(self, viewspace_grad, update_filter):
        """scene/gaussian_model.py:678-681 (torch statement; the trainer uses optim.densify_stats)."""
        self.variables["xyz_gradient_accum"][update_filter] += torch.norm(
            viewspace_grad[update_filter], dim=-1, keepdim=True)
        self.variables["denom"][update_filter] += 1
