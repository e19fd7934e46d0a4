"""Counterpart of the reference's training harness for the hot path (train.py:154-295):
`tracking_step` (pose-only, train.py:166-200) and `mapping_step` (Gaussians, train.py:236-272).

The reference's train.py runs unchanged on the drop-in packages, but it cannot travel to the GPU
box (14 third-party imports are absent), so tests and bench.py drive the path
through this harness, which reproduces the same call sequence and hyper-parameters.
"""
import numpy as np
import torch

from . import losses
from .model import GaussianCloud
from .pose import pose_to_w2c
from .rasterizer import GaussianRasterizationSettings
from .render import render, render_two_pass

LOSS_W_MAPPING = {"rgb": 5.0, "pearson": 0.05, "local_pearson": 0.15}  # train.py:254-258
LOSS_W_TRACKING = {"rgb": 1.0, "flow": 0.1}  # train.py:180-184


def settings_from_cam(cam, device="cuda"):
    """numpy camera dict (fsgs_amd.synth.make_camera) -> GaussianRasterizationSettings on the device."""
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=device)
    return GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=float(cam["tanfovx"]),
        tanfovy=float(cam["tanfovy"]), bg=t(cam["bg"]), scale_modifier=1.0,
        viewmatrix=t(cam["viewmatrix"]).unsqueeze(0), projmatrix=t(cam["projmatrix"]).unsqueeze(0), sh_degree=0,
        campos=t(cam["campos"]), prefiltered=False, debug=False)


class PoseTrack:
    """LearnPose + the slice of PoseModel the hot path touches (scene/pose_optimizer.py:772-777,
    489-516, 600-638): quaternions r[1,4,N], translations t[3,N], get_pose(t), cam_center."""

    def __init__(self, num_cams, device="cuda"):
        r = torch.zeros((1, 4, num_cams), dtype=torch.float32, device=device)
        r[:, 0, :] = 1.0
        self.r = r.requires_grad_(True)
        self.t = torch.zeros((3, num_cams), dtype=torch.float32, device=device).requires_grad_(True)
        self.cam_center = torch.zeros(3, dtype=torch.float32, device=device)  # frame 0 is the origin
        self.pred_w2c = [None] * num_cams  # kept on the device (the reference copies to host every call)
        self.optimizer = None
        self.scheduler = None

    def set_pose(self, i, q, t):
        with torch.no_grad():
            self.r[0, :, i] = torch.as_tensor(q, dtype=torch.float32, device=self.r.device)
            self.t[:, i] = torch.as_tensor(t, dtype=torch.float32, device=self.t.device)

    def get_pose(self, i):
        w2c = pose_to_w2c(self.r, self.t, int(i))
        self.pred_w2c[int(i)] = w2c.detach()
        return w2c

    def initialize_tracking_optimizer(self, tracking_iter=50):
        """Adam(lr .01, eps 1e-15) + MultiStepLR(milestones 0,16,32,48; gamma .5)
        (scene/pose_optimizer.py:489-496)."""
        from .optim import FusedAdam

        self.optimizer = FusedAdam([{"params": self.r, "lr": 0.01}, {"params": self.t, "lr": 0.01}],
                                   lr=0.001, eps=1e-15)
        step = int(tracking_iter / 3)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(
            self.optimizer, milestones=list(range(0, int(tracking_iter), step)), gamma=0.5)

    def initialize_pose(self, i):
        """constant-velocity prediction for i >= 2 (scene/pose_optimizer.py:498-516)."""
        with torch.no_grad():
            if i > 1:
                n = torch.nn.functional.normalize
                r1, r2 = n(self.r[..., i - 1]), n(self.r[..., i - 2])
                self.r[..., i] = n(r1 + (r1 - r2))
                self.t[..., i] = self.t[..., i - 1] + (self.t[..., i - 1] - self.t[..., i - 2])
            else:
                self.r[..., i] = self.r[..., i - 1]
                self.t[..., i] = self.t[..., i - 1]


class FrameData:
    """Per-frame inputs kept resident in HBM: colour [3,H,W], mono-depth [H,W]."""

    def __init__(self, colors, monodeps):
        self.colors = colors
        self.monodeps = monodeps


def mapping_loss(pkg, gt_image, mono_dep, corners=None, hip_losses=True):
    """5 * rgb + 0.05 * pearson + 0.15 * local_pearson(128, 0.5)   (train.py:253-259)."""
    if hip_losses:
        rgb_f, pe_f, lp_f = losses.rgb_loss_func, losses.pearson_depth_loss, losses.local_pearson_loss
    else:  # the plain-PyTorch statement of the same maths (numerics reference of the fused kernels)
        rgb_f, pe_f, lp_f = losses.rgb_loss_torch, losses.pearson_torch, losses.local_pearson_torch
    rgb = rgb_f(pkg["render"], gt_image) * LOSS_W_MAPPING["rgb"]
    pear = pe_f(mono_dep, pkg["render_dep"])
    lp = lp_f(mono_dep, pkg["render_dep"], 128, 0.5, corners)
    return rgb + pear * LOSS_W_MAPPING["pearson"] + lp * LOSS_W_MAPPING["local_pearson"]


def mapping_step(pc, poses, frames, timesteps, fused=True, step_optimizer=True, hip_losses=True, grad_sync=None):
    """One mapping iteration over `timesteps` views with SUMMED loss (train.py:236-272).
    Densification statistics come from view 0 only (train.py:260-263)."""
    rend = render if fused else render_two_pass
    loss = 0
    first = None
    for k, ts in enumerate(timesteps):
        pkg = rend(poses, ts, pc, gs_grad=True, cam_grad=False)
        loss = loss + mapping_loss(pkg, frames.colors[ts], frames.monodeps[ts], hip_losses=hip_losses)
        if k == 0:
            first = pkg
    loss.backward()
    if grad_sync is not None:  # frame-sharded data parallel: sum the Gaussian gradients over ranks
        grad_sync(pc)
    with torch.no_grad():
        # densification statistics from view 0 (train.py:260-263,298-303) in one launch
        from . import optim

        optim.densify_stats(first["radii"], first["viewspace_points"].grad, pc.variables["max_radii2D"],
                            pc.variables["xyz_gradient_accum"], pc.variables["denom"])
        if step_optimizer:
            pc.optimizer.step()
            pc.optimizer.zero_grad(set_to_none=True)
    return loss.detach(), first
