"""Checkpoints in the reference's own tuple layouts, so that a `chkpnt{it}.pth` / `poses{it}.pth` pair written by either
side loads on the other (train.py:107-111,371-376):

    chkpnt{it}.pth = (GaussianModel.capture(), it)   12-tuple, scene/gaussian_model.py:86-100
    poses{it}.pth  = (PoseModel.capture(), it)       5-tuple,  scene/pose_optimizer.py:472-487

The Adam state travels as a torch.optim state_dict: the HIP FusedAdam keeps `step` as a python int, torch's Adam as a
0-d float tensor -- capture converts, restore accepts both."""
import os

import numpy as np
import torch

from .model import PARAM_NAMES, OptimizationParams

# order of the parameter tensors inside the 12-tuple (NOT the optimizer's group order)
_CAPTURE_ORDER = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


# what torch.optim.Adam's step() reads from a param_group besides lr / betas / eps.  Adam.load_state_dict adopts the
# saved groups verbatim (its __setstate__ only defaults amsgrad / maximize / foreach / capturable / differentiable /
# fused), so a group saved without weight_decay raises KeyError at the first step of whoever restored it.
_TORCH_ADAM_GROUP_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                                  differentiable=False, fused=None, decoupled_weight_decay=False)


def portable_state_dict(optimizer):
    """optimizer.state_dict() as torch.optim.Adam would have written it: every `step` a 0-d float32 CPU tensor and
    every param_group carrying the keys torch's Adam reads (FusedAdam's own defaults are lr / betas / eps only)."""
    sd = optimizer.state_dict()
    for st in sd["state"].values():
        if "step" in st and not torch.is_tensor(st["step"]):
            st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
    for g in sd["param_groups"]:
        for k, v in _TORCH_ADAM_GROUP_DEFAULTS.items():
            g.setdefault(k, v)
    return sd


def capture_gaussians(pc):
    if pc.optimizer is None:
        raise RuntimeError("capture needs the optimizer (training_setup / initialize_optimizer first)")
    v = pc.variables
    return (pc.active_sh_degree, *(pc.params[k] for k in _CAPTURE_ORDER), v["max_radii2D"], v["xyz_gradient_accum"],
            v["denom"], portable_state_dict(pc.optimizer), pc.spatial_lr_scale)


def restore_gaussians(pc, model_args, opt=OptimizationParams, eps=1e-15, fused=True):
    """GaussianModel.restore (scene/gaussian_model.py:102-116): install the tensors, rebuild the optimizer with the
    mapping learning rates, load the Adam state.  Anything holding the old tensors (a FastStepper) must be rebuilt."""
    if len(model_args) != 12:
        raise ValueError("expected the 12-tuple of GaussianModel.capture(), got %d entries" % len(model_args))
    dev = pc.params["_xyz"].device
    # always a COPY: a tuple captured from a live model (not read from disk) must not end up sharing parameter or
    # moment storage with the model it is restored into
    plain = lambda t: torch.as_tensor(t).detach().to(device=dev, dtype=torch.float32).clone(
        memory_format=torch.contiguous_format)
    leaf = lambda t: plain(t).requires_grad_(True)
    pc.active_sh_degree = int(model_args[0])
    for k, t in zip(_CAPTURE_ORDER, model_args[1:7]):
        pc.params[k] = leaf(t)
    P = pc.num_points
    for k in PARAM_NAMES:
        if int(pc.params[k].shape[0]) != P:
            raise ValueError("checkpoint tensors disagree on the number of Gaussians (%s)" % k)
    pc.variables["max_radii2D"] = plain(model_args[7])
    # upstream's training_setup (scene/gaussian_model.py:384-385) runs AFTER the tuple is unpacked and replaces the two
    # densification accumulators with zeros: a restored run starts its statistics afresh.  Same here.
    pc.variables["xyz_gradient_accum"] = torch.zeros_like(plain(model_args[8]))
    pc.variables["denom"] = torch.zeros_like(plain(model_args[9]))
    pc.spatial_lr_scale = float(model_args[11])
    pc.training_setup(opt, eps=eps, fused=fused)
    sd = model_args[10]
    pc.optimizer.load_state_dict({
        "param_groups": sd["param_groups"],
        "state": {k: {n: (v.detach().clone() if torch.is_tensor(v) else v) for n, v in st.items()}
                  for k, st in sd["state"].items()}})
    return pc


def capture_poses(poses, intrinsic):
    """(optimizer state_dict, r [1,4,N], t [3,N], pred_w2c [N,4,4] float64 numpy, intrinsic [3,3])."""
    N = int(poses.r.shape[-1])
    pred = np.zeros((N, 4, 4))
    for i, m in enumerate(poses.pred_w2c):
        if m is not None:
            pred[i] = m.detach().cpu().numpy()
    sd = portable_state_dict(poses.optimizer) if poses.optimizer is not None else {"state": {}, "param_groups": []}
    # upstream assigns r / t back into an nn.Module (LearnPose), which only accepts nn.Parameter
    as_param = lambda t: torch.nn.Parameter(t.detach(), requires_grad=True)
    return (sd, as_param(poses.r), as_param(poses.t), pred, np.asarray(intrinsic))


def restore_poses(poses, model_args):
    """PoseModel.restore (scene/pose_optimizer.py:481-487): r, t, pred_w2c and the intrinsics come back; the
    optimizer state is read and dropped, as upstream does (tracking builds a fresh Adam per frame).  -> intrinsic."""
    if len(model_args) != 5:
        raise ValueError("expected the 5-tuple of PoseModel.capture(), got %d entries" % len(model_args))
    _, r, t, pred, intrinsic = model_args
    dev = poses.r.device
    r = torch.as_tensor(r).detach().to(device=dev, dtype=torch.float32).clone()
    t = torch.as_tensor(t).detach().to(device=dev, dtype=torch.float32).clone()
    if tuple(r.shape[:2]) != (1, 4) or tuple(t.shape[:1]) != (3,) or r.shape[-1] != t.shape[-1]:
        raise ValueError("pose tensors must be r[1,4,N] and t[3,N]")
    poses.r = r.contiguous().requires_grad_(True)
    poses.t = t.contiguous().requires_grad_(True)
    pred = np.asarray(pred)
    poses.pred_w2c = [torch.tensor(pred[i], dtype=torch.float32, device=dev) if np.any(pred[i]) else None
                      for i in range(pred.shape[0])]
    poses.__dict__.pop("_w2c_cache", None)
    poses.optimizer = None
    poses.scheduler = None
    return np.asarray(intrinsic)


def save(model_path, iteration, pc, poses, intrinsic):
    os.makedirs(model_path, exist_ok=True)
    torch.save((capture_gaussians(pc), iteration), os.path.join(model_path, "chkpnt%d.pth" % iteration))
    torch.save((capture_poses(poses, intrinsic), iteration), os.path.join(model_path, "poses%d.pth" % iteration))


def load(checkpoint_path, pc, poses, opt=OptimizationParams, eps=1e-15, fused=True):
    """checkpoint_path = .../chkpnt{it}.pth; the pose file is found by the same substitution train.py:109 uses.
    -> (iteration, intrinsic)."""
    model_params, first_iter = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    restore_gaussians(pc, model_params, opt, eps=eps, fused=fused)
    pose_params, first_iter = torch.load(checkpoint_path.replace("chkpnt", "poses"), map_location="cpu",
                                         weights_only=False)
    return first_iter, restore_poses(poses, pose_params)
