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


    # ---- densification / pruning / opacity reset (scene/gaussian_model.py:452-456,501-676; Appendix D) ----
    def _swap_params(self, new_tensors, moment_fn):
        """Install new leaf tensors for every group and transform the Adam moments with
        moment_fn(name, old_moment) -> new_moment (the reference's state surgery, :501-580)."""
        for group in self.optimizer.param_groups:
            name = group["name"]
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = new_tensors[name].detach().contiguous().requires_grad_(True)
            if state is not None and "exp_avg" in state:
                state["exp_avg"] = moment_fn(name, state["exp_avg"])
                state["exp_avg_sq"] = moment_fn(name, state["exp_avg_sq"])
                self.optimizer.state[new] = state
            group["params"][0] = new
            self.params[name] = new

    def prune_points(self, mask):
        keep = ~mask
        self._swap_params({k: self.params[k].detach()[keep] for k in PARAM_NAMES}, lambda n, m: m[keep])
        for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
            self.variables[k] = self.variables[k][keep]

    def densification_postfix(self, new):
        dev = self.params["_xyz"].device
        cat = {k: torch.cat((self.params[k].detach(), new[k]), dim=0) for k in PARAM_NAMES}
        self._swap_params(cat, lambda n, m: torch.cat((m, torch.zeros_like(new[n])), dim=0))
        P = self.num_points
        self.variables["xyz_gradient_accum"] = torch.zeros((P, 1), device=dev)
        self.variables["denom"] = torch.zeros((P, 1), device=dev)
        self.variables["max_radii2D"] = torch.zeros((P,), device=dev)

    def densify_and_clone(self, grads, grad_threshold):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (
            torch.max(self.get_scaling, dim=1).values <= self.variables["scene_radius"] * 0.01)
        self.densification_postfix({k: self.params[k].detach()[sel] for k in PARAM_NAMES})

    def densify_and_split(self, grads, grad_threshold, N=2):
        from .synth import quat_to_rot  # noqa: F401  (same polynomial as build_rotation)

        P0 = self.num_points
        padded = torch.zeros((P0,), device=self.params["_xyz"].device)
        padded[: grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (
            torch.max(self.get_scaling, dim=1).values > self.variables["scene_radius"] * 0.01)
        scale_sel = self.get_scaling.detach()[sel]
        stds = scale_sel.repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=stds.device), std=stds)
        q = self.params["_rotation"].detach()[sel]
        q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
        R = R.repeat(N, 1, 1)
        new = {
            "_xyz": torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + self.params["_xyz"].detach()[sel].repeat(N, 1),
            "_scaling": torch.log(scale_sel.repeat(N, 1) / (0.8 * N)),
            "_rotation": self.params["_rotation"].detach()[sel].repeat(N, 1),
            "_features_dc": self.params["_features_dc"].detach()[sel].repeat(N, 1, 1),
            "_features_rest": self.params["_features_rest"].detach()[sel].repeat(N, 1, 1),
            "_opacity": self.params["_opacity"].detach()[sel].repeat(N, 1),
        }
        self.densification_postfix(new)
        prune = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=sel.device, dtype=torch.bool)))
        self.prune_points(prune)

    def densify_and_prune(self, max_grad, min_opacity, max_screen_size):
        grads = self.variables["xyz_gradient_accum"].reshape(-1, 1) / self.variables["denom"].reshape(-1, 1)
        self.densify_and_clone(grads, max_grad)
        self.densify_and_split(grads, max_grad)
        prune = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            big_vs = self.variables["max_radii2D"] > max_screen_size
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * self.variables["scene_radius"]
            prune = prune | big_vs | big_ws
        self.prune_points(prune)

    def densify_and_prune_device(self, max_grad, min_opacity, max_screen_size):
        """densify_and_prune as ONE plan + ONE gather on the device (csrc/densify.hip): same decisions, same
        output order and the same torch.normal draws as the reference sequence above, without its ~12
        cat / boolean-index reallocations of every tensor and Adam moment.  One host read (the four totals)."""
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        P = self.num_points
        dev = self.params["_xyz"].device
        if not self.params["_xyz"].is_cuda:
            raise RuntimeError("device-side densification needs CUDA/HIP tensors; there is no CPU fallback")
        radius = float(self.variables["scene_radius"])
        acc = self.variables["xyz_gradient_accum"].detach().reshape(-1).contiguous().float()
        den = self.variables["denom"].detach().reshape(-1).contiguous().float()
        counts = torch.empty((4, P), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            stream = _lib.current_stream()
            _lib.check(lib.fsgs_densify_plan(P, _lib.ptr(acc), _lib.ptr(den), _lib.ptr(self.params["_scaling"].detach()),
                                             _lib.ptr(self.params["_opacity"].detach()), float(max_grad),
                                             float(min_opacity), float(np.float32(radius * 0.01)),
                                             float(np.float32(0.1 * radius)), int(bool(max_screen_size)),
                                             _lib.ptr(counts), stream), "fsgs_densify_plan")
            incl = torch.cumsum(counts, dim=1, dtype=torch.int32)
            totals = [int(v) for v in incl[:, -1].tolist()] if P > 0 else [0, 0, 0, 0]  # the one host read
            K0, K1, NS, K3 = totals
            Pn = K0 + K1 + 2 * K3
            # the reference's torch.normal(mean=zeros[2 NS,3], std=stds): randn of that shape, then * std
            normals = torch.randn((2 * NS, 3), device=dev) if NS > 0 else torch.zeros((0, 3), device=dev)
            roles = {"_xyz": 1, "_scaling": 2, "_rotation": 3}
            new_p, new_m, new_v, keep = {}, {}, {}, []
            arr = (_lib.FsgsDensifyGroup * len(PARAM_NAMES))()
            for k, name in enumerate(PARAM_NAMES):
                old = self.params[name].detach()
                row = int(old.numel() // max(P, 1))
                st = self.optimizer.state.get(self.params[name], None) if self.optimizer is not None else None
                has = st is not None and "exp_avg" in st
                new_p[name] = torch.empty((Pn,) + tuple(old.shape[1:]), dtype=torch.float32, device=dev)
                if has:
                    new_m[name], new_v[name] = torch.empty_like(new_p[name]), torch.empty_like(new_p[name])
                    m, v = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
                    keep += [m, v]
                arr[k].in_param = old.data_ptr()
                arr[k].in_exp_avg = m.data_ptr() if has else None
                arr[k].in_exp_avg_sq = v.data_ptr() if has else None
                arr[k].out_param = new_p[name].data_ptr()
                arr[k].out_exp_avg = new_m[name].data_ptr() if has else None
                arr[k].out_exp_avg_sq = new_v[name].data_ptr() if has else None
                arr[k].row, arr[k].role = row, roles.get(name, 0)
            src = torch.empty((max(Pn, 1),), dtype=torch.int32, device=dev)
            aux = torch.empty((max(Pn, 1),), dtype=torch.int32, device=dev)
            tot = (C.c_int32 * 4)(*totals)
            _lib.check(lib.fsgs_densify_apply(P, _lib.ptr(counts), _lib.ptr(incl), tot, len(PARAM_NAMES), arr,
                                              _lib.ptr(normals), _lib.ptr(src), _lib.ptr(aux), stream),
                       "fsgs_densify_apply")
        self._swap_params(new_p, lambda n, old_m: None)
        for group in self.optimizer.param_groups:  # the moments were gathered by the same launch
            st = self.optimizer.state.get(group["params"][0], None)
            if st is not None and "exp_avg" in st:
                st["exp_avg"], st["exp_avg_sq"] = new_m[group["name"]], new_v[group["name"]]
        self.variables["xyz_gradient_accum"] = torch.zeros((Pn, 1), device=dev)
        self.variables["denom"] = torch.zeros((Pn, 1), device=dev)
        self.variables["max_radii2D"] = torch.zeros((Pn,), device=dev)
        return {"kept": K0, "cloned": K1, "split": NS, "children_kept": 2 * K3}

    def reset_opacity(self):
        op = self.get_opacity.detach()
        new = inverse_sigmoid(torch.min(op, torch.ones_like(op) * 0.01))
        for group in self.optimizer.param_groups:
            if group["name"] != "_opacity":
                continue
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            t = new.detach().contiguous().requires_grad_(True)
            if state is not None and "exp_avg" in state:
                state["exp_avg"] = torch.zeros_like(new)
                state["exp_avg_sq"] = torch.zeros_like(new)
                self.optimizer.state[t] = state
            group["params"][0] = t
            self.params["_opacity"] = t

    def initialize_optimizer(self, fused=True):
        """global_run's fresh Adam with default eps and the mapping learning rates (scene/gaussian_model.py:372-378)."""
        lr = {"_xyz": self.opt.position_lr_init * self.spatial_lr_scale, "_features_dc": self.opt.feature_lr,
              "_features_rest": self.opt.feature_lr / 20.0, "_opacity": self.opt.opacity_lr,
              "_scaling": self.opt.scaling_lr, "_rotation": self.opt.rotation_lr}
        groups = [{"params": [self.params[k]], "lr": lr[k], "name": k} for k in PARAM_NAMES]
        if fused:
            from .optim import FusedAdam

            self.optimizer = FusedAdam(groups)
        else:
            self.optimizer = torch.optim.Adam(groups)
