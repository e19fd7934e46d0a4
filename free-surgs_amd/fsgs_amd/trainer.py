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
        if self.r.is_cuda:
            from .pose import pose_to_w2c_hip

            w2c = pose_to_w2c_hip(self.r, self.t, int(i))  # one launch each way (csrc/pose.hip)
        else:
            w2c = pose_to_w2c(self.r, self.t, int(i))  # CPU: the torch statement (gloo tests)
        self.pred_w2c[int(i)] = w2c.detach()
        return w2c

    def peek_pose(self, i):
        """w2c of frame i WITHOUT recording it in pred_w2c (the reference's pose_param_net.forward, as
        get_fundamental_matrix calls it, scene/pose_optimizer.py:640-648)."""
        old = self.pred_w2c[int(i)]
        with torch.no_grad():
            w = self.get_pose(i).detach()
        self.pred_w2c[int(i)] = old
        return w

    def _pose_key(self):
        # the tensor OBJECTS (kept alive by the entry) and their version counters: addresses alone can be recycled
        return (self.r, self.t, self.r._version, self.t._version)

    @staticmethod
    def _same_pose_key(a, b):
        return a[0] is b[0] and a[1] is b[1] and a[2:] == b[2:]

    def get_pose_detached(self, i):
        """w2c of frame i for callers that do not differentiate through the pose (mapping): cached until r / t change
        (in-place updates bump the tensors' version counters), so a mapping iteration does not relaunch the kernel."""
        key = self._pose_key()
        cache = self.__dict__.setdefault("_w2c_cache", {})
        hit = cache.get(int(i))
        if hit is None or not self._same_pose_key(hit[0], key):
            with torch.no_grad():
                w2c = self.get_pose(i).detach().contiguous()
            # the fused optimizer kernels write r / t behind autograd's back and bump the versions themselves
            # (optim.mark_updated); mapping never touches the poses, so its frames stay cached
            if len(cache) > 4096:
                cache.clear()
            cache[int(i)] = hit = (key, w2c)
        return hit[1]

    def initialize_tracking_optimizer(self, tracking_iter=50):
        """Adam(lr .01, eps 1e-15) + MultiStepLR(milestones 0,16,32,48; gamma .5)
        (scene/pose_optimizer.py:489-496)."""
        from .optim import FusedAdam

        self.optimizer = FusedAdam([{"params": self.r, "lr": 0.01}, {"params": self.t, "lr": 0.01}],
                                   lr=0.001, eps=1e-15)
        step = int(tracking_iter / 3)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(
            self.optimizer, milestones=list(range(0, int(tracking_iter), step)), gamma=0.5)

    @staticmethod
    def _lr_table(tracking_iter):
        """the learning rates the reference's Adam(lr .01) + MultiStepLR(milestones 0, s, 2s, ..; gamma .5) pair shows after
        its construction and after each scheduler.step() -- recorded ONCE from the torch objects themselves
        (scene/pose_optimizer.py:489-496), so a frame does not have to build them again to get the same floats"""
        import warnings

        dummy = [torch.zeros(1, requires_grad=True), torch.zeros(1, requires_grad=True)]
        opt = torch.optim.Adam([{"params": dummy[0], "lr": 0.01}, {"params": dummy[1], "lr": 0.01}], lr=0.001, eps=1e-15)
        step = int(tracking_iter / 3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=list(range(0, int(tracking_iter), step)), gamma=0.5)
            table = [tuple(float(g["lr"]) for g in opt.param_groups)]
            for _ in range(int(tracking_iter) + 1):
                sch.step()
                table.append(tuple(float(g["lr"]) for g in opt.param_groups))
        return table

    class _TableScheduler:
        """scheduler.step() of the per-frame MultiStepLR, replayed from PoseTrack._lr_table"""

        def __init__(self, optimizer, table):
            self.optimizer, self.table, self.pos = optimizer, table, 0
            self._apply()

        def _apply(self):
            row = self.table[min(self.pos, len(self.table) - 1)]
            for g, lr in zip(self.optimizer.param_groups, row):
                g["lr"] = lr

        def step(self):
            self.pos += 1
            self._apply()

        def rewind(self):
            self.pos = 0
            self._apply()

    def begin_frame(self, i, tracking_iter=50):
        """Start of a tracked frame (train.py:322-331): initialize_pose(i) -- constant velocity for i > 1, a copy of frame
        i-1 for i = 1 -- and initialize_tracking_optimizer(tracking_iter), i.e. an Adam with zero moments and the
        schedule back at its start.  On the device: ONE launch (fsgs_pose_frame_begin) on an optimizer / schedule that are
        built once and reset, instead of ~20 small torch kernels and two new Python objects per frame."""
        i = int(i)
        if not self.r.is_cuda:  # CPU (gloo tests): the torch statements
            self.initialize_pose(i)
            self.initialize_tracking_optimizer(tracking_iter)
            return
        from . import _lib
        from .optim import FusedAdam, mark_updated

        hit = self.__dict__.get("_frame_opt")
        if hit is None or hit[0] != int(tracking_iter) or hit[1] is not self.r or hit[2] is not self.t:
            opt = FusedAdam([{"params": self.r, "lr": 0.01}, {"params": self.t, "lr": 0.01}], lr=0.001, eps=1e-15)
            for p in (self.r, self.t):
                opt.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p, memory_format=torch.preserve_format),
                                "exp_avg_sq": torch.zeros_like(p, memory_format=torch.preserve_format)}
            hit = self._frame_opt = (int(tracking_iter), self.r, self.t, opt,
                                     PoseTrack._TableScheduler(opt, self._lr_table(tracking_iter)))
        opt, sch = hit[3], hit[4]
        self.optimizer, self.scheduler = opt, sch
        sch.rewind()
        sr, st_ = opt.state[self.r], opt.state[self.t]
        sr["step"] = st_["step"] = 0
        with torch.cuda.device(self.r.device):
            _lib.check(_lib.load().fsgs_pose_frame_begin(
                _lib.ptr(self.r), _lib.ptr(self.t), int(self.r.shape[-1]), i, 1, _lib.ptr(sr["exp_avg"]),
                _lib.ptr(sr["exp_avg_sq"]), _lib.ptr(st_["exp_avg"]), _lib.ptr(st_["exp_avg_sq"]), _lib.current_stream()),
                "fsgs_pose_frame_begin")
        mark_updated([self.r, self.t])

    def fused_step(self, i, dw2c_a, weight_a, dw2c_b):
        """scheduler-independent tail of a tracking iteration in one launch (csrc/pose.hip pose_adam_kernel):
        dW = weight_a * dw2c_a + dw2c_b -> LearnPose adjoint -> Adam on r, t -> the new w2c of frame i (cached for the
        next get_pose_detached).  Call scheduler.step() first, as train.py:189,194 does."""
        import ctypes as C

        from . import _lib
        from .optim import FusedAdam, mark_updated

        opt = self.optimizer
        if not isinstance(opt, FusedAdam) or not self.r.is_cuda:
            raise RuntimeError("fused pose step needs the HIP FusedAdam on device tensors; there is no CPU fallback")
        gr, gt_ = opt.param_groups[0], opt.param_groups[1]
        sts = []
        for p in (self.r, self.t):
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] = int(st["step"]) + 1
            sts.append(st)
        w2c_next = torch.empty((4, 4), dtype=torch.float32, device=self.r.device)
        N = int(self.r.shape[-1])
        with torch.cuda.device(self.r.device):
            rc = _lib.load().fsgs_pose_adam_step(
                _lib.ptr(self.r), _lib.ptr(self.t), N, int(i), _lib.ptr(dw2c_a), float(weight_a), _lib.ptr(dw2c_b),
                _lib.ptr(sts[0]["exp_avg"]), _lib.ptr(sts[0]["exp_avg_sq"]), _lib.ptr(sts[1]["exp_avg"]),
                _lib.ptr(sts[1]["exp_avg_sq"]), float(gr["lr"]), float(gt_["lr"]), int(sts[0]["step"]),
                int(sts[1]["step"]), float(gr["betas"][0]), float(gr["betas"][1]), float(gr["eps"]),
                _lib.ptr(w2c_next), _lib.current_stream())
        _lib.check(rc, "fsgs_pose_adam_step")
        mark_updated([self.r, self.t])
        self.pred_w2c[int(i)] = w2c_next
        self.__dict__.setdefault("_w2c_cache", {})[int(i)] = (self._pose_key(), w2c_next)

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
    """Per-frame inputs kept resident in HBM (PoseModel.record_data, scene/pose_optimizer.py:441-460):
    colours [3,H,W], mono-depth [H,W] (already normalised to [0.5,1.5], :406-407), forward flows
    flows_fw[i] [2,H,W] (frame i -> i+1), intrinsics K [3,3], predicted depths [H,W] per frame,
    optional ground-truth w2c poses for evaluation."""

    def __init__(self, colors, monodeps, flows_fw=None, K=None, gt_w2c=None):
        self.colors = colors
        self.monodeps = monodeps
        self.flows_fw = flows_fw
        self.K = K
        self.gt_w2c = gt_w2c
        self.pred_depths = [None] * len(colors)
        n = len(colors)
        idx = np.arange(n)
        self.i_test = idx[4::8]  # scene/pose_optimizer.py:416-419
        self.i_train = np.array([i for i in idx if i not in self.i_test])


def mapping_loss(pkg, gt_image, mono_dep, corners=None, hip_losses=True):
    """5 * rgb + 0.05 * pearson + 0.15 * local_pearson(128, 0.5)   (train.py:253-259)."""
    if hip_losses:
        rgb_f, pe_f, lp_f = losses.rgb_loss_func, losses.pearson_depth_loss, losses.local_pearson_loss
    else:  # the plain-PyTorch statement of the same maths (numerics reference of the fused kernels)
        rgb_f, pe_f, lp_f = losses.rgb_loss_torch, losses.pearson_torch, losses.local_pearson_torch
    rgb = rgb_f(pkg["render"], gt_image) * LOSS_W_MAPPING["rgb"]
    if hip_losses:  # global + patch Pearson from ONE pair of launches
        from . import loss_ops

        pear, lp = loss_ops.pearson_pair(mono_dep, pkg["render_dep"], 128, 0.5, corners)
    else:
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


def tracking_step(pc, poses, frames, t, targets, rigid_mask, fused=True):
    """One pose-only iteration of FreeSurGS.tracking (train.py:166-200): masked photometric loss +
    0.1 * flow reprojection loss, backward to the pose, scheduler.step() BEFORE optimizer.step()."""
    from .flow import flow_pose_loss

    rend = render if fused else render_two_pass
    pkg = rend(poses, t, pc, gs_grad=False, cam_grad=True)
    mask = pkg["render_dep"] > 0
    if rigid_mask is not None:  # None = every pixel rigid (frame 1 has no Sampson mask, train.py:158-162)
        mask = mask * rigid_mask
    mask = mask.unsqueeze(0)
    rgb = LOSS_W_TRACKING["rgb"] * losses.rgb_loss_func(pkg["render"], frames.colors[t], mask=mask)
    flow = LOSS_W_TRACKING["flow"] * flow_pose_loss(pkg["render_w2c"], targets)
    loss = flow + rgb
    loss.backward()
    poses.scheduler.step()
    with torch.no_grad():
        poses.optimizer.step()
        poses.optimizer.zero_grad(set_to_none=True)
    return loss.detach(), rgb.detach(), flow.detach(), pkg


class Runner:
    """Counterpart of the FreeSurGS driver class (train.py:33-443) for the hot path: progressive_run
    (track each new frame, map the training frames), global_run (random-frame mapping iterations with
    densification), validation (PSNR on the test frames) and eval_pose (RPE / ATE)."""

    def __init__(self, pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, fused=True,
                 seed=0, densify=True, row0_depth_quirk=True, densify_interval=300, opacity_reset_interval=3000,
                 densify_until=15000, trace=False, test_frame_quirks=True, profile=False):
        import random

        # profile=True: wall time of every phase of a frame cycle / the global loop, each bracketed by a device
        # synchronisation, appended to self.phase_ms as (tag, frame or iteration, ms) -- bench.py's `harness` extra and
        # scripts/harness_trace.py read it; off by default (the synchronisations are the cost)
        self.phase_ms = [] if profile else None

        # train.py:305-311: densify_and_prune at iteration % 300 == 0 while iteration < 15000, opacity reset at % 3000.
        # (Parameters so that a pinned short trajectory can cross a densification, tests/test_harness_pin_gpu.py.)
        self.densify_interval, self.opacity_reset_interval = int(densify_interval), int(opacity_reset_interval)
        self.densify_until = int(densify_until)
        # trace=True: every iteration's loss values are read back and appended to self.trace (one host sync each, like
        # the reference's per-iteration .item() prints); off by default
        self.trace = [] if trace else None

        self.pc, self.poses, self.frames = pc, poses, frames
        self.tracking_iter, self.mapping_iter, self.first_mapping_iter = tracking_iter, mapping_iter, first_mapping_iter
        self.fused = fused
        self.iteration = 0
        self.keyframes = []
        self.rng = random.Random(seed)
        self.densify = densify
        # train.py:343 stores render_dep[0] -- ROW 0 of the [H,W] depth, broadcast over all rows -- as the
        # previous-frame depth of the flow loss.  True reproduces the reference; False stores the full map.
        self.row0_depth_quirk = row0_depth_quirk
        # What an unchanged train.py does with a TEST frame t (every 8th, never mapped; train.py:333-343):
        #  * record_data['pred_depths'][t] stays at its initial zeros, so when frame t+1 is tracked get_pointcloud finds
        #    no valid pixel and projection_flow_loss returns the constant 0 (scene/pose_optimizer.py:176-186): that
        #    frame is tracked by the RGB loss alone;
        #  * record_data['pred_w2c'][t] is whatever the LAST get_pose(t) wrote (scene/pose_optimizer.py:635-638): the
        #    pose the 50th tracking iteration rendered with, i.e. BEFORE its optimizer step -- a mapped frame is
        #    refreshed by its mapping renders, a test frame is not (get_fundamental_matrix calls the network directly).
        # True reproduces both; False renders the test frame's depth once and records its final pose.
        self.test_frame_quirks = test_frame_quirks
        self.fast = None
        if fused:
            from .fast_step import FastStepper

            self.fast = FastStepper(pc, poses, frames)
        self.log = []
        H, W = (getattr(frames.colors, "shape", None) or frames.colors[0].shape)[-2:]  # a staged lane knows its item shape
        self.h, self.w = int(H), int(W)

    def _phase(self, tag, idx):
        """context manager: with profile=True the wall time of the block, device work included, lands in self.phase_ms"""
        import contextlib
        import time

        if getattr(self, "phase_ms", None) is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def timed():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                yield
            finally:
                torch.cuda.synchronize()
                self.phase_ms.append((tag, idx, (time.perf_counter() - t0) * 1e3))

        return timed()

    # ---- train.py:297-316 -------------------------------------------------------------------------------
    def densification(self):
        from . import dist as fdist

        it = self.iteration
        if not self.densify:
            return
        if it % self.densify_interval == 0 and it < self.densify_until:
            with self._phase("densify_and_prune", it):
                fdist.sync_densification_stats(self.pc)
                size_threshold = 20 if it > 4000 else None
                if self.fast is not None:  # one plan + one gather on the device (csrc/densify.hip)
                    self.pc.densify_and_prune_device(self.pc.opt.densify_grad_threshold, 0.05, size_threshold)
                else:
                    self.pc.densify_and_prune(self.pc.opt.densify_grad_threshold, 0.05, size_threshold)
            if self.trace is not None:
                self.trace.append(("densify", it, self.pc.num_points))
        if it % self.opacity_reset_interval == 0:
            self.pc.reset_opacity()

    def mapping(self, cur_t, mapping_iter, progressive, want_pkg=True):
        """want_pkg=False (global_run, train.py:386-393: the reference copies the render to the host there and nothing reads
        it): no render is handed back.  Otherwise `render` / `render_dep` of the last iteration, as tensors of the caller's
        own (the zero-copy hand-out of the step driver's buffers is _mapping(borrow=True), for callers inside this class
        that copy what they keep before the next step: ADVICE r4)."""
        # (_mapping(borrow=False) has already cloned what it hands out and carries no "views_of_step_buffers" marker)
        return self._mapping(cur_t, mapping_iter, progressive, want_pkg)

    def _mapping(self, cur_t, mapping_iter, progressive, want_pkg=True, borrow=False):
        """borrow=True: the returned `render` / `render_dep` may be VIEWS of the step driver's buffers (marked
        "views_of_step_buffers"), valid until the next step is enqueued."""
        views = 2 if (progressive and cur_t != 0) else 1
        self.pc.optimizer.zero_grad(set_to_none=True)
        pkg = None
        # the keyframe of the NEXT iteration is drawn during this one (same draws, same order, none more than train.py:239's
        # one per iteration), so that a staged sequence can have it on the device in time
        nxt = self.rng.choice(self.keyframes) if (views == 2 and mapping_iter > 0) else None
        self._prefetch(nxt, flows=False)  # (the first keyframe of the call as well: ADVICE r3)
        for k in range(mapping_iter):
            self.iteration += 1
            ts = [nxt, cur_t] if views == 2 else [cur_t]
            if views == 2:
                nxt = self.rng.choice(self.keyframes) if k + 1 < mapping_iter else None
                self._prefetch(nxt, flows=False)
            it = self.iteration
            special = self.densify and ((it % self.densify_interval == 0 and it < self.densify_until) or
                                        it % self.opacity_reset_interval == 0)
            if self.fast is not None and not special:
                # nothing happens between backward and optimizer.step() on this iteration (train.py:266-272):
                # the step driver may consume the gradient itself (Adam fused / compact gradient)
                self.fast.pc = self.pc
                # the statistics feed densify_and_prune only, whose last call is at iteration 14 700 (train.py:305)
                loss = self.fast.mapping_step(ts, step_optimizer=True,
                                              collect_stats=self.densify and it < self.densify_until)
                if self.trace is not None:
                    self.trace.append(("map", it, tuple(ts), float(loss)))
                pkg = None
                continue
            if self.fast is not None:
                self.fast.pc = self.pc
                loss = self.fast.mapping_step(ts, step_optimizer=False)
                _first = None
            else:
                loss, _first = mapping_step(self.pc, self.poses, self.frames, ts, fused=False, step_optimizer=False)
            if self.trace is not None:
                self.trace.append(("map", it, tuple(ts), float(loss)))
            with torch.no_grad():
                self.densification()
                self.pc.optimizer.step()
                self.pc.optimizer.zero_grad(set_to_none=True)
            pkg = _first if views == 1 else None
        if not want_pkg:
            return None
        if pkg is None and self.fast is not None and self.fast.last:
            # the reference returns the render of the LAST view of the LAST iteration (cur_t), taken before that
            # iteration's optimizer step (train.py:236-265, 291): exactly what the step driver still holds -- handed out
            # as views (round 3 cloned 21 MB here after every call, global_run's single iterations included)
            pkg = {"render": self.fast.last["image"], "render_dep": self.fast.last["depth_sil"][0], "views_of_step_buffers": True}
            if not borrow:
                pkg = {"render": pkg["render"].clone(), "render_dep": pkg["render_dep"].clone()}
        if pkg is None:
            with torch.no_grad():
                pkg = (render if self.fused else render_two_pass)(self.poses, cur_t, self.pc, gs_grad=False, cam_grad=False)
        return pkg

    def tracking(self, t):
        from .flow import FlowTargets

        # Sampson-distance rigid mask of frame t-2 under the optimised poses t-2, t-1 (train.py:157-165); all rigid
        # for t <= 1.  Once per frame: two launches (csrc/flow.hip), no sync beyond reading the two 4x4 poses.
        rigid = None
        with self._phase("tracking.rigid_mask", t):
            if t > 1 and self.frames.flows_fw is not None:
                from .epipolar import fundamental_from_w2c, rigid_mask

                with torch.no_grad():  # get_fundamental_matrix asks the network, not get_pose: pred_w2c is not refreshed
                    # (both poses in ONE device-to-host copy: each .cpu() is a synchronisation of its own)
                    pair = torch.stack((self.poses.peek_pose(t - 2), self.poses.peek_pose(t - 1))).cpu().numpy()
                    Fm = fundamental_from_w2c(pair[0], pair[1], self.frames.K)
                rigid, self.last_sampson, _ = rigid_mask(self.frames.flows_fw[t - 2], Fm)
            all_rigid = rigid is None
        with self._phase("tracking.flow_targets", t):
            depth_prev = self.frames.pred_depths[t - 1].reshape(1, self.h, self.w)
            # rigid=None = every pixel rigid (t <= 1): no H x W mask of ones is built, the kernels take a null pointer
            targets = FlowTargets(depth_prev, self.poses.pred_w2c[t - 1], self.frames.K, self.frames.flows_fw[t - 1], rigid)
        out = None
        rendered_last = None
        with self._phase("tracking.iterations", t):
            for it_ in range(self.tracking_iter):
                if it_ == self.tracking_iter - 1 and self.test_frame_quirks and t not in self._train_set():
                    with torch.no_grad():
                        rendered_last = self.poses.peek_pose(t).detach().clone()
                if self.fast is not None:
                    # the loss values are only logged once per frame (the reference prints them every iteration through
                    # .item(), i.e. a host sync per iteration)
                    out = self.fast.tracking_step(t, targets, None if all_rigid else rigid,
                                                  want_losses=self.trace is not None or it_ == self.tracking_iter - 1) + (None,)
                else:
                    out = tracking_step(self.pc, self.poses, self.frames, t, targets, rigid, fused=False)
                if self.trace is not None:
                    self.trace.append(("track", t, it_, float(out[0]), float(out[1]), float(out[2])))
        if rendered_last is not None:
            self.poses.pred_w2c[t] = rendered_last
        return out

    def _train_set(self):
        """frames.i_train as a set (built once: `t in ndarray` is a scan, and list(i_train) per global iteration a copy)"""
        arr = self.frames.i_train
        hit = self.__dict__.get("_train_cache")
        if hit is None or hit[0] is not arr:  # (the array object itself is kept: an id() alone can be recycled)
            lst = [int(i) for i in arr]
            hit = self._train_cache = (arr, frozenset(lst), lst)
        return hit[1]

    def _train_list(self):
        self._train_set()
        return self._train_cache[2]

    def progressive_run(self):
        n = len(self.frames.colors)
        self.poses.initialize_tracking_optimizer(self.tracking_iter)
        with torch.no_grad():
            self.poses.get_pose(0)
        train = self._train_set()
        for t in range(n):
            # staged sequences (fsgs_amd/staging.py): what the NEXT frame's tracking reads (colours, flows) travels while this
            # frame is optimised and stays resident until it is read, whatever keyframes the mapping iterations in between pull
            # through the lanes; the mono-depth is read by a frame's MAPPING only: asked for at the start of the frame's own
            # cycle (its 50 tracking iterations ahead), and only for a training frame
            self._prefetch(t + 1, monodeps=False)
            if t in train:
                self._prefetch(t, flows=False, colors=False)
            self.pc.update_learning_rate(self.iteration)
            if t > 0:
                with self._phase("frame.setup", t):
                    if self.fast is not None and hasattr(self.poses, "begin_frame"):
                        self.poses.begin_frame(t, self.tracking_iter)  # pose initialisation + fresh Adam / schedule: one launch
                    else:
                        if t > 1:
                            self.poses.initialize_pose(t)
                        else:
                            with torch.no_grad():
                                self.poses.r[..., t] = self.poses.r[..., t - 1]
                                self.poses.t[..., t] = self.poses.t[..., t - 1]
                        self.poses.initialize_tracking_optimizer(self.tracking_iter)
                with self._phase("frame.tracking", t):
                    loss, rgb, flow, _ = self.tracking(t)
                # the three logged values in ONE device-to-host copy (three float() calls are three synchronisations)
                self.log.append(("track", t) + tuple(float(v) for v in torch.stack((loss, rgb, flow)).tolist()))
            if t in train:
                if self.iteration % 1000 == 0:
                    self.pc.oneupSHdegree()
                it = self.first_mapping_iter if t == 0 else self.mapping_iter
                with self._phase("frame.mapping", t):
                    pkg = self._mapping(t, it, progressive=True, borrow=True)
                    self.frames.pred_depths[t] = self._stored_depth(pkg)  # (copies: pkg may alias the step driver's buffers)
                self.keyframes.append(t)
            elif self.frames.pred_depths[t] is None:
                if self.test_frame_quirks:  # never rendered upstream: the next frame's flow loss sees no valid depth
                    self.frames.pred_depths[t] = torch.zeros((self.h, self.w), dtype=torch.float32,
                                                             device=self._frames_device())
                else:
                    with torch.no_grad():
                        pkg = (render if self.fused else render_two_pass)(self.poses, t, self.pc, False, False)
                    self.frames.pred_depths[t] = self._stored_depth(pkg)
        self._release_prefetches()

    def _release_prefetches(self):
        """end of a phase: look-aheads that were fetched and never read (the frame behind the last one, a keyframe drawn for
        an iteration that did not run) must not keep their buffers pinned (staging.StagedLane.protected)"""
        rel = getattr(self.frames, "clear_protected", None)
        if rel is not None:
            rel()

    def _prefetch(self, t, flows=True, **lanes):
        pf = getattr(self.frames, "prefetch", None)
        if pf is not None and t is not None and 0 <= t < len(self.frames.colors):
            pf(int(t), flows=flows, **lanes)

    def _frames_device(self):
        return getattr(self.frames, "device", None) or self.frames.colors[0].device

    def _stored_depth(self, pkg):
        d = pkg["render_dep"].detach().float()
        if self.row0_depth_quirk:
            return d[0].expand(self.h, self.w).contiguous()  # (a copy: never aliases the step driver's buffer)
        if pkg.get("views_of_step_buffers"):
            return d.clone()  # what mapping() handed out is overwritten by the next step
        return d.contiguous()

    def global_run(self, iterations, first_iter=0, eval_every=5000, model_path=None, save_every=5000):
        """train.py:378-443: `for iter in range(self.first_iter, iterations + 1)` -- first_iter is 0 unless a checkpoint
        was loaded, so iteration 0 runs too (iterations + 1 mapping steps) and, 0 being a multiple of 1000, raises the SH
        degree at the very start of the global phase.  Every `eval_every` iterations (iter % 5000 == 0, so also at
        iteration 0) the test frames are evaluated as train.py:401-432 does (PSNR / SSIM into self.eval_log; the
        reference also writes comparison images and LPIPS: not reproduced); with a `model_path`, chkpnt{iter}.pth and
        poses{iter}.pth are written at iter % save_every == save_every - 1 in the reference's tuple layouts
        (train.py:437-443, checkpoint.py)."""
        self.pc.initialize_optimizer()
        self.eval_log = getattr(self, "eval_log", [])
        # the frame of iteration it + 1 is drawn while iteration it is set up (the same draws in the same order: nothing else
        # takes from self.rng in this loop), so that a staged sequence can start its copy one iteration ahead
        train = self._train_list()  # (random.choice over the same sequence draws the same elements as over list(i_train))
        nxt = int(self.rng.choice(train)) if iterations + 1 > int(first_iter) else None
        for it in range(int(first_iter), iterations + 1):
            ts = nxt
            nxt = int(self.rng.choice(train)) if it < iterations else None
            self._prefetch(nxt, flows=False)
            if it % 1000 == 0:
                self.pc.oneupSHdegree()
            self.pc.update_learning_rate(it)
            self.mapping(ts, 1, progressive=False, want_pkg=False)
            if eval_every and it % eval_every == 0 and len(self.frames.i_test):
                self.validation()
                self.eval_log.append((it, dict(self.last_validation)))
            if model_path and save_every and it % save_every == save_every - 1:
                from . import checkpoint

                checkpoint.save(model_path, it, self.pc, self.poses, np.asarray(self.frames.K, np.float32))
        self._release_prefetches()

    def validation(self):
        from . import metrics

        preds, gts = [], []
        with torch.no_grad():
            for i in self.frames.i_test:
                pkg = (render if self.fused else render_two_pass)(self.poses, int(i), self.pc, False, False)
                preds.append(pkg["render"].detach().cpu().numpy())
                gts.append(self.frames.colors[int(i)].detach().cpu().numpy())
        self.last_validation = ({"psnr": metrics.psnr(np.stack(gts), np.stack(preds)),
                                 "ssim": metrics.ssim(np.clip(np.stack(gts), 0, 1), np.clip(np.stack(preds), 0, 1))}
                                if preds else {"psnr": float("nan"), "ssim": float("nan")})
        # (rgb_evaluation also prints LPIPS: it needs the pretrained AlexNet weights of the `lpips` package, which cannot
        # be fetched here -- not reported)
        return self.last_validation["psnr"]

    def eval_pose(self):
        from . import metrics

        n = len(self.frames.colors)
        with torch.no_grad():
            if self.test_frame_quirks:
                # train.py:499-500 reads record_data['pred_w2c'] -- what the LAST get_pose of each frame recorded (for a
                # test frame the pose its last tracking iteration rendered with, before that iteration's optimizer step;
                # zeros for a frame no get_pose ever reached, scene/pose_optimizer.py:454) -- and records nothing itself
                pred = np.stack([np.zeros((4, 4), np.float32) if w is None else w.detach().cpu().numpy()
                                 for w in self.poses.pred_w2c[:n]])
            else:  # the poses as they stand now; an evaluation leaves the record alone either way
                pred = np.stack([self.poses.peek_pose(i).cpu().numpy() for i in range(n)])
        # train.py:492-506: every <data> run of the sequence is Sim(3)-aligned on its own and the three metrics are summed
        # with the runs' frame-count weights (dataset.read_sequence keeps data_ind / weights / gt_poses per run)
        runs = getattr(self.frames, "gt_poses", None)
        if runs:
            ind, wts = self.frames.data_ind, self.frames.weights
            total = np.zeros(3)
            for i, (_, gt_run) in enumerate(runs.items()):
                total += np.array(metrics.pose_metrics(pred[ind[i]:ind[i + 1]], gt_run)[1]) * wts[i]
            return [float(v) for v in total]
        gt = np.stack([np.asarray(g, np.float32) for g in self.frames.gt_w2c])
        return metrics.pose_metrics(pred, gt)[1]
