"""CPU-oracle harness (TEST INFRASTRUCTURE): the reference's training sequence -- FreeSurGS.tracking / mapping /
densification / progressive_run, train.py:154-376 -- restated on CPU tensors with

    render        = fsgs_amd.render.render_two_pass around the CPU oracle rasteriser (tests/ref_cpu.py)
    losses        = the torch statements pinned to the reference's golden vectors (losses.rgb_loss_torch, pearson_torch,
                    local_pearson_torch, flow.projection_flow_loss_torch, epipolar.*_torch)
    optimisers    = torch.optim.Adam / MultiStepLR exactly as the reference builds them
    densification = GaussianCloud.densify_and_prune (the reference's sequence, bit-for-bit vs tests/golden/densify.npz)

It exists to PIN the product harness (fsgs_amd.trainer.Runner on the HIP step driver): the per-iteration losses, the
poses and the cloud size after a densification that tests/golden/make_harness_golden.py records with it are what
tests/test_harness_pin_gpu.py demands of the GPU run on the same inputs."""
import contextlib
import random

import numpy as np
import torch

from fsgs_amd import epipolar, flow, losses, synth
from fsgs_amd.model import GaussianCloud
from fsgs_amd.render import render_two_pass
from fsgs_amd.trainer import LOSS_W_MAPPING, LOSS_W_TRACKING, FrameData, PoseTrack, settings_from_cam
from tests import ref_cpu


# How far the HIP harness may sit from the recorded trajectory in a per-iteration loss once the cloud has been densified: the
# children start with zero Adam moments, their first steps are lr * sign(gradient), and a last-bit difference in a near-zero
# gradient becomes a full-size step.  MEASURED (round 4): 200 runs of tests/test_harness_pin_gpu.py's schedule
# on one MI355X (profiles/r04_pin_deviation_200runs.txt; the runs differ by the arrival order of the blend's float atomics)
# leave the fixture by 0.47e-4 .. 3.32e-4 after the densification (median 1.8e-4, 99th percentile 3.25e-4; tracking losses
# <= 2.8e-4) and by <= 1.3e-5 before it.  The bound is 1.5 x the worst of those 200.  (Rounds 2-3 carried 5e-4, then 2e-3.)
POST_DENSIFY_RTOL = 5e-4
# Round 5: the backward of a small tile grid now runs four waves per tile (csrc/raster_kernels.h), i.e. ~2x as many,
# smaller float atomics per Gaussian: 180 further runs (profiles/r05_pin_deviation.txt) leave the fixture by 0.33e-4 .. 5.84e-4
# (median 1.6e-4 .. 2.8e-4 depending on the batch; forced back to one wave per tile: <= 3.1e-4 as before) -- the same size as
# the reference's own 0.9e-4 .. 5.4e-4.  The product path is therefore held to 1e-3 (1.7 x the worst of those), and the bound
# above is kept for FSGS_FLAG_DETERMINISTIC, whose ONE reproducible trajectory sits 2.73e-4 (mapping) / 2.88e-4 (tracking) away.
POST_DENSIFY_RTOL_PRODUCT = 1e-3
# ... and how far the REFERENCE moves against itself there (this CPU harness with 8 OpenMP threads in the oracle's backward
# against its own 1-thread fixture: 0.9e-4 .. 5.4e-4 over the thread counts tried, 6e-7 before the densification;
# tests/test_harness_pin_cpu.py): the yardstick that says the number above is not a property of the HIP path
REFERENCE_SELF_RTOL = 1e-3
# The pinned GLOBAL iterations (Runner.global_run behind the progressive phase: a fresh Adam, i.e. zero moments for EVERY
# Gaussian, not only for the children of a densification): 100 runs leave the fixture by up to 7.0e-4 in a per-iteration loss
# (median of the per-run maximum 3.0e-4, 90th percentile 4.8e-4; profiles/r04_pin_global_deviation_100runs.txt).  1.5 x the worst.
GLOBAL_PHASE_RTOL = 1.1e-3
# Round 5 (four-waves-per-tile backward on this small grid): 100 further runs leave it by up to 8.9e-4 (median of the per-run
# maximum 4.0e-4, 90th percentile 5.8e-4; profiles/r05_pin_global_deviation_100runs.txt) -> 1.5 x the worst for the product
# path; FSGS_FLAG_DETERMINISTIC's one trajectory sits 1.9e-4 away and is held to 5e-4.
GLOBAL_PHASE_RTOL_PRODUCT = 1.4e-3
GLOBAL_PHASE_RTOL_DETERMINISTIC = 5e-4


@contextlib.contextmanager
def deterministic_rng(seed):
    """Both harnesses consume random numbers on their own device (patch corners: torch.randint; split samples:
    torch.normal / torch.randn); the streams of a CPU and a GPU generator differ.  Inside this context those three draw
    from ONE seeded CPU generator and move the result to the device that was asked for, so a CPU run and a GPU run see
    the same numbers in the same order."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    o_randint, o_randn, o_normal = torch.randint, torch.randn, torch.normal

    def randint(low, high=None, size=None, **kw):
        if high is None:
            low, high = 0, low
        dev = kw.pop("device", None)
        kw.pop("generator", None)
        return o_randint(low, high, size, generator=g, **kw).to(dev or "cpu")

    def randn(*size, **kw):
        dev = kw.pop("device", None)
        kw.pop("generator", None)
        size = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
        return o_randn(tuple(size), generator=g, **kw).to(dev or "cpu")

    def normal(mean, std, **kw):
        z = o_randn(tuple(std.shape), generator=g, dtype=torch.float32).to(std.device)
        return mean + std * z

    torch.randint, torch.randn, torch.normal = randint, randn, normal
    try:
        yield
    finally:
        torch.randint, torch.randn, torch.normal = o_randint, o_randn, o_normal


class CpuHarness:
    """Runner's counterpart on CPU; same constructor meaning, same trace format."""

    def __init__(self, oracle, pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, seed=0,
                 row0_depth_quirk=True, densify_interval=300, opacity_reset_interval=3000, densify_until=15000):
        self.oracle, self.pc, self.poses, self.frames = oracle, pc, poses, frames
        self.tracking_iter, self.mapping_iter, self.first_mapping_iter = tracking_iter, mapping_iter, first_mapping_iter
        self.iteration, self.keyframes, self.rng = 0, [], random.Random(seed)
        self.row0_depth_quirk = row0_depth_quirk
        self.densify_interval, self.opacity_reset_interval, self.densify_until = densify_interval, opacity_reset_interval, densify_until
        self.trace = []
        self.h, self.w = (int(v) for v in frames.colors[0].shape[-2:])

    def render(self, t, gs_grad, cam_grad):
        with ref_cpu.oracle_backend(self.oracle):
            return render_two_pass(self.poses, t, self.pc, gs_grad=gs_grad, cam_grad=cam_grad)

    # ---- train.py:297-316 ----
    def densification(self):
        it = self.iteration
        if it % self.densify_interval == 0 and it < self.densify_until:
            self.pc.densify_and_prune(self.pc.opt.densify_grad_threshold, 0.05, 20 if it > 4000 else None)
            self.trace.append(("densify", it, self.pc.num_points))
        if it % self.opacity_reset_interval == 0:
            self.pc.reset_opacity()

    # ---- train.py:213-295 ----
    def mapping(self, cur_t, mapping_iter, progressive):
        views = 2 if (progressive and cur_t != 0) else 1
        self.pc.optimizer.zero_grad(set_to_none=True)
        pkg = None
        for _ in range(mapping_iter):
            self.iteration += 1
            ts = [self.rng.choice(self.keyframes), cur_t] if views == 2 else [cur_t]
            loss, first = 0, None
            for k, t in enumerate(ts):
                pkg = self.render(t, gs_grad=True, cam_grad=False)
                rgb = losses.rgb_loss_torch(pkg["render"], self.frames.colors[t]) * LOSS_W_MAPPING["rgb"]
                pear = losses.pearson_torch(self.frames.monodeps[t], pkg["render_dep"])
                lp = losses.local_pearson_torch(self.frames.monodeps[t], pkg["render_dep"], 128, 0.5)
                loss = loss + rgb + pear * LOSS_W_MAPPING["pearson"] + lp * LOSS_W_MAPPING["local_pearson"]
                if k == 0:
                    first = pkg
            loss.backward()
            self.trace.append(("map", self.iteration, tuple(ts), float(loss)))
            with torch.no_grad():
                vis = first["visibility_filter"]  # train.py:260-263,298-303: statistics from view 0
                self.pc.variables["max_radii2D"][vis] = torch.max(self.pc.variables["max_radii2D"][vis],
                                                                  first["radii"][vis].float())
                self.pc.add_densification_stats(first["viewspace_points"].grad, vis)
                self.densification()
                self.pc.optimizer.step()
                self.pc.optimizer.zero_grad(set_to_none=True)
        return pkg  # the render of the LAST view of the last iteration (train.py:291)

    # ---- train.py:154-210 ----
    def tracking(self, t):
        rigid = None
        if t > 1:
            with torch.no_grad():  # get_fundamental_matrix asks the network directly: pred_w2c is NOT refreshed
                # (scene/pose_optimizer.py:640-648; the same values as get_pose for every frame that has been mapped)
                Fm = epipolar.fundamental_from_w2c(self.poses.peek_pose(t - 2), self.poses.peek_pose(t - 1), self.frames.K)
            rigid = epipolar.rigid_mask_torch(epipolar.sampson_distance_torch(self.frames.flows_fw[t - 2], Fm))
        depth_prev = self.frames.pred_depths[t - 1].reshape(1, self.h, self.w)
        w2c_prev = self.poses.pred_w2c[t - 1]
        for it_ in range(self.tracking_iter):
            pkg = self.render(t, gs_grad=False, cam_grad=True)
            mask = pkg["render_dep"] > 0
            if rigid is not None:
                mask = mask * rigid
            rgb = LOSS_W_TRACKING["rgb"] * losses.rgb_loss_torch(pkg["render"], self.frames.colors[t], mask=mask.unsqueeze(0))
            fl = LOSS_W_TRACKING["flow"] * flow.projection_flow_loss_torch(depth_prev, w2c_prev, pkg["render_w2c"],
                                                                          self.frames.K, self.frames.flows_fw[t - 1], rigid)
            loss = fl + rgb
            loss.backward()
            self.poses.scheduler.step()  # before optimizer.step(), as train.py:189,194
            with torch.no_grad():
                self.poses.optimizer.step()
                self.poses.optimizer.zero_grad(set_to_none=True)
            self.trace.append(("track", t, it_, float(loss), float(rgb), float(fl)))

    def _tracking_optimizer(self):
        """Adam(lr .01, eps 1e-15) + MultiStepLR(milestones 0,16,32,48 for 50 iterations; gamma .5)
        (scene/pose_optimizer.py:489-496)."""
        p = self.poses
        p.optimizer = torch.optim.Adam([{"params": p.r, "lr": 0.01}, {"params": p.t, "lr": 0.01}], lr=0.001, eps=1e-15)
        step = int(self.tracking_iter / 3)
        p.scheduler = torch.optim.lr_scheduler.MultiStepLR(p.optimizer, milestones=list(range(0, int(self.tracking_iter), step)),
                                                           gamma=0.5)

    # ---- train.py:318-345 ----
    def progressive_run(self):
        n = len(self.frames.colors)
        with torch.no_grad():
            self.poses.get_pose(0)
        for t in range(n):
            self.pc.update_learning_rate(self.iteration)
            if t > 0:
                if t > 1:
                    self.poses.initialize_pose(t)
                else:
                    with torch.no_grad():
                        self.poses.r[..., t] = self.poses.r[..., t - 1]
                        self.poses.t[..., t] = self.poses.t[..., t - 1]
                self._tracking_optimizer()
                self.tracking(t)
            if t in self.frames.i_train:
                if self.iteration % 1000 == 0:
                    self.pc.oneupSHdegree()
                pkg = self.mapping(t, self.first_mapping_iter if t == 0 else self.mapping_iter, progressive=True)
                d = pkg["render_dep"].detach().float()
                self.frames.pred_depths[t] = d[0].expand(self.h, self.w).contiguous() if self.row0_depth_quirk else d.contiguous()
                self.keyframes.append(t)
            elif self.frames.pred_depths[t] is None:
                # a TEST frame (every 8th) is never mapped: record_data['pred_depths'][t] keeps its initial zeros, the next
                # frame's flow loss finds no valid pixel and is the constant 0 (train.py:333-343,
                # scene/pose_optimizer.py:176-186); its recorded pose is the one its last tracking iteration rendered with
                self.frames.pred_depths[t] = torch.zeros((self.h, self.w), dtype=torch.float32)


    # ---- train.py:378-443 (the training part: evaluation / checkpoints are not on the path) ----
    def global_run(self, iterations, first_iter=0):
        """`for iter in range(first_iter, iterations + 1)`: a fresh Adam with default eps (scene/gaussian_model.py:372-378), per
        iteration a random training frame, the SH degree raised when iter % 1000 == 0 (so at iteration 0), the xyz learning
        rate of that iteration, ONE one-view mapping iteration -- self.iteration keeps counting from the progressive phase, so
        the densification schedule does too (train.py:236,305)."""
        self.pc.initialize_optimizer(fused=False)
        for it in range(int(first_iter), iterations + 1):
            ts = int(self.rng.choice(list(self.frames.i_train)))
            if it % 1000 == 0:
                self.pc.oneupSHdegree()
            self.pc.update_learning_rate(it)
            self.mapping(ts, 1, progressive=False)


# ---- the pinned inputs: a tiny synthetic sequence made entirely on CPU ----------------------------------------------
def make_inputs(oracle, W=256, H=192, n_frames=3, P_scene=3000, ratio=0.02, seed=0):
    """A hidden opaque scene rendered along a smooth trajectory by the CPU reference render -> per frame colours
    (8-bit), mono-depth (affine-normalised to [0.5,1.5], fp16) and forward flow (fp16), plus the learner's first-frame
    cloud (GaussianModel.initialize_first_timestep, scene/gaussian_model.py:237-258, scales from the oracle's exact
    3-NN).  Everything is quantised BEFORE it is used, so the arrays written to the fixture are bit-for-bit what both
    harnesses consume.  -> dict of numpy arrays."""
    from fsgs_amd.sequence import _rot_to_quat, gt_trajectory

    cam = synth.make_camera(W, H)
    knn = lambda pts: oracle.knn_meandist2(pts)
    sc = dict(synth.init_scene(W, H, P_scene, seed=seed, knn_fn=knn))
    sc["_opacity"] = np.full_like(sc["_opacity"], 3.0)
    sc["_scaling"] = sc["_scaling"] + 0.35
    gt = GaussianCloud(sc, sh_degree=3, device="cpu")
    gt.cam = settings_from_cam(cam, "cpu")
    w2cs = gt_trajectory(n_frames, seed=seed)
    poses = PoseTrack(n_frames, "cpu")
    for i, m in enumerate(w2cs):
        poses.set_pose(i, _rot_to_quat(m[:3, :3]), m[:3, 3])
    colors, depths = [], []
    with torch.no_grad(), ref_cpu.oracle_backend(oracle):
        for i in range(n_frames):
            pkg = render_two_pass(poses, i, gt, gs_grad=False, cam_grad=False)
            colors.append(pkg["render"].clamp(0, 1))
            depths.append(pkg["render_dep"])
    colors_u8 = np.stack([(c * 255.0 + 0.5).to(torch.uint8).numpy() for c in colors])
    mono = np.stack([(((d - d.min()) / (d.max() - d.min())) + 0.5).numpy() for d in depths]).astype(np.float16)
    K = torch.tensor(cam["K"], dtype=torch.float32)
    vv, uu = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    flows = []
    for i in range(n_frames - 1):
        d = depths[i]
        ci = torch.stack([(uu - K[0, 2]) / K[0, 0] * d, (vv - K[1, 2]) / K[1, 1] * d, d, torch.ones_like(d)], 0).reshape(4, -1)
        rel = torch.tensor(w2cs[i + 1] @ np.linalg.inv(w2cs[i]), dtype=torch.float32)
        p = K @ (rel @ ci)[:3]
        flows.append(torch.stack([(p[0] / (p[2] + 1e-5)).reshape(H, W) - uu, (p[1] / (p[2] + 1e-5)).reshape(H, W) - vv], 0).numpy())
    flows = np.stack(flows).astype(np.float16)
    # the learner's cloud from frame 0, from the QUANTISED inputs
    c0 = torch.tensor(colors_u8[0].astype(np.float32) / 255.0)
    m0 = torch.tensor(mono[0].astype(np.float32))
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(H * W, generator=g)[: int(ratio * H * W)].sort().values
    v_, u_ = perm // W, perm % W
    z = m0[v_, u_]
    xyz = torch.stack([(u_.float() - K[0, 2]) / K[0, 0] * z, (v_.float() - K[1, 2]) / K[1, 1] * z, z], 1).numpy().astype(np.float32)
    rgb = c0[:, v_, u_].T.numpy()
    dist2 = np.maximum(oracle.knn_meandist2(xyz), 1e-7).astype(np.float32)
    Pn = xyz.shape[0]
    return {
        "colors_u8": colors_u8, "monodeps_f16": mono, "flows_fw_f16": flows, "K": np.asarray(cam["K"], np.float64),
        "gt_w2c": np.stack(w2cs).astype(np.float32), "W": W, "H": H,
        "_xyz": xyz, "_features_dc": ((rgb - 0.5) / synth.SH_C0).reshape(Pn, 1, 3).astype(np.float32),
        "_features_rest": np.zeros((Pn, 15, 3), np.float32),
        "_opacity": np.full((Pn, 1), np.log(0.1 / 0.9), np.float32),
        "_scaling": np.repeat(np.log(np.sqrt(dist2))[:, None], 3, 1).astype(np.float32),
        "_rotation": np.tile(np.array([[1.0, 0, 0, 0]], np.float32), (Pn, 1)),
        "scene_radius": float(mono[0].astype(np.float32).max()) / 2.0,
    }


def load_inputs(fx, device):
    """fixture arrays -> (GaussianCloud with its progressive-run optimizer NOT yet built, PoseTrack, FrameData)."""
    from fsgs_amd.model import PARAM_NAMES

    W, H = int(fx["W"]), int(fx["H"])
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=device)
    colors = [t(c.astype(np.float32) / 255.0) for c in fx["colors_u8"]]
    mono = [t(m.astype(np.float32)) for m in fx["monodeps_f16"]]
    flows = [t(f.astype(np.float32)) for f in fx["flows_fw_f16"]]
    frames = FrameData(colors, mono, flows_fw=flows, K=np.asarray(fx["K"], np.float64), gt_w2c=[m for m in fx["gt_w2c"]])
    # (copies: on CPU a tensor made from a numpy array shares its memory, and Adam updates parameters in place)
    pc = GaussianCloud({k: np.array(fx[k], dtype=np.float32, copy=True) for k in PARAM_NAMES}, sh_degree=3, device=device,
                       scene_radius=float(fx["scene_radius"]))
    pc.cam = settings_from_cam(synth.make_camera(W, H), device)
    poses = PoseTrack(len(colors), device)
    return pc, poses, frames


# what the pinned run does (shared by the fixture script and the GPU test)
# (global_iters: Runner.global_run(6) behind the progressive phase = 7 iterations, counter 16 .. 22: the first one densifies)
PIN = dict(tracking_iter=5, mapping_iter=5, first_mapping_iter=5, densify_interval=8, rng_seed=11, seed=0, global_iters=6)


# ---- BASELINE.json configs[0] as stated: 8 frames, 640x512, 20 k initial Gaussians, the reference's own schedule ----------
C1 = dict(W=640, H=512, n_frames=8, P=20_000, tracking_iter=50, mapping_iter=30, first_mapping_iter=200,
          densify_interval=300, rng_seed=11, seed=0, P_scene=60_000, input_seed=3)


def make_c1_inputs(oracle):
    """the C1 sequence, regenerated from its seed wherever the oracle runs (~10 s): a fixture of 8 frames at 640x512 would be
    7.8 MB compressed; tests/golden/harness_c1.npz stores coarse fingerprints of it instead (c1_input_stats)"""
    ratio = C1["P"] / float(C1["W"] * C1["H"])
    return make_inputs(oracle, W=C1["W"], H=C1["H"], n_frames=C1["n_frames"], P_scene=C1["P_scene"], ratio=ratio,
                       seed=C1["input_seed"])


def c1_input_stats(fx):
    """coarse fingerprints of the regenerated sequence: per-frame means of colours / mono-depth / flow, the first points"""
    return np.concatenate([fx["colors_u8"].astype(np.float64).mean(axis=(1, 2, 3)) / 255.0,
                           fx["monodeps_f16"].astype(np.float64).mean(axis=(1, 2)),
                           np.abs(fx["flows_fw_f16"].astype(np.float64)).mean(axis=(1, 2, 3)),
                           fx["_xyz"][:16].astype(np.float64).reshape(-1), fx["_scaling"][:16, 0].astype(np.float64)])


def c1_outcome(trace, pc, poses, frames, render_fn):
    """What a C1 run is judged by, from a harness's trace (CpuHarness and fsgs_amd.trainer.Runner share the format) and its
    final state: per tracked frame the first / last iteration's losses, per mapped frame the mean and last mapping loss,
    the densification record, the tracked poses, RPE / ATE against the ground truth (train.py:492-506), PSNR of the test
    frame(s) at their tracked pose (train.py:401-432).  render_fn(t) -> the rendered [3,H,W] image of frame t."""
    from fsgs_amd import metrics

    n = len(frames.colors)
    maps = [e for e in trace if e[0] == "map"]
    tracks = [e for e in trace if e[0] == "track"]
    out = {}
    out["track_last"] = np.array([[e[3], e[4], e[5]] for e in tracks if e[2] == C1["tracking_iter"] - 1], np.float64)
    out["track_first"] = np.array([[e[3], e[4], e[5]] for e in tracks if e[2] == 0], np.float64)
    bounds, it = [], 0
    train = set(int(i) for i in frames.i_train)
    for t in range(n):
        if t in train:
            k = C1["first_mapping_iter"] if t == 0 else C1["mapping_iter"]
            bounds.append((t, it, it + k))
            it += k
    ml = np.array([e[3] for e in maps], np.float64)
    out["map_mean"] = np.array([[t, ml[a:b].mean(), ml[b - 1]] for t, a, b in bounds], np.float64)
    out["densify"] = np.array([[e[1], e[2]] for e in trace if e[0] == "densify"], np.int64)
    out["final_P"] = pc.num_points
    out["pose_r"] = poses.r.detach().cpu().numpy()
    out["pose_t"] = poses.t.detach().cpu().numpy()
    from fsgs_amd.pose import pose_to_w2c

    with torch.no_grad():  # (the torch statement on host copies: the same arithmetic for both harnesses)
        r_, t_ = poses.r.detach().cpu(), poses.t.detach().cpu()
        pred = np.stack([pose_to_w2c(r_, t_, i).numpy() for i in range(n)])
    gt = np.stack([np.asarray(g, np.float32) for g in frames.gt_w2c])
    out["pose_metrics"] = np.array(metrics.pose_metrics(pred, gt)[1], np.float64)  # rpe_t, rpe_r (deg), ate
    ps = []
    with torch.no_grad():
        for i in frames.i_test:
            img = render_fn(int(i))
            ps.append(metrics.psnr(frames.colors[int(i)].detach().cpu().numpy()[None], img.detach().cpu().numpy()[None]))
    out["psnr_test"] = np.array(ps, np.float64)
    return out
