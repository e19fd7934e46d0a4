#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native Free-SurGS hot path.

    python bench.py [--gpus N --steps K --warmup W]           (N > 1 without a launcher: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): train iterations/s at 1280x1024 with 300k Gaussians (config C2).
A "step" is one mapping iteration of the reference (train.py:236-272, global_run view = 1):
render (RGB pass + depth/silhouette pass) -> 5*rgb_loss + 0.05*pearson + 0.15*local_pearson ->
backward through both passes -> densification statistics -> Adam step on all 59 floats/Gaussian.
With N ranks every rank optimises its own camera of the sequence over the shared cloud and the
Gaussian gradients are all-reduced (RCCL) each step: N views per step, weak scaling.

One JSON line on rank 0, with `roofline` (dominant kernel: blend_bwd, HIP events on the launching
stream) and `cpu_baseline` (the CPU oracle's rasteriser fwd+bwd, rank 0 / N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "free-surgs_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# scene variants of the trained-like cloud: scale multiplier of every Gaussian.  "dense" brings the UPSTREAM pair count
# (sum of rect tiles, what SURVEY.md s8d's nominal 3.0 M at C2 refers to) to ~10 tiles per Gaussian
SCENES = {"default": 1.0, "dense": 1.62}
CONFIGS = {
    # name: (W, H, P, scene)
    "C1": (640, 512, 20_000, "init"),
    "C2": (1280, 1024, 300_000, "trained"),
    "C4": (1920, 1080, 1_000_000, "trained"),
    "C4x4": (1920, 1080, 4_000_000, "trained"),  # size stress only (tests); not a BASELINE.json configuration
    # tile-grid sizes between C1 and C2, to place the crossover of the two blend-kernel flavours (scripts/gpu_blend_variants.sh);
    # not BASELINE.json configurations
    "X1": (832, 640, 60_000, "trained"),    # 2080 tiles
    "X2": (960, 768, 120_000, "trained"),   # 2880 tiles
    "X3": (1088, 896, 200_000, "trained"),  # 3808 tiles
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
VALU_SIMDS, VALU_CYCLES_PER_WAVE_INST, VALU_CLOCK_HZ = 1024, 2.0, 2.4e9  # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md (2.4 GHz = the peak)


def valu_frac(wave_insts, kernel_seconds, clock_hz=VALU_CLOCK_HZ):
    """fraction of the spec VALU issue rate: (wave-instructions per SIMD x 2 cycles / shader clock) / kernel time"""
    return (wave_insts / VALU_SIMDS) * VALU_CYCLES_PER_WAVE_INST / clock_hz / kernel_seconds


def blend_flavours(W, H):
    from fsgs_amd import _lib, rasterizer

    lib = _lib.load()
    flags = {"auto": 0, "one": _lib.FSGS_FLAG_BLEND_ONE_WAVE, "quad": _lib.FSGS_FLAG_BLEND_QUAD_WAVES}[rasterizer.blend_variant()]
    q = lambda bwd, pose: int(lib.fsgs_blend_waves_per_tile(int(W), int(H), int(flags), bwd, pose))
    return {"forward": q(0, 0), "backward": q(1, 0), "pose_only_backward": q(1, 1), "variant": rasterizer.blend_variant()}


def load_issue_rates():
    """newest profiles/r*_issue_rates.json (scripts/make_issue_rates_json.py): the shader clock MEASURED inside the blend kernels
    and the issue micro-benchmark's ns per instruction and SIMD at 1..8 resident waves -- offline constants like the PMC counters"""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_rates.json")))
    if not files:
        return None, None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


def valu_block(ent, kernel_name, family, avg_s, issue, issue_src):
    """What actually bounds a blend kernel -- instruction issue -- in three numbers:
      valu_frac                 share of the SPEC issue rate (one wave64 VALU instruction per SIMD per 2 cycles) at the shader clock
                                MEASURED inside the kernel (s_memtime / s_memrealtime per tile, diagnostics flavour); rounds 1-5
                                assumed 2.4 GHz
      valu_frac_of_achievable   share of what scripts/ubench/issue_clock.hip reaches with the backward body's instruction MIX
                                (1 exp + 1 rcp + 3 cmp/cndmask + FMAs + broadcast LDS reads) at the number of waves per SIMD this
                                kernel's registers allow -- the roof a kernel of this mix and occupancy really has
    (rounds 4-5 also printed `valu_busy` = SQ_ACTIVE_INST_VALU x 4 / SIMDs / clock / time: the counter reads 4.0-4.2 per
    instruction on EVERY kernel, i.e. it is a per-instruction constant, not pipe occupancy -- dropped)"""
    wi = ent.get("valu_wave_insts")
    if not wi:
        return None
    per_simd = wi / VALU_SIMDS
    out = {"wave_insts": wi, "salu_wave_insts": ent.get("salu_wave_insts"), "ns_per_valu_inst_per_simd": avg_s * 1e9 / per_simd,
           "sq": ent.get("sq")}
    clock_hz, clock_src = VALU_CLOCK_HZ, "assumed peak clock (no profiles/r*_issue_rates.json)"
    if issue:
        mhz = issue.get("shader_clock_mhz", {}).get(family)
        if mhz:
            clock_hz, clock_src = mhz * 1e6, "%s: delta s_memtime / delta s_memrealtime inside the kernel" % issue_src
        w = None
        for k, v in issue.get("waves_per_simd", {}).items():
            if kernel_name.startswith(k) or k.startswith(kernel_name):
                w = v["waves_per_simd"]
        rows = issue.get("ubench", {}).get("blend", {})
        if w and str(w) in rows:
            ns = rows[str(w)]["ns_per_inst_per_simd"]
            out.update({"waves_per_simd": w, "ubench_ns_per_inst_per_simd_at_that_occupancy": ns,
                        "valu_frac_of_achievable": per_simd * ns * 1e-9 / avg_s})
        out["ubench_blend_mix_ns_per_inst_per_simd"] = {k: v["ns_per_inst_per_simd"] for k, v in sorted(rows.items(), key=lambda kv: int(kv[0]))}
    out.update({"valu_frac": valu_frac(wi, avg_s, clock_hz), "shader_clock_mhz": clock_hz / 1e6, "shader_clock_source": clock_src})
    return out


def build_problem(cfg_name, device, rank, world, n_frames=8, scene="default", texture=0.0):
    from fsgs_amd import synth
    from fsgs_amd.model import GaussianCloud
    from fsgs_amd.trainer import FrameData, PoseTrack, settings_from_cam
    from simple_knn._C import distCUDA2

    W, H, P, kind = CONFIGS[cfg_name]
    knn = lambda pts: distCUDA2(torch.tensor(pts, device=device)).cpu().numpy()
    if kind == "init":
        sc = synth.init_scene(W, H, P, seed=0, knn_fn=knn)
    else:
        sc = synth.trained_like_scene(W, H, P, seed=0, knn_fn=knn)
        if SCENES[scene] != 1.0:
            sc = dict(sc)
            sc["_scaling"] = (sc["_scaling"] + np.log(SCENES[scene])).astype(np.float32)
    cam = synth.make_camera(W, H)
    pc = GaussianCloud(sc, sh_degree=3, device=device, scene_radius=float(sc["depth_map"].max()) / 2.0)
    pc.cam = settings_from_cam(cam, device)
    pc.active_sh_degree = 3 if kind == "trained" else 0
    pc.training_setup(eps=1e-8)  # global_run re-creates Adam with default eps (scene/gaussian_model.py:378)
    poses = PoseTrack(n_frames, device)
    rng = np.random.default_rng(1234)
    for i in range(1, n_frames):  # a short camera sweep around frame 0
        q = np.array([1.0, 0, 0, 0]) + 0.01 * rng.standard_normal(4)
        poses.set_pose(i, q, 0.02 * rng.standard_normal(3))
    # per-frame targets resident in HBM: a colour image and a mono-depth map
    img = torch.tensor(sc["image"], device=device)
    dep = torch.tensor(sc["depth_map"], device=device)
    if texture > 0.0:
        # Detail the cloud does not have yet, in a third of the image (a fine oblique grating): what an under-reconstructed
        # region looks like to the optimiser -- view-space gradients there cross the reference's densification threshold
        # (2e-4, train.py:307) and the cloud GROWS by clone / split instead of only being pruned (--densify-every)
        v, u = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                              torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
        region = ((u > 0.15 * W) & (u < 0.65 * W) & (v > 0.2 * H) & (v < 0.85 * H)).float()
        grating = torch.sin(2 * np.pi * (u + 0.5 * v) / 9.0) * torch.cos(2 * np.pi * (v - 0.3 * u) / 13.0)
        img = (img + texture * region * grating.unsqueeze(0) * torch.tensor([1.0, -0.7, 0.5], device=device).reshape(3, 1, 1)).clamp(0, 1)
    frames = FrameData([img + 0.01 * i for i in range(n_frames)], [dep * (1 + 0.01 * i) for i in range(n_frames)])
    return pc, poses, frames, cam, sc


def upstream_pairs(stepper, W, H):
    """Sum over the visible Gaussians of the tiles in their 3-sigma rect: UPSTREAM's num_rendered (what SURVEY.md s8d's
    nominal R refers to); the HIP path's own R counts only the pairs that survive the exact footprint test."""
    import ctypes as C

    from fsgs_amd import _lib

    last = getattr(stepper, "last", None)
    if not last or "state" not in last:
        return None
    P = int(last["P"])  # of that forward (a densification may have resized the cloud since)
    off = (C.c_size_t * 9)()
    _lib.check(_lib.load().fsgs_render_state_layout(P, W, H, int(last["max_pairs"]), off), "fsgs_render_state_layout")
    xy = last["state"][off[0]:off[0] + 8 * P].view(torch.float32).reshape(P, 2)
    r = last["radii_last"][:P].float()
    vis = r > 0
    gx, gy = (W + 15) // 16, (H + 15) // 16
    lo = lambda c, n: torch.clamp(torch.trunc((c - r) / 16), 0, n)
    hi = lambda c, n: torch.clamp(torch.trunc((c + r + 15) / 16), 0, n)
    area = (hi(xy[:, 0], gx) - lo(xy[:, 0], gx)) * (hi(xy[:, 1], gy) - lo(xy[:, 1], gy))
    return int(area[vis].sum().item())


def cpu_baseline(sc, cam, pc_sh_degree, seconds_budget=30.0):
    """The oracle (kind "port": the reference has no CPU path and its rasteriser source is absent) timed on
    the host cores: rasteriser fwd + bwd of the RGB pass and of the depth/silhouette pass = the
    rasteriser part of ONE step (losses and Adam are left out, which only flatters the CPU)."""
    from fsgs_amd import synth
    from oracle.fsgs_oracle import Oracle, usable_cores

    o = Oracle(np.float32)
    cores = min(usable_cores(), o.max_threads())
    o.set_threads(cores)
    s, r, op = synth.activate(sc)
    col = np.clip(sc["_features_dc"][:, 0, :] * synth.SH_C0 + 0.5, 0, None).astype(np.float32)
    z = sc["_xyz"][:, 2:3]
    dcol = np.concatenate([z, np.ones_like(z), z * z], 1).astype(np.float32)
    H, W = cam["image_height"], cam["image_width"]
    dL = (np.random.default_rng(0).uniform(-1, 1, (3, H, W)) / (3 * H * W)).astype(np.float32)
    t0 = time.time()
    n = 0
    R = 0
    t_fwd = t_bwd = 0.0
    while True:
        for c in (col, dcol):
            ta = time.time()
            img, dep, radii, st = o.raster_forward(cam, sc["_xyz"], c, op.reshape(-1), s, r)
            tb = time.time()
            o.raster_backward(st, dL)
            t_fwd += tb - ta
            t_bwd += time.time() - tb
            R = st.num_rendered
        n += 1
        if time.time() - t0 > seconds_budget * 0.5 or n >= 3:
            break
    dt = (time.time() - t0) / n
    out = {"value": 1.0 / dt, "unit": "iters/s", "cores": cores, "kind": "port",
           "raster_fwd_ms_per_pass": 1e3 * t_fwd / (2 * n), "raster_bwd_ms_per_pass": 1e3 * t_bwd / (2 * n),
           "sample": "%d full step(s) of the rasteriser part only (2 passes fwd+bwd, %dx%d, P=%d, R=%d), "
                     "oracle/raster_oracle.c with OpenMP" % (n, W, H, len(sc["_xyz"]), R)}

    # ---- the rest of SURVEY.md s8(d)'s CPU set: one thread, C1, and the torch-CPU losses -------------
    def one_pass(scene, camera, colours, threads):
        o.set_threads(threads)
        s_, r_, op_ = synth.activate(scene)
        h, w = camera["image_height"], camera["image_width"]
        g = (np.random.default_rng(0).uniform(-1, 1, (3, h, w)) / (3 * h * w)).astype(np.float32)
        ta = time.time()
        st_ = o.raster_forward(camera, scene["_xyz"], colours, op_.reshape(-1), s_, r_)[3]
        tb = time.time()
        o.raster_backward(st_, g)
        return {"raster_fwd_ms_per_pass": 1e3 * (tb - ta), "raster_bwd_ms_per_pass": 1e3 * (time.time() - tb),
                "threads": threads, "num_rendered_rect_rule": int(st_.num_rendered)}

    try:
        if len(sc["_xyz"]) <= 400_000:  # (C4 on one thread is ~40 s: not inside a bench run)
            out["one_thread"] = one_pass(sc, cam, col, 1)
            out["one_thread"]["sample"] = ("ONE RGB pass fwd + bwd on one thread at this configuration; a step's rasteriser part is two "
                                           "such passes, i.e. ~%.0f ms extrapolated" % (2 * (out["one_thread"]["raster_fwd_ms_per_pass"] +
                                                                                             out["one_thread"]["raster_bwd_ms_per_pass"])))
        w1, h1, p1, _ = CONFIGS["C1"]
        if (W, H, len(sc["_xyz"])) != (w1, h1, p1):
            sc1 = synth.init_scene(w1, h1, p1, seed=0)  # (scales from the oracle's own KNN: this leg must not touch the GPU)
            cam1 = synth.make_camera(w1, h1)
            col1 = np.clip(sc1["_features_dc"][:, 0, :] * synth.SH_C0 + 0.5, 0, None).astype(np.float32)
            out["c1"] = {"all_cores": one_pass(sc1, cam1, col1, cores), "one_thread": one_pass(sc1, cam1, col1, 1),
                         "sample": "ONE RGB pass fwd + bwd at C1 (640x512, 20 000 init Gaussians)"}
        out["losses_torch_cpu_ms"] = losses_torch_cpu(H, W, cores)
    except Exception as e:  # a reported baseline, never a reason to lose the line
        out["extras_error"] = "%s: %s" % (type(e).__name__, e)
    finally:
        o.set_threads(cores)
    return out


def losses_torch_cpu(H, W, threads):
    """the build's own plain-torch restatement of the three mapping losses (fsgs_amd/losses.py *_torch: utils/loss_utils.py:41-127)
    forward + backward on the host cores -- the CPU counterpart of the fused loss kernels"""
    from fsgs_amd import losses

    old = torch.get_num_threads()
    torch.set_num_threads(max(1, int(threads)))
    try:
        g = torch.Generator().manual_seed(0)
        img = torch.rand(3, H, W, generator=g, requires_grad=True)
        gt = torch.rand(3, H, W, generator=g)
        dep = (torch.rand(H, W, generator=g) + 0.5).requires_grad_(True)
        mono = torch.rand(H, W, generator=g) + 0.5
        corners = losses.draw_patch_corners(H, W, 128, 0.5, "cpu")
        res = {}
        for name, fn in (("rgb_l1_ssim", lambda: losses.rgb_loss_torch(img, gt)),
                         ("pearson_global_and_local", lambda: losses.pearson_torch(mono, dep) * 0.05 +
                          losses.local_pearson_torch(mono, dep, 128, 0.5, corners) * 0.15)):
            fn().backward()  # warm-up
            t0 = time.time()
            n = 0
            while n < 3:
                fn().backward()
                n += 1
            res[name] = 1e3 * (time.time() - t0) / n
        res["threads"] = int(threads)
        res["what"] = "forward + backward, %dx%d, torch CPU" % (W, H)
        return res
    finally:
        torch.set_num_threads(old)


def harness_extra(cfg_name, device, scene="default"):
    """The loop BASELINE.json's metric names ("train iters/sec ... joint pose+GS optimisation as in train.py"): the harness
    counterpart of FreeSurGS.progressive_run / global_run (train.py:318-443; fsgs_amd/trainer.py:Runner) on this
    configuration's cloud, timed inside this process behind the headline loop.

      progressive: Runner.progressive_run over the 8 synthetic frames -- frame 0: 200 one-view mapping iterations; every
        further frame: pose initialisation + fresh pose Adam, the Sampson rigid mask (t >= 2), the flow targets, 50 tracking
        iterations, and for a training frame 30 two-view mapping iterations (+ densify_and_prune when the iteration counter
        reaches 300).  Phase times from Runner(profile=True): each phase bracketed by a device synchronisation.
      global: Runner.global_run, 300 iterations (fresh Adam, random training frame per iteration, xyz learning-rate
        schedule, densification statistics, densify_and_prune at iteration 300), one synchronisation at either end.
    """
    from fsgs_amd.trainer import Runner

    W, H, P, _ = CONFIGS[cfg_name]

    def problem():
        pc, poses, frames, cam, sc = build_problem(cfg_name, device, 0, 1, scene=scene)
        n = len(frames.colors)
        zero_flow = torch.zeros((2, H, W), dtype=torch.float32, device=device)  # a static camera's flow: timing only
        frames.flows_fw = [zero_flow] * (n - 1)
        frames.K = cam["K"]
        from fsgs_amd.trainer import PoseTrack

        return pc, PoseTrack(n, device), frames  # (every pose starts at the identity, as progressive_run expects)

    out = {}
    # ---- progressive_run, profiled ----
    pc, poses, frames = problem()
    run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, profile=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.progressive_run()
    torch.cuda.synchronize()
    wall_profiled = (time.perf_counter() - t0) * 1e3
    ph = run.phase_ms
    by = lambda tag: {t: ms for k, t, ms in ph if k == tag}
    setup, track, mapp = by("frame.setup"), by("frame.tracking"), by("frame.mapping")
    rigid, targets, iters = by("tracking.rigid_mask"), by("tracking.flow_targets"), by("tracking.iterations")
    dens = [(t, ms) for k, t, ms in ph if k == "densify_and_prune"]
    # steady frames: t >= 2 (frame 1 pays the one-time costs: first launches of the flow kernels, torch.sort's workspace, the
    # tracking buffers; it is reported on its own), training frames with no densification inside their mapping
    n = len(frames.colors)
    dens_frames = set()
    it_at = 200
    for t in range(1, n):
        if t in set(int(i) for i in frames.i_train):
            if any(it_at < d <= it_at + 30 for d, _ in dens):
                dens_frames.add(t)
            it_at += 30
    steady = [t for t in range(2, n)]
    steady_map = [t for t in steady if t in mapp and t not in dens_frames]
    mean = lambda xs: float(sum(xs) / len(xs)) if xs else None
    trk = mean([track[t] for t in steady])
    out["progressive"] = {
        "frames": n, "gaussians_start": P, "gaussians_end": pc.num_points,
        "tracking_ms_per_frame": trk,
        "tracking_iterations_ms_per_frame": mean([iters[t] for t in steady]),
        "tracking_ms_per_iter": mean([iters[t] for t in steady]) / 50.0,
        "per_frame_setup_ms": {"pose_init_and_optimizer": mean([setup[t] for t in steady]),
                               "rigid_mask": mean([rigid[t] for t in steady]),
                               "flow_targets": mean([targets[t] for t in steady]),
                               "sum": mean([setup[t] + rigid[t] + targets[t] for t in steady])},
        "first_tracked_frame_ms": {"setup": setup.get(1), "tracking": track.get(1), "iterations": iters.get(1)},
        "mapping_ms_per_frame": mean([mapp[t] for t in steady_map]),
        "mapping_ms_per_iter_two_views": mean([mapp[t] for t in steady_map]) / 30.0 if steady_map else None,
        "first_frame_mapping_ms": mapp.get(0), "first_frame_mapping_ms_per_iter": mapp.get(0, 0.0) / 200.0,
        "ms_per_frame": (mean([setup[t] + track[t] for t in steady]) or 0.0) + (mean([mapp[t] for t in steady_map]) or 0.0),
        "densify_and_prune": [{"iteration": d, "ms": ms} for d, ms in dens],
        "wall_ms_profiled": wall_profiled,
        "what": "Runner.progressive_run (train.py:318-345): per steady frame = pose init + pose Adam/MultiStepLR + Sampson "
                "rigid mask + flow targets + 50 tracking iterations + 30 two-view mapping iterations; profile=True brackets "
                "every phase with a device synchronisation",
    }
    out["progressive"]["iters_per_sec"] = 80.0 / (out["progressive"]["ms_per_frame"] * 1e-3) if out["progressive"]["ms_per_frame"] else None
    # the same run without the phase synchronisations: what the profile costs
    pc, poses, frames = problem()
    run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.progressive_run()
    torch.cuda.synchronize()
    out["progressive"]["wall_ms_unprofiled"] = (time.perf_counter() - t0) * 1e3
    del run, pc, poses, frames

    # ---- global_run ----
    pc, poses, frames = problem()
    run = Runner(pc, poses, frames, profile=True)  # (the only phase inside global_run is densify_and_prune)
    run.global_run(9, eval_every=0)  # warm-up: Adam state, buffers (iterations 1 .. 10 of the counter)
    run.iteration = 0
    torch.cuda.synchronize()
    n_it = 300
    t0 = time.perf_counter()
    run.global_run(n_it - 1, eval_every=0)  # range(0, iterations + 1): n_it mapping iterations, densify at the 300th
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    dens = [ms for k, t, ms in run.phase_ms if k == "densify_and_prune"]
    out["global"] = {
        "iterations": n_it, "ms_total": total, "densify_and_prune_ms": dens, "gaussians_start": P,
        "gaussians_end": pc.num_points,
        "ms_per_iter": (total - sum(dens)) / n_it, "ms_per_iter_incl_densify": total / n_it,
        "iters_per_sec": n_it / ((total - sum(dens)) * 1e-3),
        "what": "Runner.global_run (train.py:378-443): fresh Adam (eps 1e-8), per iteration a random training frame, the xyz "
                "learning-rate schedule, one one-view mapping iteration with densification statistics; densify_and_prune at "
                "iteration 300 (timed apart); no test-frame evaluation inside the timed span",
    }
    return out



def comm_model(nbytes, world, step_ms_one_gpu):
    """What DESIGN.md s6 predicts for the step's one exchange on a fully connected xGMI node, stated in the line so that a
    scaling record can be judged against it (VERDICT r3 #2c).  t = launches * a_launch + steps * a_step + wire_bytes / b:
    every GPU has a link to every other (7 links at N = 8), both routes can keep all N - 1 links of a GPU busy, so the
    bandwidth term is the same -- 2 (N-1)/N * S bytes per GPU over (N-1) links -- and the routes differ in their dependent
    steps: a ring all-reduce has 2 (N-1), the direct reduce-scatter + all-gather has 2 (and two launches)."""
    link_GBps = 64.0      # per direction and link, effective (76.8 GB/s peak = half of the 153.6 GB/s a link carries both ways)
    a_launch, a_step = 0.020, 0.005  # ms: one collective launch end to end; one dependent hop inside a collective
    wire_ms = 2.0 * (world - 1) / world * nbytes / ((world - 1) * link_GBps * 1e9) * 1e3
    rccl = a_launch + 2 * (world - 1) * a_step + wire_ms
    direct = 2 * a_launch + 2 * a_step + wire_ms + 0.010  # + the local sum of N shards
    best = min(rccl, direct)
    return {"rccl_ms": rccl, "direct_ms": direct, "wire_ms": wire_ms,
            "params": {"link_GBps_per_direction": link_GBps, "links_used": world - 1, "launch_ms": a_launch,
                       "hop_ms": a_step, "bytes": nbytes},
            "step_ms": {"one_collective": step_ms_one_gpu + best, "per_rank_path_without_exchange": step_ms_one_gpu},
            "weak_scaling_efficiency": step_ms_one_gpu / (step_ms_one_gpu + best),
            "note": "a model (DESIGN.md s6) of the exchange on top of THIS run's own one-rank step (comm.one_rank: the same "
                    "configuration and scene, timed on every rank without an exchange before the N-rank loop), not a measurement"}



def drop_in_breakdown(pc, poses, frames, reps=5):
    """Where a step of the unchanged-checkout route spends its wall time: the phases of trainer.mapping_step(fused=False,
    hip_losses=False), each bracketed by a device synchronisation, and the HOST time inside this library's own operator
    calls (GaussianRasterizer forward x2 / backward x2) -- everything else is the reference's own torch code."""
    from fsgs_amd import optim, rasterizer
    from fsgs_amd.render import render_two_pass
    from fsgs_amd.trainer import mapping_loss

    acc = {"render_fwd_two_pass": 0.0, "torch_losses_fwd": 0.0, "backward": 0.0, "stats_and_torch_adam": 0.0,
           "of_which_host_inside_rasteriser_fwd_calls": 0.0, "of_which_host_inside_rasteriser_bwd_calls": 0.0}
    orig_f, orig_b = rasterizer.raster_forward, rasterizer.raster_backward

    def wrap(fn, key):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            acc[key] += (time.perf_counter() - t) * 1e3
            return r
        return w

    rasterizer.raster_forward = wrap(orig_f, "of_which_host_inside_rasteriser_fwd_calls")
    rasterizer.raster_backward = wrap(orig_b, "of_which_host_inside_rasteriser_bwd_calls")
    try:
        n = len(frames.colors)
        for it in range(reps):
            ts = it % n
            marks = []
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            pkg = render_two_pass(poses, ts, pc, gs_grad=True, cam_grad=False)
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            loss = mapping_loss(pkg, frames.colors[ts], frames.monodeps[ts], hip_losses=False)
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            loss.backward()
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            with torch.no_grad():
                optim.densify_stats(pkg["radii"], pkg["viewspace_points"].grad, pc.variables["max_radii2D"],
                                    pc.variables["xyz_gradient_accum"], pc.variables["denom"])
                pc.optimizer.step()
                pc.optimizer.zero_grad(set_to_none=True)
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            for k, (a, b) in zip(("render_fwd_two_pass", "torch_losses_fwd", "backward", "stats_and_torch_adam"),
                                 zip(marks, marks[1:])):
                acc[k] += (b - a) * 1e3
    finally:
        rasterizer.raster_forward, rasterizer.raster_backward = orig_f, orig_b
    return {k: v / reps for k, v in acc.items()}



def drop_in_extra(cfg_name, device, steps=30, warmup=5):
    """The route an UNCHANGED reference checkout takes (one PYTHONPATH entry, nothing else) and the route
    with INTEGRATION.md s3's three one-line edits, both under torch.autograd at this configuration:
      unchanged   : render() = the reference's own sequence of ~40 small torch ops around TWO GaussianRasterizer calls
                    (fsgs_amd.render.render_two_pass restates it, gaussian_renderer/__init__.py:49-92), the losses as plain
                    torch (utils/loss_utils.py:41-127), torch.optim.Adam on six groups -- only the rasteriser is this library;
      losses_edit_only : as above with INTEGRATION.md s3's second edit alone (the three loss imports): what the reference's
                    own torch losses -- 40 patch crops with a host synchronisation each, ~1500 small launches forward and
                    backward -- cost in the unchanged route;
      three_edits : the fused render op + the HIP loss kernels + FusedAdam, still driven by loss.backward()."""
    from fsgs_amd.trainer import mapping_step

    out = {}
    for name, fused, hip_losses, fused_adam in (("unchanged", False, False, False), ("losses_edit_only", False, True, False),
                                                ("three_edits", True, True, True)):
        pc, poses, frames, cam, sc = build_problem(cfg_name, device, 0, 1)
        pc.training_setup(eps=1e-8, fused=fused_adam)
        n = len(frames.colors)
        for it in range(warmup):
            mapping_step(pc, poses, frames, [it % n], fused=fused, hip_losses=hip_losses)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(steps):
            mapping_step(pc, poses, frames, [it % n], fused=fused, hip_losses=hip_losses)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"ms_per_step": dt / steps * 1e3, "iters_per_sec": steps / dt, "host_issue_ms_per_step": t_issue / steps * 1e3,
                     "steps": steps}
        if name == "unchanged":
            out[name]["breakdown_ms"] = drop_in_breakdown(pc, poses, frames)
        del pc, poses, frames
        torch.cuda.empty_cache()
    out["autobind"] = drop_in_autobind(cfg_name, device, steps, warmup)
    out["autobind"]["over_three_edits"] = out["autobind"]["ms_per_step_without_driver_statistics"] / out["three_edits"]["ms_per_step"]
    out["autobind"]["over_three_edits_incl_driver_statistics"] = out["autobind"]["ms_per_step"] / out["three_edits"]["ms_per_step"]
    out["what"] = ("trainer.mapping_step under torch.autograd at %s: `unchanged` = two drop-in GaussianRasterizer calls + torch glue "
                   "+ torch losses + torch.optim.Adam (INTEGRATION s2); `losses_edit_only` = the same with the HIP loss kernels; "
                   "`three_edits` = fused render + HIP losses + FusedAdam (INTEGRATION s3)" % cfg_name)
    return out



def drop_in_autobind(cfg_name, device, steps=30, warmup=5):
    """INTEGRATION.md s2's "no edit, one environment variable" row: a stand-in checkout (scripts/standin_checkout.py: the
    reference's module / function names; its own bodies are the `unchanged` route) imported with fsgs_amd.autobind
    installed, so its driver's `render`, the three losses and `torch.optim.Adam` are the fused op, the HIP loss kernels and
    FusedAdam BY NAME -- the same kernels as `three_edits`, reached with no source edit."""
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import standin_checkout
    from fsgs_amd import autobind

    tree = standin_checkout.write_tree(tempfile.mkdtemp(prefix="fsgs_standin_"))
    standin_checkout.forget()
    autobind.install()
    sys.path.insert(0, tree)
    try:
        import standin_train
        from scene import GaussianModel

        pc0, poses, frames, cam, sc = build_problem(cfg_name, device, 0, 1)
        pc = GaussianModel(dict(sc), sh_degree=3, device=device, scene_radius=float(sc["depth_map"].max()) / 2.0)
        pc.cam, pc.active_sh_degree, pc.spatial_lr_scale = pc0.cam, pc0.active_sh_degree, pc0.spatial_lr_scale
        del pc0
        pc.training_setup(eps=1e-8)
        n = len(frames.colors)

        def timed(statistics):
            for it in range(warmup):
                standin_train.mapping_iteration(poses, pc, frames.colors, frames.monodeps, it % n, statistics=statistics)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(steps):
                standin_train.mapping_iteration(poses, pc, frames.colors, frames.monodeps, it % n, statistics=statistics)
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            return time.perf_counter() - t0, t_issue

        dt_core, _ = timed(False)
        dt, t_issue = timed(True)
        return {"ms_per_step": dt / steps * 1e3, "iters_per_sec": steps / dt, "host_issue_ms_per_step": t_issue / steps * 1e3,
                # render + losses + backward + Adam alone: what `three_edits` times, which takes the statistics through ONE
                # fused call (optim.densify_stats) instead of the driver's own boolean-mask statements (train.py:297-303)
                "ms_per_step_without_driver_statistics": dt_core / steps * 1e3,
                "steps": steps, "bound": autobind.bound(), "optimizer": type(pc.optimizer).__name__,
                "render": standin_train.render.__module__, "losses": standin_train.rgb_loss_func.__module__,
                "what": "stand-in checkout (reference names, unchanged-route bodies) + FSGS_AUTOBIND: fused render + HIP losses "
                        "+ FusedAdam bound by name; densification statistics as the reference's torch statements"}
    finally:
        sys.path.remove(tree)
        standin_checkout.forget()
        autobind.uninstall()
        torch.cuda.empty_cache()



def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this script under torch.distributed.run with one rank
    per GPU (RCCL), pass rank 0's JSON line through, return the launcher's exit code.  Non-zero when fewer than N GPUs
    are visible (FSGS_DIST_ONE_GPU=1: every rank on device 0 over gloo -- the smoke test of this code path on a
    one-GPU box, never a measurement)."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if os.environ.get("FSGS_DIST_ONE_GPU") != "1" and have < n:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (n, have))
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rccl_diagnostics_env():
    """N > 1: have RCCL write what it decided at communicator INIT (topology graph, rings / trees, channels, the
    transport of every peer connection: P2P over xGMI or not) to a per-rank file that rank 0 folds into the bench line's
    `comm.rccl_info`, so that a poor scaling point can be attributed.  Init-time subsystems only: nothing
    is logged per collective, the timed loop is not touched.  FSGS_RCCL_TUNING_LOG=1 adds the TUNING subsystem (the
    algorithm / protocol chosen per call -- one log line per collective, a diagnostic run, not a measurement).
    NCCL_DEBUG is raised to INFO unless it already asks for more than VERSION / WARN; the other variables only get
    defaults."""
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN"):  # (the image presets VERSION)
        os.environ["NCCL_DEBUG"] = "INFO"
    subsys = "INIT,GRAPH,ENV" + (",TUNING" if os.environ.get("FSGS_RCCL_TUNING_LOG") == "1" else "")
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", subsys)
    os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/fsgs_rccl.%h.%p.log")


def rccl_info_lines(limit=40):
    """the lines of this rank's RCCL INFO log that name the topology, channels, transports, algorithm / protocol"""
    import glob
    import re

    pat = os.environ.get("NCCL_DEBUG_FILE", "")
    if not pat:
        return None
    files = glob.glob(pat.replace("%h", "*").replace("%p", str(os.getpid())))
    keep = re.compile(r"(RCCL version|NCCL version|nRanks|Channel \d+/\d+\s*:|Ring \d+|Trees|channels|via P2P|via SHM|via NET|"
                      r"xGMI|XGMI|Algo|[Pp]rotocol|proto |threshold|NCCL_[A-Z_]+ set|RCCL_[A-Z_]+ set|Connected all|Dmabuf|Using network|"
                      r"NCCL WARN)")
    out = []
    for f in files:
        try:
            for line in open(f, errors="replace"):
                if keep.search(line):
                    # drop the "[time] host:pid:tid [dev]" prefix and the build path RCCL prints in front of its warnings
                    l = re.sub(r"^(\[[^\]]*\] )?\S+:\d+:\d+ \[\d+\] ", "", line.strip())
                    out.append(re.sub(r"/\S*/src/", "src/", l)[:200])
        except OSError:
            pass
    seen, uniq = set(), []
    for l in out:  # channel lines repeat per peer: one of each shape is enough
        key = re.sub(r"\d+", "#", l)
        if key not in seen:
            seen.add(key)
            uniq.append(l)
    return uniq[:limit]


EXTRAS_DEADLINE_S = int(os.environ.get("FSGS_BENCH_EXTRAS_DEADLINE", "300"))  # N > 1: what the reporting / extras behind the timed loop may take before the line is printed without them


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--two-pass", action="store_true", help="unfused render (two rasteriser calls)")
    ap.add_argument("--torch-losses", action="store_true", help="plain-PyTorch losses instead of the HIP kernels")
    ap.add_argument("--autograd", action="store_true",
                    help="drive the step through torch.autograd (trainer.mapping_step) instead of the autograd-free "
                         "stepper (fast_step.FastStepper); same arithmetic, more host overhead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tracking", action="store_true", help="skip the extra tracking-iteration timing")
    ap.add_argument("--no-stats", action="store_true",
                    help="mapping iteration WITHOUT the densification statistics (the second half of global_run: "
                         "iteration >= 15000); the default keeps them, as the first 15000 iterations do")
    ap.add_argument("--ar-chunks", type=int, default=1,
                    help="N > 1: > 1 produces the compact gradient in that many row chunks and starts each chunk's "
                         "all-reduce while the next is produced, Adam per chunk behind it (dist.ProducerPipelinedReducer); "
                         "default one collective in the first timed loop; every N > 1 run then times the x4 pipelined route over "
                         "the same K steps as well and reports it beside the configured route (exchange_routes_ms_per_step); `value` is always the configured route")
    ap.add_argument("--smoke", action="store_true",
                    help="allow FSGS_DIST_ONE_GPU=1 (every rank on device 0 over gloo): a smoke test of the N > 1 code path on a "
                         "one-GPU box.  Without this flag a run with that variable set is refused -- its line must never be "
                         "mistaken for a scaling point")
    ap.add_argument("--ar-algo", default="all_reduce", choices=["all_reduce", "rccl", "direct"],
                    help="N > 1: the step's exchange -- `all_reduce` (`rccl` = the same, the name of rounds 2-4): one torch.distributed all_reduce (the backend picks algorithm and "
                         "protocol); `direct`: dist.DirectAllReduce, an explicit all-to-all of shards + local sum + all-gather "
                         "over the point-to-point xGMI links (SURVEY s5).  The extras of every N > 1 run time both alone")
    ap.add_argument("--dp-path", action="store_true", help="N = 1 only: run the per-rank code path of N > 1 (compact gradient + Adam from it) with a no-op all-reduce")
    ap.add_argument("--scene", default="default", choices=sorted(SCENES),
                    help="dense: every Gaussian 1.62x larger -> upstream pair count ~ SURVEY s8d's nominal 10 tiles per "
                         "Gaussian (3 M at C2); the default run reports it as the extra `dense_scene` anyway")
    ap.add_argument("--densify-every", type=int, default=0,
                    help="> 0: densify_and_prune (device-side, csrc/densify.hip) every N steps INSIDE the timed loop, as "
                         "train.py:305-311 does every 300 iterations -- configuration 4 as BASELINE.json states it "
                         "(--config C4 --densify-every 300)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (raster-only timing, dense scene)")
    ap.add_argument("--no-harness", action="store_true", help="skip the harness extra (Runner.progressive_run / global_run)")
    ap.add_argument("--profile-all", action="store_true", help="HIP-event timing of every kernel (adds overhead)")
    ap.add_argument("--profile-stride", type=int, default=3,
                    help="HIP-event time every n-th launch of the dominant kernels inside the timed region (each timed "
                         "launch costs two event packets of dispatch gap); 1 = every launch.  Keep it coprime with the 8-frame cycle")
    args = ap.parse_args()
    if args.ar_algo == "rccl":
        args.ar_algo = "all_reduce"  # the ALGORITHM is what the route fields name; the transport is comm.backend
    if os.environ.get("FSGS_DIST_ONE_GPU") == "1" and args.gpus > 1 and not args.smoke:
        sys.stderr.write("bench.py: FSGS_DIST_ONE_GPU=1 puts every rank on one GPU over gloo -- a smoke test, not a scaling "
                         "point; pass --smoke to run it\n")
        sys.exit(2)

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        sys.exit(self_launch(args.gpus))  # no launcher around us: start the N ranks ourselves

    from fsgs_amd import _lib, dist as fdist
    from fsgs_amd.trainer import mapping_step

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        rccl_diagnostics_env()
    rank, world, local = fdist.init_from_env()
    if args.gpus != world:
        # a launcher started a different number of ranks than --gpus names: refuse, a mislabelled line is worse than none
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback on the product path)"
    if world > 1 and os.environ.get("FSGS_DIST_ONE_GPU") != "1" and torch.cuda.device_count() < world:
        if rank == 0:
            sys.stderr.write("bench.py: %d ranks but only %d GPU(s) visible\n" % (world, torch.cuda.device_count()))
        sys.exit(3)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    np.random.seed(0)
    _lib.load()

    # with densification inside the timed loop the targets carry detail the cloud lacks (build_problem: texture), so that
    # part of the cloud crosses the reference's 2e-4 threshold and clone / split outgrow the opacity prune
    texture = 0.15 if args.densify_every else 0.0
    pc, poses, frames, cam, sc = build_problem(args.config, device, rank, world, scene=args.scene, texture=texture)
    W, H, P, _ = CONFIGS[args.config]
    fused = not args.two_pass
    hip_losses = not args.torch_losses
    # which implementation pieces exist in this build (recorded in the JSON, never silently swapped)
    try:
        from fsgs_amd import render_ops  # noqa: F401
    except ImportError:
        fused = False
    try:
        from fsgs_amd import loss_ops  # noqa: F401
    except ImportError:
        hip_losses = False

    n_frames = len(frames.colors)
    use_fast = fused and hip_losses and not args.autograd
    bucket = fdist.GradBucket(pc) if (world > 1 and not use_fast) else None
    if use_fast:
        from fsgs_amd.fast_step import FastStepper

        stepper = FastStepper(pc, poses, frames)

    # N > 1: ONE all-reduce of the compact gradient (optionally chunked and pipelined with Adam, --ar-chunks)
    reducer = (fdist.ProducerPipelinedReducer(args.ar_chunks) if args.ar_chunks > 1 else
               fdist.DirectAllReduce() if args.ar_algo == "direct" else fdist.all_reduce_compact)

    densify_log = []

    def maybe_densify(it):
        """train.py:305-311 inside the timed region: statistics were accumulated by every step since the last call."""
        if not args.densify_every or (it + 1) % args.densify_every:
            return
        if world > 1:
            fdist.sync_densification_stats(pc)
        t0d = time.perf_counter()
        P0 = pc.num_points
        pc.densify_and_prune_device(pc.opt.densify_grad_threshold, 0.05, None)
        torch.cuda.synchronize()
        densify_log.append({"step": it + 1, "P_before": P0, "P_after": pc.num_points, "ms": (time.perf_counter() - t0d) * 1e3})
        if use_fast:
            stepper.pc = pc

    def one_step(it):
        ts = (rank + it * world) % n_frames  # 1 camera per rank, a different one each step
        if use_fast:
            # N > 1: ONE all-reduce of the compact [P,14] gradient (56 B / Gaussian), then Adam from it
            red = reducer if world > 1 else ((lambda t_: None) if args.dp_path else None)
            return stepper.mapping_step([ts], reduce_compact=red, collect_stats=not args.no_stats), None
        if bucket is not None:
            bucket.attach(pc)
        sync = (lambda pc_: fdist.sync_gradients(pc_, bucket)) if world > 1 else None
        return mapping_step(pc, poses, frames, [ts], fused=fused, hip_losses=hip_losses, grad_sync=sync)

    def barrier():
        if world > 1:
            if torch.distributed.get_backend() == "nccl":
                torch.distributed.barrier(device_ids=[local])  # name the device: no rank -> GPU guessing inside RCCL
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    # N > 1: the one-rank step of THIS box, configuration and scene, timed on every rank at once (no exchange: the N = 1
    # code path with Adam fused into the backward, on a cloud of its own so that the replicas stay identical) before the
    # N-rank loop -- efficiency_measured = value / (N x this), and the exchange model is laid on top of it
    one_rank = None
    if world > 1 and use_fast:
        pc1, poses1, frames1, _c1, _s1 = build_problem(args.config, device, rank, world, scene=args.scene, texture=texture)
        st1 = FastStepper(pc1, poses1, frames1)
        k1 = max(10, min(args.steps, 100))
        for it in range(min(args.warmup, 10)):
            st1.mapping_step([(rank + it) % n_frames], collect_stats=not args.no_stats)
        barrier()
        t1 = time.perf_counter()
        for it in range(k1):
            st1.mapping_step([(rank + it) % n_frames], collect_stats=not args.no_stats)
        barrier()
        d1 = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(d1, op=torch.distributed.ReduceOp.MAX)
        one_rank = {"ms_per_step": float(d1.item()) / k1 * 1e3, "iters_per_sec": k1 / float(d1.item()), "steps": k1,
                    "what": "the N = 1 step (no exchange, Adam fused into the backward) on every rank at once, max over ranks"}
        del st1, pc1, poses1, frames1
        torch.cuda.empty_cache()

    for it in range(args.warmup):
        one_step(it)
    barrier()
    dominant = "blend_bwd"
    if use_fast:
        stepper.pairs_total = stepper.forward_calls = 0
    _lib.profile_enable(None if args.profile_all else [dominant, "blend_fwd"], stride=max(1, args.profile_stride))

    def timed_block(first_it):
        """EXACTLY K steps between two barrier + synchronize brackets; max over ranks"""
        t0 = time.perf_counter()
        for it in range(args.steps):
            res = one_step(args.warmup + first_it + it)
            maybe_densify(first_it + it)
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d = float(tt.item())
        return d, res

    # A K-step block of a driver run (K = 20) is 13 ms of GPU work: too thin a sample.  The block is
    # repeated -- every repetition EXACTLY K steps inside its own brackets -- until at least MIN_TIMED_S have been timed;
    # `ms_per_step` / `value` are the mean over the blocks, `timed_blocks` carries every block and their spread.  All ranks
    # derive the block count from the same all-reduced first block.
    MIN_TIMED_S = 0.1
    d0, (loss, pkg) = timed_block(0)
    blocks = [d0]
    if not args.densify_every:  # (a densification schedule is defined over ONE pass of K steps)
        n_more = min(63, max(0, int(np.ceil(MIN_TIMED_S / max(d0, 1e-6))) - 1))
        for b_ in range(n_more):
            d_, (loss, pkg) = timed_block((b_ + 1) * args.steps)
            blocks.append(d_)
    dt = float(np.mean(blocks))
    prof = _lib.profile_read()
    _lib.profile_enable([])
    timed_blocks = {"blocks": len(blocks), "steps_per_block": args.steps, "timed_seconds": float(np.sum(blocks)),
                    "ms_per_step_by_block": [b_ / args.steps * 1e3 for b_ in blocks],
                    "ms_per_step_min": float(np.min(blocks)) / args.steps * 1e3,
                    "ms_per_step_max": float(np.max(blocks)) / args.steps * 1e3,
                    "ms_per_step_std": float(np.std(blocks)) / args.steps * 1e3}

    # The number the run exists for is in hand.  Everything below is reporting and untimed extras, some of them with
    # collectives of their own (N > 1): if any of that stalls -- a transport that mishandles the chunked exchange, a
    # rank that died -- rank 0 still prints the line with what it has and every rank leaves, instead of the scaling
    # record losing the measurement to a hang.
    watchdog = None
    if world > 1:
        import threading

        def bail():
            if rank == 0:
                print(json.dumps({
                    "metric": "train_iters_per_sec", "value": args.steps * world / dt, "unit": "iters/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "%s: mapping iteration, %dx%d, %d Gaussians (%s scene), 1 camera/rank" % (
                        args.config, W, H, P, CONFIGS[args.config][3]), "parallelism": "dp%d" % world},
                    "roofline": None, "cpu_baseline": None,
                    # machine-readable: the timed loop (barrier + synchronize on both sides) completed on every rank, what
                    # was abandoned is the reporting phase behind it -- a scaling record can flag the run on this field
                    "extras_incomplete": True,
                    "note": "the untimed extras behind the timed loop did not finish within %d s and were abandoned" % EXTRAS_DEADLINE_S,
                }), flush=True)
            os._exit(0)

        watchdog = threading.Timer(EXTRAS_DEADLINE_S, bail)
        watchdog.daemon = True
        watchdog.start()

    # ---- roofline of the dominant kernel (per-unit bytes stated in DESIGN.md) ----
    from fsgs_amd import rasterizer

    R = int(getattr(rasterizer, "last_num_rendered", 0) or 0)
    if use_fast and stepper.forward_calls:  # the cameras differ: mean pair count over the timed steps, like the mean time
        R = int(round(stepper.pairs_total / stepper.forward_calls))
    C = 6 if fused else 3
    ms_total, launches = prof.get(dominant, (0.0, 0))
    roofline = None
    if launches:
        # blend_bwd algorithmic bytes per launch: R*(4 idx + 24 xy/conic/opacity + 4C colour) +
        # HW*(4C dL/dpixel + 8 final_T/n_contrib) + P*4*(6 + C) accumulated gradients
        cg = 4 if (use_fast and C == 6) else C  # planes of dL/dpixel actually read (rgb + depth)
        b_alg = R * (4 + 24 + 4 * C) + H * W * (4 * cg + 8) + P * 4 * (6 + C)
        avg_s = ms_total / launches / 1e3
        ach = b_alg / avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_kernel_ms": ms_total / launches,
                    "launches": launches, "algorithmic_bytes": b_alg, "num_rendered": R, "channels": C}
        # HBM bytes per launch measured offline with rocprofv3 --pmc (scripts/gpu_pmc.sh: separate passes, gfx950
        # FETCH_SIZE x2 correction calibrated on the Adam kernel) and VALU instruction counts (scripts/gpu_sq.sh) for
        # exactly this workload; a missing file or kernel key is an ERROR in the JSON line, never a silent null
        # template arguments <C; SPLIT; POSE_ONLY; CGRAD[; ROW]> as rocprofv3 prints them; matched as a PREFIX so that a
        # template parameter appended later does not orphan the entry (and if nothing matches, that is an error below)
        key = "blend_bwd_kernel<%d; %s; false%s" % (C, "true" if fused else "false", "; 4" if (use_fast and C == 6) else "")
        try:
            import glob

            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
            if not files:
                raise FileNotFoundError("no profiles/r*_pmc_traffic.json")
            pmc = json.load(open(files[-1]))
            src = os.path.relpath(files[-1], ROOT)
            if world != 1 or pmc.get("config") != args.config or args.scene != "default" or args.densify_every:
                roofline["traffic_note"] = "counters in %s were collected on config %s, default scene, 1 GPU" % (src, pmc.get("config"))
            else:
                hits = [k for k in pmc["kernels"] if k.startswith(key)]
                if len(hits) != 1:
                    raise KeyError("%s has %d entries starting with %r (kernels: %s)" % (
                        src, len(hits), key, [k for k in pmc["kernels"] if "blend" in k]))
                ent = pmc["kernels"][hits[0]]
                roofline["traffic_kernel"] = hits[0]
                roofline["traffic"] = ent["traffic_bytes"]
                roofline["traffic_over_algorithmic"] = ent["traffic_bytes"] / b_alg
                roofline["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" % src
                # the counters are an offline constant of that file, collected at ITS pair count; this run's R is `num_rendered`
                roofline["traffic_collected_at_num_rendered"] = pmc.get("num_rendered")
                issue, issue_src = load_issue_rates()
                vb = valu_block(ent, hits[0], "blend_bwd", avg_s, issue, issue_src)
                if vb:
                    roofline["valu_frac"] = vb["valu_frac"]
                    if "valu_frac_of_achievable" in vb:
                        roofline["valu_frac_of_achievable"] = vb["valu_frac_of_achievable"]
                    vb["source"] = "SQ_INSTS_VALU and the SQ_* stall counters in %s" % src
                    roofline["valu"] = vb
                # the second issue-bound kernel, same definitions: blend_fwd (HIP events of this run; counters offline)
                fms, fl = prof.get("blend_fwd", (0.0, 0))
                Cf = 4 if (use_fast and C == 6 and getattr(stepper, "mapping_planes4", False)) else C  # planes the step's forward blends
                # (round 5: the forward blends with four waves per tile at every size -- blend_fwd_quad_kernel)
                fhits = [k for k in pmc["kernels"] if k.startswith(("blend_fwd_quad_kernel<%d" % Cf, "blend_fwd_kernel<%d" % Cf))]
                if fl and len(fhits) == 1:
                    fent = pmc["kernels"][fhits[0]]
                    f_alg = R * (4 + 24 + 4 * Cf) + H * W * (4 * Cf + 4 + 8)
                    f_s = fms / fl / 1e3
                    fvb = valu_block(fent, fhits[0], "blend_fwd", f_s, issue, issue_src) or {}
                    roofline["blend_fwd"] = {
                        "kernel": fhits[0], "avg_kernel_ms": fms / fl, "algorithmic_bytes": f_alg,
                        "achieved": f_alg / f_s / 1e9, "frac": f_alg / f_s / 1e9 / HBM_PEAK_GBS,
                        "traffic": fent.get("traffic_bytes"), "valu_frac": fvb.get("valu_frac"),
                        "valu_frac_of_achievable": fvb.get("valu_frac_of_achievable"), "valu": fvb or None}
        except Exception as e:  # noqa: BLE001
            roofline["traffic_error"] = "%s: %s" % (type(e).__name__, e)
            sys.stderr.write("bench.py: roofline.traffic unavailable -- %s\n" % roofline["traffic_error"])
    kernels = {k: {"avg_ms": v[0] / v[1], "launches": v[1]} for k, v in prof.items()}

    upstream_main = upstream_pairs(stepper, W, H) if use_fast else None

    # ---- extra: the rasteriser alone (BASELINE.json: "fwd+bwd ms @1280x1024, 300k Gaussians"), rank 0 / N = 1 ----
    # Untimed w.r.t. `value`: a second, short loop with HIP events around EVERY kernel of the fused rasteriser
    # (preprocess + SH/activations/transform, binning, 6-channel blend, blend backward, per-Gaussian backward with the
    # glue adjoints) and step_optimizer=False, so that no Adam work hides inside the backward kernel.
    raster = None
    RASTER_GROUPS = ("render_pre_fwd", "sort_depth", "sort_tile", "blend_fwd", "blend_bwd", "render_pre_bwd")
    if use_fast and world == 1 and not args.no_extras:
        _lib.profile_enable(list(RASTER_GROUPS), stride=1)
        stepper.pairs_total = stepper.forward_calls = 0
        n_r = 24
        stepper.reuse_colors = False  # the rasteriser ALONE does all of its work here: every forward evaluates the SH
        # colours (and, as in the timed loop, all six planes of both passes of render() are blended)
        for it in range(n_r):
            stepper.mapping_step([it % n_frames], step_optimizer=False)
        torch.cuda.synchronize()
        stepper.reuse_colors = True
        pr = _lib.profile_read()
        _lib.profile_enable([])
        Rr = int(round(stepper.pairs_total / max(stepper.forward_calls, 1)))
        parts = {k: pr[k][0] / pr[k][1] for k in RASTER_GROUPS if k in pr and pr[k][1]}
        if len(parts) == len(RASTER_GROUPS):
            ms = sum(parts.values())
            Pn = pc.num_points
            b_r = 232 * Pn + 108 * Rr + 68 * H * W  # SURVEY.md s8d, fused 6-channel single pass, with the measured R
            raster = {"raster_fwd_bwd_ms": ms, "kernels_ms": parts, "num_rendered": Rr,
                      "what": "one fused 6-channel pass = both passes of render(): preprocess(+glue) + binning + blend fwd "
                              "+ blend bwd + per-Gaussian bwd(+glue adjoints); Adam not included",
                      "roofline_raster": {"bound": "hbm", "algorithmic_bytes": b_r, "achieved": b_r / (ms * 1e-3) / 1e9,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_r / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "formula": "232 P + 108 R + 68 HW bytes (SURVEY.md s8d)"}}

    # ---- extra: the pose-only tracking iteration (train.py:166-200) on the same scene, rank 0 / N = 1 ----
    tracking = None
    if use_fast and world == 1 and not args.no_tracking:
        from fsgs_amd.flow import FlowTargets

        poses.initialize_tracking_optimizer(50)
        rigid = torch.ones((H, W), dtype=torch.bool, device=device)
        depth_prev = frames.monodeps[0].reshape(1, H, W)
        flow_fw = torch.zeros((2, H, W), device=device)
        targets = FlowTargets(depth_prev, np.eye(4, dtype=np.float32), cam["K"], flow_fw, rigid)
        for _ in range(10):
            stepper.tracking_step(1, targets, None, want_losses=False)  # all-rigid frame (Runner.tracking does the same)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nt = 40
        for _ in range(nt):
            stepper.tracking_step(1, targets, None, want_losses=False)
        torch.cuda.synchronize()
        tracking = {"iters_per_sec": nt / (time.perf_counter() - t1), "ms_per_iter": (time.perf_counter() - t1) / nt * 1e3,
                    "what": "render(gs_grad=False, cam_grad=True) + masked rgb loss + flow loss + pose Adam"}

    # ---- extra: the TWO-VIEW mapping iteration of progressive_run (train.py:214-259: a random keyframe + the current
    # frame, summed loss, one Adam step) on the same scene, N = 1 ----
    two_view = None
    if use_fast and world == 1 and not args.no_extras and not args.densify_every:
        for it in range(6):
            stepper.mapping_step([it % n_frames, (it + 3) % n_frames], collect_stats=not args.no_stats)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n2 = 40
        for it in range(n2):
            stepper.mapping_step([it % n_frames, (it + 3) % n_frames], collect_stats=not args.no_stats)
        torch.cuda.synchronize()
        two_view = {"ms_per_iter": (time.perf_counter() - t2) / n2 * 1e3, "iters_per_sec": n2 / (time.perf_counter() - t2),
                    "what": "2 views x (fused render fwd + losses + render bwd into the compact [P,14] gradient; the second view "
                            "reuses the first view's per-Gaussian colours) + Adam from the summed gradient"}

    # ---- extra: the reference's own loops through the harness (train.py:318-443), N = 1 ----
    harness = None
    if use_fast and world == 1 and not args.no_extras and not args.no_harness and not args.densify_every:
        try:
            harness = harness_extra(args.config, device, scene=args.scene)
            harness["bare_step_ms"] = dt / args.steps * 1e3
            harness["global"]["over_bare_step"] = harness["global"]["ms_per_iter"] / harness["bare_step_ms"]
        except Exception as e:  # noqa: BLE001  (an extra must never cost the measurement)
            harness = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("bench.py: harness extra failed -- %s\n" % harness["error"])

    # ---- extra: what an unchanged reference checkout gets (drop-in rasteriser only) and the three-edit route, N = 1 ----
    drop_in = None
    if world == 1 and not args.no_extras and not args.no_harness and not args.densify_every and args.scene == "default":
        try:
            drop_in = drop_in_extra(args.config, device)
            drop_in["over_fused_step"] = drop_in["unchanged"]["ms_per_step"] / (dt / args.steps * 1e3)
        except Exception as e:  # noqa: BLE001
            drop_in = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("bench.py: drop-in extra failed -- %s\n" % drop_in["error"])

    # ---- extra: the same mapping step on the DENSE scene (upstream pair count ~ SURVEY s8d's nominal) ----
    dense = None
    if use_fast and world == 1 and not args.no_extras and args.scene == "default" and args.config in ("C2", "C4") \
            and not args.densify_every:
        del stepper
        torch.cuda.empty_cache()
        pc2, poses2, frames2, cam2, sc2 = build_problem(args.config, device, rank, world, scene="dense")
        st2 = FastStepper(pc2, poses2, frames2)
        for it in range(8):
            st2.mapping_step([it % n_frames])
        torch.cuda.synchronize()
        _lib.profile_enable(["blend_bwd", "blend_fwd"], stride=3)
        st2.pairs_total = st2.forward_calls = 0
        td = time.perf_counter()
        nd = 60
        for it in range(nd):
            st2.mapping_step([it % n_frames])
        torch.cuda.synchronize()
        dd = time.perf_counter() - td
        pd_ = _lib.profile_read()
        _lib.profile_enable([])
        Rd = int(round(st2.pairs_total / max(st2.forward_calls, 1)))
        dense = {"scene": "dense (every Gaussian x%.2f)" % SCENES["dense"], "num_rendered": Rd,
                 "upstream_num_rendered": upstream_pairs(st2, W, H), "ms_per_step": dd / nd * 1e3, "iters_per_sec": nd / dd,
                 "kernels_ms": {k: v[0] / v[1] for k, v in pd_.items() if v[1]}}
        # round 6 (VERDICT r5 #4 e): the dense scene's OWN roofline block -- algorithmic bytes at ITS pair count, HIP events of THIS
        # loop, counters from ITS PMC / SQ passes (profiles/r*_pmc_traffic_dense.json: PMC_ARGS="--scene dense" scripts/gpu_pmc.sh)
        try:
            import glob

            dfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_dense.json")))
            dpmc = json.load(open(dfiles[-1])) if dfiles else None
            dsrc = os.path.relpath(dfiles[-1], ROOT) if dfiles else None
            issue, issue_src = load_issue_rates()
            Pd = st2.pc.num_points if hasattr(st2, "pc") else P
            droof = {}
            for fam, alg in (("blend_bwd", Rd * (4 + 24 + 4 * 6) + H * W * (4 * 4 + 8) + Pd * 4 * (6 + 6)),
                             ("blend_fwd", Rd * (4 + 24 + 4 * 6) + H * W * (4 * 6 + 4 + 8))):
                if fam not in pd_ or not pd_[fam][1]:
                    continue
                t_s = pd_[fam][0] / pd_[fam][1] / 1e3
                blk = {"bound": "hbm", "avg_kernel_ms": t_s * 1e3, "algorithmic_bytes": alg, "achieved": alg / t_s / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t_s / 1e9 / HBM_PEAK_GBS, "traffic": None}
                if dpmc:
                    hits = [k for k in dpmc["kernels"] if k.startswith(fam)]
                    if len(hits) == 1:
                        ent = dpmc["kernels"][hits[0]]
                        blk.update({"kernel": hits[0], "traffic": ent["traffic_bytes"], "traffic_over_algorithmic": ent["traffic_bytes"] / alg,
                                    "traffic_source": dsrc, "traffic_collected_at_num_rendered": dpmc.get("num_rendered")})
                        vb = valu_block(ent, hits[0], fam, t_s, issue, issue_src)
                        if vb:
                            blk.update({"valu_frac": vb["valu_frac"], "valu_frac_of_achievable": vb.get("valu_frac_of_achievable"), "valu": vb})
                    else:
                        blk["traffic_error"] = "%s has %d entries starting with %r" % (dsrc, len(hits), fam)
                else:
                    blk["traffic_error"] = "no profiles/r*_pmc_traffic_dense.json"
                droof[fam] = blk
            dense["roofline"] = droof.get("blend_bwd")
            if "blend_fwd" in droof:
                dense["roofline_blend_fwd"] = droof["blend_fwd"]
        except Exception as e:  # noqa: BLE001
            dense["roofline_error"] = "%s: %s" % (type(e).__name__, e)
        stepper = st2

    # ---- extra (N > 1): the step's one collective on its own, so the scaling numbers can be read ----
    comm = None
    if world > 1:
        nfloat = P * (14 if use_fast else 59)
        buf = torch.zeros((nfloat,), dtype=torch.float32, device=device)
        for _ in range(3):
            torch.distributed.all_reduce(buf)
        barrier()
        tc = time.perf_counter()
        for _ in range(10):
            torch.distributed.all_reduce(buf)
        torch.cuda.synchronize()
        comm_ms = (time.perf_counter() - tc) / 10 * 1e3
        comm = {"what": "all-reduce(SUM) of the %s gradient, alone" % ("compact [P,14]" if use_fast else "[59 P]"),
                "bytes": nfloat * 4, "ms": comm_ms,
                "algbw_GBps": nfloat * 4 / (comm_ms * 1e-3) / 1e9,
                "backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                "devices": torch.cuda.device_count(), "one_gpu_smoke": os.environ.get("FSGS_DIST_ONE_GPU") == "1"}

        def timed(fn, reps=10):
            for _ in range(2):
                fn()
            barrier()
            t_ = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) / reps * 1e3

        def probe(name, fn, reps=10):
            """a diagnostic exchange on its own: an exception here (a transport that cannot run it) is recorded, it must
            never cost the line.  Every rank takes the same branch: the error is agreed on by an all-reduce."""
            err = None
            try:
                val = timed(fn, reps)
            except Exception as e:  # noqa: BLE001
                err, val = "%s: %s" % (type(e).__name__, str(e)[:200]), None
            flag = torch.tensor([0.0 if err is None else 1.0], device=device)
            try:
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            except Exception:  # noqa: BLE001
                pass
            if err is not None or float(flag.item()) > 0:
                comm[name] = None
                comm[name + "_error"] = err or "failed on another rank"
            else:
                comm[name] = val

        # the same bytes as 4 row chunks back to back (what --ar-chunks 4 issues; chunk bounds = multiples of 256 rows) and
        # one tiny message: latency vs bandwidth of this fabric, so that t(one collective) = a + bytes / b can be read off
        rows = nfloat // (14 if use_fast else 59)
        per = max(256, -(-(-(-rows // 4)) // 256) * 256)
        width = nfloat // rows
        chunks = [buf[lo * width:min(rows, lo + per) * width] for lo in range(0, rows, per)]
        probe("chunks4_ms", lambda: [torch.distributed.all_reduce(c) for c in chunks])
        direct = fdist.DirectAllReduce()
        probe("direct_ms", lambda: direct(buf))  # all-to-all of shards + local sum + all-gather (--ar-algo direct)
        comm["timed_route"] = "pipelined x%d" % args.ar_chunks if args.ar_chunks > 1 else args.ar_algo  # (the algorithm; the transport is comm.backend)
        tiny = torch.zeros((1024,), dtype=torch.float32, device=device)
        probe("latency_4KB_ms", lambda: torch.distributed.all_reduce(tiny), reps=50)
        comm["one_rank"] = one_rank
        step1 = one_rank["ms_per_step"] if one_rank else None
        comm["expected"] = comm_model(nfloat * 4, world, step1) if step1 else None
        comm["efficiency_expected"] = comm["expected"]["weak_scaling_efficiency"] if step1 else None
        # measured weak-scaling efficiency of THIS run: whole-job rate over N times the one-rank rate of the same box
        comm["efficiency_measured"] = (args.steps * world / dt) / (world * one_rank["iters_per_sec"]) if one_rank else None
        if os.environ.get("FSGS_DIST_ONE_GPU") == "1":
            comm["efficiency_note"] = "one-GPU smoke (--smoke): every rank shares one device; not a scaling point"
        comm["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_", "OMP_NUM"))}
        try:
            comm["rccl_info"] = rccl_info_lines()
        except Exception as e:  # noqa: BLE001  (diagnostics must never cost the measurement)
            comm["rccl_info"] = "unavailable: %s" % e

    # ---- extra (N > 1): the same step with the exchange pipelined on the producer side, so that the scaling record
    # shows what hiding the collective behind the per-Gaussian backward and Adam is worth on this fabric ----
    pipelined = None
    if world > 1 and use_fast and not args.no_extras and args.ar_chunks == 1:
        try:
            red4 = fdist.ProducerPipelinedReducer(4)
            tw = time.perf_counter()
            for it in range(5):
                stepper.mapping_step([(rank + it * world) % n_frames], reduce_compact=red4, collect_stats=not args.no_stats)
            barrier()
            # a transport that handles small collectives badly (gloo in the one-GPU smoke) gets 5 timed steps, not 40
            slow = torch.tensor([float((time.perf_counter() - tw) / 5 > 5 * dt / args.steps)], device=device)
            torch.distributed.all_reduce(slow, op=torch.distributed.ReduceOp.MAX)
            # as many steps as the timed loop above, bracketed the same way (reported beside the configured route, never
            # instead of it: `value` is the route --ar-chunks / --ar-algo name)
            npipe = 5 if float(slow.item()) > 0 else args.steps
            tp = time.perf_counter()
            for it in range(npipe):
                stepper.mapping_step([(rank + it * world) % n_frames], reduce_compact=red4, collect_stats=not args.no_stats)
            barrier()
            dtp = time.perf_counter() - tp
            tt = torch.tensor([dtp], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            pipelined = {"ar_chunks": 4, "steps": npipe, "seconds": float(tt.item()),
                         "ms_per_step": float(tt.item()) / npipe * 1e3, "iters_per_sec": npipe * world / float(tt.item()),
                         "what": "dist.ProducerPipelinedReducer(4): all-reduce of row chunk i beside the production of chunk "
                                 "i+1 and the Adam of chunk i-1 (python bench.py --ar-chunks 4 makes it the timed route); timed "
                                 "AFTER the configured route, on a cloud that has taken 2 K further steps"}
        except Exception as e:  # noqa: BLE001  (an extra with collectives of its own must not cost the line)
            pipelined = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sc, cam, pc.active_sh_degree)

    if rank == 0:
        iters = args.steps * world
        # N > 1: `value` is the route the command line configured (--ar-algo / --ar-chunks; default: one RCCL all_reduce).  The
        # x4 producer-pipelined route is timed behind it over the same K steps and reported beside it -- never instead of it:
        # it runs later, on a further-trained cloud, and roofline / kernels_ms describe the first loop
        plain_route = "pipelined x%d" % args.ar_chunks if args.ar_chunks > 1 else args.ar_algo
        exchange, routes = (plain_route if world > 1 else None), None
        if world > 1:
            routes = {plain_route: dt / args.steps * 1e3}
            if pipelined is not None and pipelined.get("steps") == args.steps:
                routes["pipelined x4"] = pipelined["ms_per_step"]
        out = {
            "metric": "train_iters_per_sec", "value": iters / dt, "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: mapping iteration, %dx%d, %d Gaussians (%s scene), 1 camera/rank" % (
                args.config, W, H, P, CONFIGS[args.config][3]),
                "num_rendered": R, "upstream_num_rendered": upstream_main,
                "scene": args.scene, "densify_every": args.densify_every or None, "gaussians_at_end": pc.num_points,
                "target_texture": texture or None,
                "step_forward": "all six planes of both passes; the SH colours of the updated parameters are left behind by "
                                "the previous step's Adam kernel (colour cache: evaluated once per step, on the updated "
                                "parameters)" if use_fast else None,
                "fused_render": fused, "hip_losses": hip_losses,
                # which flavour of the blend kernels ran (waves per 16x16 tile; fsgs_blend_waves_per_tile on THIS device and
                # the flags in force): the backward's summation order depends on it, so every A/B and parity log states it
                "blend_waves_per_tile": blend_flavours(W, H),
                "step_driver": "fast_step (one C-ABI call per stage, no autograd)" if use_fast else "torch.autograd",
                "optimizer": ("Adam on all 59 floats/Gaussian every step: fused into the render-backward kernel (fsgs_render_backward_adam)"
                              if (use_fast and world == 1) else "Adam on all 59 floats/Gaussian every step, from the all-reduced compact [P,14] gradient (fsgs_adam_step_compact)"
                              if use_fast else "FusedAdam / torch path"),
                "parallelism": "dp%d" % world, "exchange": exchange, "loss": float(loss),
                # the headline scene is the lighter of the two this line measures: the same step on the
                # dense scene (upstream pair count ~ SURVEY s8d's nominal 3.0 M) and the progressive phase's two-view step
                "also_measured": {
                    "dense_scene_ms_per_step": None if dense is None else dense["ms_per_step"],
                    "dense_scene_iters_per_sec": None if dense is None else dense["iters_per_sec"],
                    "dense_scene_upstream_num_rendered": None if dense is None else dense["upstream_num_rendered"],
                    "two_view_mapping_ms_per_iter": None if two_view is None else two_view["ms_per_iter"],
                    "tracking_ms_per_iter": None if tracking is None else tracking["ms_per_iter"],
                    "harness_global_run_ms_per_iter": (harness or {}).get("global", {}).get("ms_per_iter"),
                    "harness_progressive_ms_per_frame": (harness or {}).get("progressive", {}).get("ms_per_frame"),
                    "drop_in_unchanged_ms_per_step": (drop_in or {}).get("unchanged", {}).get("ms_per_step")}},
            "roofline": roofline, "cpu_baseline": cpu, "kernels_ms": kernels,
            "raster_fwd_bwd_ms": None if raster is None else raster["raster_fwd_bwd_ms"], "raster": raster,
            "tracking_step": tracking, "two_view_mapping_step": two_view, "dense_scene": dense, "harness": harness, "drop_in_step": drop_in, "densify": densify_log or None, "comm": comm,
            "comm_pipelined": pipelined, "exchange_routes_ms_per_step": routes,
            "timed_blocks": timed_blocks,
            "extras_incomplete": False,
        }
        print(json.dumps(out), flush=True)
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
