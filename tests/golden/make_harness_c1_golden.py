"""Writes tests/golden/harness_c1.npz: BASELINE.json configs[0] AS STATED -- "first 8 frames, 20k init Gaussians, 640x512" --
taken end to end through the CPU-oracle harness (tests/ref_harness.py: oracle rasteriser + reference-pinned torch losses +
torch Adam / MultiStepLR + the reference's densify sequence) with the reference's OWN schedule (train.py:318-345: 200
mapping iterations on frame 0, then per frame 50 tracking + 30 two-view mapping iterations, densify_and_prune at iteration
300), and what comes out of it: per-frame losses, the tracked poses, the cloud size, PSNR of the test frame, RPE / ATE.

A trajectory of ~1000 Adam steps is not reproducible number by number between two correct fp32 implementations (see
tests/test_harness_pin_cpu.py for where that starts), so the fixture holds the OUTCOME and -- from a second run of the same
harness with another summation order (8 OpenMP threads instead of 1) -- how far the reference's outcome moves against
itself; tests/test_harness_c1_gpu.py holds fsgs_amd.trainer.Runner to the first within a multiple of the second.

Runs in the dev container (no GPU, no reference import), ~20-30 min:   python tests/golden/make_harness_c1_golden.py"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "free-surgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from fsgs_amd import metrics  # noqa: E402
from oracle.fsgs_oracle import Oracle  # noqa: E402
from tests import ref_harness  # noqa: E402

C1 = dict(W=640, H=512, n_frames=8, P=20_000, tracking_iter=50, mapping_iter=30, first_mapping_iter=200,
          densify_interval=300, rng_seed=11, seed=0)


def outcome(h, frames):
    """what the GPU test compares: see the module docstring"""
    tr = h.trace
    n = len(frames.colors)
    maps = [e for e in tr if e[0] == "map"]
    tracks = [e for e in tr if e[0] == "track"]
    out = {}
    out["track_last"] = np.array([[e[3], e[4], e[5]] for e in tracks if e[2] == C1["tracking_iter"] - 1], np.float64)  # [n-1, 3]
    out["track_first"] = np.array([[e[3], e[4], e[5]] for e in tracks if e[2] == 0], np.float64)
    # mean mapping loss of every mapped frame's block of iterations (frame 0: 200, then 30 each)
    bounds, it = [], 0
    for t in range(n):
        if t in set(int(i) for i in frames.i_train):
            k = C1["first_mapping_iter"] if t == 0 else C1["mapping_iter"]
            bounds.append((t, it, it + k))
            it += k
    ml = np.array([e[3] for e in maps], np.float64)
    out["map_mean"] = np.array([[t, ml[a:b].mean(), ml[b - 1]] for t, a, b in bounds], np.float64)
    out["densify"] = np.array([[e[1], e[2]] for e in tr if e[0] == "densify"], np.int64)
    out["final_P"] = h.pc.num_points
    out["pose_r"] = h.poses.r.detach().cpu().numpy()
    out["pose_t"] = h.poses.t.detach().cpu().numpy()
    with torch.no_grad():
        pred = np.stack([h.poses.get_pose(i).detach().cpu().numpy() for i in range(n)])
    gt = np.stack([np.asarray(g, np.float32) for g in frames.gt_w2c])
    out["pose_metrics"] = np.array(metrics.pose_metrics(pred, gt)[1], np.float64)  # rpe_t, rpe_r (deg), ate
    # PSNR of the test frame(s) at their tracked pose (train.py:401-432 evaluates exactly these)
    ps = []
    with torch.no_grad():
        for i in frames.i_test:
            pkg = h.render(int(i), False, False)
            ps.append(metrics.psnr(frames.colors[int(i)].cpu().numpy()[None], pkg["render"].detach().cpu().numpy()[None]))
    out["psnr_test"] = np.array(ps, np.float64)
    return out


def make_c1_inputs(oracle):
    ratio = C1["P"] / float(C1["W"] * C1["H"])
    return ref_harness.make_inputs(oracle, W=C1["W"], H=C1["H"], n_frames=C1["n_frames"], P_scene=60_000, ratio=ratio, seed=3)


def input_stats(fx):
    """coarse fingerprints of the regenerated sequence: per-frame means of colours / mono-depth / flow, the first points"""
    return np.concatenate([fx["colors_u8"].astype(np.float64).mean(axis=(1, 2, 3)) / 255.0,
                           fx["monodeps_f16"].astype(np.float64).mean(axis=(1, 2)),
                           np.abs(fx["flows_fw_f16"].astype(np.float64)).mean(axis=(1, 2, 3)),
                           fx["_xyz"][:16].astype(np.float64).reshape(-1), fx["_scaling"][:16, 0].astype(np.float64)])


def run(fx, oracle):
    pc, poses, frames = ref_harness.load_inputs(fx, "cpu")
    pc.training_setup(fused=False)
    h = ref_harness.CpuHarness(oracle, pc, poses, frames, tracking_iter=C1["tracking_iter"], mapping_iter=C1["mapping_iter"],
                               first_mapping_iter=C1["first_mapping_iter"], densify_interval=C1["densify_interval"],
                               seed=C1["seed"])
    torch.manual_seed(0)
    with ref_harness.deterministic_rng(C1["rng_seed"]):
        h.progressive_run()
    return h, frames


def main():
    oracle = Oracle(np.float32)
    oracle.set_threads(8)
    fx = make_c1_inputs(oracle)
    print("inputs: P0 %d" % fx["_xyz"].shape[0], flush=True)
    if "--inputs-only" in sys.argv:
        np.savez_compressed("/tmp/harness_c1_inputs.npz", **fx)
        print("input bytes", os.path.getsize("/tmp/harness_c1_inputs.npz"))
        return
    results = []
    for threads in (1, 8):
        oracle.set_threads(threads)
        t0 = time.time()
        h, frames = run(fx, oracle)
        o = outcome(h, frames)
        print("threads %d: %.0f s; densify %s final P %d; pose metrics %s; PSNR %s" % (
            threads, time.time() - t0, o["densify"].tolist(), o["final_P"], o["pose_metrics"].tolist(), o["psnr_test"].tolist()),
            flush=True)
        results.append(o)
    # The inputs are NOT stored (8 frames at 640x512 are 7.8 MB compressed): ref_harness.make_inputs regenerates them from
    # the seed in ~10 s wherever the oracle runs; what is stored of them is enough to tell that it is the same sequence
    out = {"input_stats": input_stats(fx), "P0": fx["_xyz"].shape[0], "config": np.array(sorted(C1.items()), dtype=object).astype(str)}
    for k, v in results[0].items():
        out[k] = v
    for k, v in results[1].items():
        out["alt_" + k] = v  # the same harness, another summation order: the reference's own reproducibility
    np.savez_compressed(os.path.join(HERE, "harness_c1.npz"), **out)
    print("fixture bytes", os.path.getsize(os.path.join(HERE, "harness_c1.npz")))


if __name__ == "__main__":
    main()
