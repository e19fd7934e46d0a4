"""BASELINE.json configs[0] AS STATED, end to end: "scared_demo first 8 frames, 20k init Gaussians, 640x512" through the
reference's own schedule (train.py:318-345: 200 mapping iterations on frame 0, then per frame 50 tracking + 30 two-view
mapping iterations, densify_and_prune when the iteration counter reaches 300) -- fsgs_amd.trainer.Runner on the HIP step
driver against what the CPU-oracle harness (tests/ref_harness.py) made of the same sequence
(tests/golden/harness_c1.npz, written by tests/golden/make_harness_c1_golden.py).

~1000 Adam steps are not reproducible number by number between two correct fp32 implementations (tests/test_harness_pin_*
pin the FIRST steps that way); what is compared is the outcome -- per-frame losses, tracked poses, cloud size, PSNR of the
test frame, RPE / ATE -- and the yardstick is the reference itself: the fixture also holds the outcome of the SAME CPU
harness with another summation order of its backward (8 OpenMP threads instead of 1), and the HIP run may sit a small
multiple of that distance away (plus a floor where the two reference runs happen to agree closely).

The 8 frames at 640x512 are regenerated from their seed by the oracle (ref_harness.make_c1_inputs, ~10 s on the box's
cores) instead of being stored (7.8 MB compressed); the fixture's coarse fingerprints say it is the same sequence."""
import os

import numpy as np
import pytest
import torch

from tests import ref_harness

pytestmark = pytest.mark.gpu
FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness_c1.npz")


def _within(got, ref, alt, mult, floor, what):
    """|got - ref| <= mult * |alt - ref| + floor, element-wise (alt = the reference's own second run)"""
    got, ref, alt = (np.asarray(a, np.float64) for a in (got, ref, alt))
    bound = mult * np.abs(alt - ref) + floor
    bad = np.abs(got - ref) > bound
    assert not bad.any(), "%s: got %s, reference %s (its own second run %s): off by %s, allowed %s" % (
        what, got[bad][:6], ref[bad][:6], alt[bad][:6], np.abs(got - ref)[bad][:6], bound[bad][:6])


def test_runner_reaches_the_outcome_of_the_cpu_oracle_harness_on_c1_as_stated(oracle32):
    from fsgs_amd.render import render
    from fsgs_amd.trainer import Runner
    from oracle.fsgs_oracle import usable_cores

    fx = dict(np.load(FX, allow_pickle=True))
    C1 = ref_harness.C1
    oracle32.set_threads(usable_cores())
    try:
        inputs = ref_harness.make_c1_inputs(oracle32)
    finally:
        oracle32.set_threads(1)
    # the same sequence as the one the fixture's outcome belongs to (another host CPU may round a pixel of the 8-bit frames
    # the other way: coarse statistics, not a checksum)
    np.testing.assert_allclose(ref_harness.c1_input_stats(inputs), fx["input_stats"], rtol=2e-3, atol=2e-4)
    assert inputs["_xyz"].shape[0] == int(fx["P0"]) == C1["P"]

    pc, poses, frames = ref_harness.load_inputs(inputs, "cuda")
    assert len(frames.colors) == 8 and tuple(frames.colors[0].shape) == (3, 512, 640) and list(frames.i_test) == [4]
    pc.training_setup()
    run = Runner(pc, poses, frames, tracking_iter=C1["tracking_iter"], mapping_iter=C1["mapping_iter"],
                 first_mapping_iter=C1["first_mapping_iter"], densify_interval=C1["densify_interval"], seed=C1["seed"],
                 trace=True)
    torch.manual_seed(0)
    with ref_harness.deterministic_rng(C1["rng_seed"]):
        run.progressive_run()
    torch.cuda.synchronize()

    def render_fn(t):
        with torch.no_grad():
            return render(run.poses, t, run.pc, gs_grad=False, cam_grad=False)["render"]

    got = ref_harness.c1_outcome(run.trace, run.pc, run.poses, frames, render_fn)
    ref = {k: fx[k] for k in got}
    alt = {k: fx["alt_" + k] for k in got}
    summary = {k: (np.asarray(got[k]).tolist(), np.asarray(ref[k]).tolist(), np.asarray(alt[k]).tolist())
               for k in ("pose_metrics", "psnr_test", "final_P", "densify")}
    print("C1 outcome (HIP, reference, reference's second run):", summary)
    _record(got, ref, alt)

    # the schedule: the densification happened at the same iteration; the cloud size follows the accumulated statistics of
    # ~300 diverging iterations, so it is compared like everything else
    assert got["densify"][:, 0].tolist() == ref["densify"][:, 0].tolist() == [C1["densify_interval"]]
    _within(got["densify"][:, 1], ref["densify"][:, 1], alt["densify"][:, 1], 3.0, 0.01 * ref["densify"][:, 1], "cloud size after densify_and_prune")
    _within(got["final_P"], ref["final_P"], alt["final_P"], 3.0, 0.01 * float(ref["final_P"]), "final cloud size")
    # The floors below are ~3x the spread of the HIP harness against ITSELF over three runs of this test body (round 4, one box:
    # last tracking loss of a frame +-7 %, mean mapping loss +-2.5 %, PSNR of the test frame 39.3 .. 40.6 dB, RPE_t +-4 %,
    # RPE_r +-0.004 deg, ATE +-2 %, cloud size 20 614 .. 20 626 -- against the reference's 39.45 dB, 5.27e-3, 0.260 deg, 4.61e-3,
    # 20 613): ~1000 Adam steps amplify the arrival order of the float atomics, in the reference's own second run as well.
    # per-frame losses: the first tracking iteration of a frame sees the map as the previous frames left it, the last one the
    # optimised pose; per mapped frame the mean and the last mapping loss
    # (the LAST tracking loss of a single frame is the noisiest number of a run: in 3 of 210 runs of this test body the last
    # two frames ended 27-33 % above the reference's value -- another basin of the 7-parameter pose fit --, otherwise within
    # 17 %; so per frame only a gross bar, and the sequence mean, where a wrong weight or schedule would still show, tighter)
    for k, floor in (("track_first", 0.6), ("track_last", 0.6), ("map_mean", 0.15)):
        cols = slice(1, None) if k == "map_mean" else slice(None)
        _within(got[k][:, cols], ref[k][:, cols], alt[k][:, cols], 3.0, floor * np.abs(ref[k][:, cols]) + 1e-5, k)
        gm, rm, am = (np.asarray(v[k][:, cols]).mean(axis=0) for v in (got, ref, alt))
        _within(gm, rm, am, 3.0, (0.2 if k != "map_mean" else 0.08) * np.abs(rm) + 1e-5, k + " (mean over the frames)")
    # the tracked trajectory: translation error of every frame against the reference's, in units of the ground-truth step
    gt = np.stack([np.asarray(g, np.float32) for g in frames.gt_w2c])
    step = float(np.mean([np.linalg.norm(gt[i + 1][:3, 3] - gt[i][:3, 3]) for i in range(len(gt) - 1)]))
    _within(got["pose_t"], ref["pose_t"], alt["pose_t"], 3.0, 0.15 * step, "tracked translations")
    # (the stored quaternions are NOT unit: LearnPose normalises on use, Adam moves the raw parameter, and its norm is a gauge
    # that drifts with the run -- 1.049 against 1.058 in one HIP run -- so the rotations are compared normalised)
    unit = lambda q: q / np.linalg.norm(q, axis=1, keepdims=True)
    _within(unit(got["pose_r"]), unit(ref["pose_r"]), unit(alt["pose_r"]), 3.0, 1e-3, "tracked rotations (unit quaternions)")
    # train.py:492-506 / 401-432: RPE_t, RPE_r (degrees), ATE and the PSNR of the test frame
    _within(got["pose_metrics"], ref["pose_metrics"], alt["pose_metrics"], 3.0, np.array([0.2 * step, 0.05, 0.2 * step]), "RPE / ATE")
    _within(got["psnr_test"], ref["psnr_test"], alt["psnr_test"], 3.0, 3.0, "PSNR of the test frame (dB)")
    # (no absolute bar on the ATE: the mono-depth of every frame is min-max normalised on its own, as the reference's loader
    # does, scene/pose_optimizer.py:406-407, so the map's gauge is not a similarity of the ground truth and the REFERENCE's
    # own ATE on this sequence is about one ground-truth step -- what is asserted is that the HIP harness lands where it does)


def _record(got, ref, alt):
    import json

    from tests.util import dump_attribution_log

    rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                                    (np.abs(np.asarray(b, np.float64)) + 1e-12)))
    rec = {"test": "harness_c1"}
    for k in ("track_first", "track_last", "map_mean", "pose_metrics", "psnr_test"):
        rec[k] = {"hip_vs_reference_max_rel": rel(got[k], ref[k]), "reference_vs_itself_max_rel": rel(alt[k], ref[k])}
    unit = lambda q: q / np.linalg.norm(q, axis=1, keepdims=True)
    for k, f in (("pose_t", lambda a: a), ("pose_r", unit)):
        rec[k] = {"hip_vs_reference_max_abs": float(np.abs(f(got[k]) - f(ref[k])).max()),
                  "reference_vs_itself_max_abs": float(np.abs(f(alt[k]) - f(ref[k])).max())}
    rec["final_P"] = [int(got["final_P"]), int(ref["final_P"]), int(alt["final_P"])]
    rec["pose_metrics_values"] = {"hip": np.asarray(got["pose_metrics"]).tolist(), "reference": np.asarray(ref["pose_metrics"]).tolist(),
                                  "reference_second_run": np.asarray(alt["pose_metrics"]).tolist()}
    rec["psnr_values"] = {"hip": np.asarray(got["psnr_test"]).tolist(), "reference": np.asarray(ref["psnr_test"]).tolist(),
                          "reference_second_run": np.asarray(alt["psnr_test"]).tolist()}
    dump_attribution_log("r04_harness_c1", json.loads(json.dumps(rec)))
