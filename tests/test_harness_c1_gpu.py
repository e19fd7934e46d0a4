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


_inputs_cache = {}


def _c1_inputs(oracle32):
    """the 8 frames of C1, regenerated once per session (~10 s of oracle time)"""
    from oracle.fsgs_oracle import usable_cores

    if "inputs" not in _inputs_cache:
        oracle32.set_threads(usable_cores())
        try:
            _inputs_cache["inputs"] = ref_harness.make_c1_inputs(oracle32)
        finally:
            oracle32.set_threads(1)
    return _inputs_cache["inputs"]


def _run_c1(inputs):
    """Runner.progressive_run on C1 as stated -> (outcome, frames, the run's full trace and final state for bitwise checks)"""
    from fsgs_amd.render import render
    from fsgs_amd.trainer import Runner

    C1 = ref_harness.C1
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
    final = {k: v.detach().cpu().clone() for k, v in run.pc.params.items()}
    final["r"], final["t"] = run.poses.r.detach().cpu().clone(), run.poses.t.detach().cpu().clone()
    return got, frames, list(run.trace), final


def test_runner_reaches_the_outcome_of_the_cpu_oracle_harness_on_c1_as_stated(oracle32):
    fx = dict(np.load(FX, allow_pickle=True))
    C1 = ref_harness.C1
    inputs = _c1_inputs(oracle32)
    # the same sequence as the one the fixture's outcome belongs to (another host CPU may round a pixel of the 8-bit frames
    # the other way: coarse statistics, not a checksum)
    np.testing.assert_allclose(ref_harness.c1_input_stats(inputs), fx["input_stats"], rtol=2e-3, atol=2e-4)
    assert inputs["_xyz"].shape[0] == int(fx["P0"]) == C1["P"]

    got, frames, _trace, _final = _run_c1(inputs)
    _check_outcome(got, fx, frames, tight=False, record=True)


# What a C1 run may differ from the CPU-oracle harness by.  Every bound is  3 x |the reference's own second run - its first|
# + a floor  (element-wise); the floors:
#   gross (the product path: float atomics, a different sample of ~1000 chaotic Adam steps every run): ~3x the spread of the
#     HIP harness against ITSELF over 210 runs of the test body (round 4: last tracking loss of a frame +-7 % with 3 runs in
#     another basin at 27-33 %, mean mapping loss +-2.5 %, PSNR 39.3 .. 40.6 dB, RPE_t +-4 %, ATE +-2 %, cloud 20 614 .. 20 626);
#   tight (FSGS_FLAG_DETERMINISTIC: ONE reproducible trajectory, no run-to-run spread to allow for): ~3x what that
#     trajectory is measured to differ by (round 5: per-frame losses <= 1.9 %, their sequence means <= 0.5 %, PSNR +0.20 dB,
#     RPE_t +0.3 %, ATE -0.14 %, cloud 20 615 against 20 613) -- a wrong loss weight of 15 % does not fit in there, see the
#     negative control below.
FLOORS = {
    "gross": dict(per_frame=dict(track_first=0.6, track_last=0.6, map_mean=0.15), means=dict(track_first=0.2, track_last=0.2, map_mean=0.08),
                  # (round 6, 1 750 runs of the test body, scripts/dev/harness_margins.py, profiles/r06_harness_margins.json: with
                  # pose_t_steps = 0.15 / pose_r = 1e-3 the tracked rotations came to 1.19 x their bound and crossed it in 14 of 1 500
                  # runs, the translations to 1.05 x in 1 of 250 -- a one-in-a-hundred red suite; every other bound stays below
                  # 0.77 x.  The two floors now sit at ~1.6 x the largest excursion seen; the deterministic run keeps 2e-4 / 0.03)
                  cloud=0.01, pose_t_steps=0.25, pose_r=2e-3, rpe_ate_steps=0.2, rpe_r_deg=0.05, psnr_db=3.0),
    "tight": dict(per_frame=dict(track_first=0.06, track_last=0.06, map_mean=0.04), means=dict(track_first=0.02, track_last=0.02, map_mean=0.015),
                  cloud=0.002, pose_t_steps=0.03, pose_r=2e-4, rpe_ate_steps=0.03, rpe_r_deg=0.01, psnr_db=0.75),
}


def _check_outcome(got, fx, frames, tight, record=False):
    C1 = ref_harness.C1
    F = FLOORS["tight" if tight else "gross"]
    ref = {k: fx[k] for k in got}
    alt = {k: fx["alt_" + k] for k in got}
    if record:
        summary = {k: (np.asarray(got[k]).tolist(), np.asarray(ref[k]).tolist(), np.asarray(alt[k]).tolist())
                   for k in ("pose_metrics", "psnr_test", "final_P", "densify")}
        print("C1 outcome (HIP, reference, reference's second run):", summary)
        _record(got, ref, alt)

    # the schedule: the densification happened at the same iteration; the cloud size follows the accumulated statistics of
    # ~300 diverging iterations, so it is compared like everything else
    assert got["densify"][:, 0].tolist() == ref["densify"][:, 0].tolist() == [C1["densify_interval"]]
    _within(got["densify"][:, 1], ref["densify"][:, 1], alt["densify"][:, 1], 3.0, F["cloud"] * ref["densify"][:, 1], "cloud size after densify_and_prune")
    _within(got["final_P"], ref["final_P"], alt["final_P"], 3.0, F["cloud"] * float(ref["final_P"]), "final cloud size")
    # per-frame losses: the first tracking iteration of a frame sees the map as the previous frames left it, the last one the
    # optimised pose; per mapped frame the mean and the last mapping loss
    # (the LAST tracking loss of a single frame is the noisiest number of a product-path run: in 3 of 210 runs of this test
    # body the last two frames ended 27-33 % above the reference's value -- another basin of the 7-parameter pose fit --,
    # otherwise within 17 %; so per frame only a gross bar there, and the sequence mean, where a wrong weight or schedule
    # would still show, tighter)
    for k in ("track_first", "track_last", "map_mean"):
        cols = slice(1, None) if k == "map_mean" else slice(None)
        _within(got[k][:, cols], ref[k][:, cols], alt[k][:, cols], 3.0, F["per_frame"][k] * np.abs(ref[k][:, cols]) + 1e-5, k)
        gm, rm, am = (np.asarray(v[k][:, cols]).mean(axis=0) for v in (got, ref, alt))
        _within(gm, rm, am, 3.0, F["means"][k] * np.abs(rm) + 1e-5, k + " (mean over the frames)")
    # the tracked trajectory: translation error of every frame against the reference's, in units of the ground-truth step
    gt = np.stack([np.asarray(g, np.float32) for g in frames.gt_w2c])
    step = float(np.mean([np.linalg.norm(gt[i + 1][:3, 3] - gt[i][:3, 3]) for i in range(len(gt) - 1)]))
    _within(got["pose_t"], ref["pose_t"], alt["pose_t"], 3.0, F["pose_t_steps"] * step, "tracked translations")
    # (the stored quaternions are NOT unit: LearnPose normalises on use, Adam moves the raw parameter, and its norm is a gauge
    # that drifts with the run -- 1.049 against 1.058 in one HIP run -- so the rotations are compared normalised)
    unit = lambda q: q / np.linalg.norm(q, axis=1, keepdims=True)
    _within(unit(got["pose_r"]), unit(ref["pose_r"]), unit(alt["pose_r"]), 3.0, F["pose_r"], "tracked rotations (unit quaternions)")
    # train.py:492-506 / 401-432: RPE_t, RPE_r (degrees), ATE and the PSNR of the test frame
    _within(got["pose_metrics"], ref["pose_metrics"], alt["pose_metrics"], 3.0,
            np.array([F["rpe_ate_steps"] * step, F["rpe_r_deg"], F["rpe_ate_steps"] * step]), "RPE / ATE")
    _within(got["psnr_test"], ref["psnr_test"], alt["psnr_test"], 3.0, F["psnr_db"], "PSNR of the test frame (dB)")
    # (no absolute bar on the ATE: the mono-depth of every frame is min-max normalised on its own, as the reference's loader
    # does, scene/pose_optimizer.py:406-407, so the map's gauge is not a similarity of the ground truth and the REFERENCE's
    # own ATE on this sequence is about one ground-truth step -- what is asserted is that the HIP harness lands where it does)


def _record(got, ref, alt, test="harness_c1"):
    import json

    from tests.util import dump_attribution_log

    rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                                    (np.abs(np.asarray(b, np.float64)) + 1e-12)))
    rec = {"test": test}
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
    dump_attribution_log("r06_harness_c1", json.loads(json.dumps(rec)))


def test_deterministic_c1_runs_are_bit_identical_and_sit_close_to_the_reference(oracle32):
    """FSGS_FLAG_DETERMINISTIC: two whole C1 runs -- 730 iterations (380 cloud + 350 pose Adam steps), a densification, seven tracked frames
    -- leave the SAME bits (every traced loss, the final cloud, the poses), and with the run-to-run noise gone the outcome
    is held to the tight floors: a few per cent per frame, 2 % on the sequence means, 0.75 dB."""
    from fsgs_amd import rasterizer

    fx = dict(np.load(FX, allow_pickle=True))
    inputs = _c1_inputs(oracle32)
    prev = rasterizer.set_deterministic(True)
    try:
        a = _run_c1(inputs)
        b = _run_c1(inputs)
    finally:
        rasterizer.set_deterministic(prev)
    assert len(a[2]) == len(b[2]) > 700  # 200 + 6 x 30 mapping and 7 x 50 tracking iterations + the densification
    for ea, eb in zip(a[2], b[2]):
        assert tuple(ea) == tuple(eb), (ea, eb)
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
    _check_outcome(a[0], fx, a[1], tight=True)
    _record(a[0], {k: fx[k] for k in a[0]}, {k: fx["alt_" + k] for k in a[0]}, test="harness_c1_deterministic")


@pytest.mark.parametrize("which,factor", [("rgb", 1.15), ("rgb", 0.87), ("local_pearson", 2.0)])
def test_a_wrong_loss_weight_fails_the_tight_check(oracle32, which, factor):
    """negative control of the tight floors: the mapping loss with one weight off (train.py:254-258: 5 rgb + 0.05 pearson +
    0.15 local pearson) must NOT pass as the reference's outcome"""
    from fsgs_amd import fast_step, rasterizer, trainer

    fx = dict(np.load(FX, allow_pickle=True))
    inputs = _c1_inputs(oracle32)
    prev = rasterizer.set_deterministic(True)
    old = (fast_step.LOSS_W_MAPPING[which], trainer.LOSS_W_MAPPING[which])
    fast_step.LOSS_W_MAPPING[which] = old[0] * factor
    trainer.LOSS_W_MAPPING[which] = old[1] * factor
    try:
        got, frames, _trace, _final = _run_c1(inputs)
    finally:
        fast_step.LOSS_W_MAPPING[which], trainer.LOSS_W_MAPPING[which] = old
        rasterizer.set_deterministic(prev)
    with pytest.raises(AssertionError):
        _check_outcome(got, fx, frames, tight=True)
