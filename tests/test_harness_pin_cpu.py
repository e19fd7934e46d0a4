"""tests/golden/harness_pin.npz is what the CPU-oracle harness (tests/ref_harness.py) makes of its own inputs: re-run
it here from the stored inputs and demand the stored trajectory back (guards the fixture against drifting away from
the harness that defines it; the GPU side is tests/test_harness_pin_gpu.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_cpu_oracle_harness_regenerates_the_fixture(oracle32):
    import make_harness_golden as M

    fx = dict(np.load(os.path.join(HERE, "golden", "harness_pin.npz")))
    oracle32.set_threads(1)
    h = M.run(fx, oracle32)
    n = h.after_progressive["n_trace"]
    maps = np.array([e[3] for e in h.trace[:n] if e[0] == "map"])
    tracks = np.array([[e[3], e[4], e[5]] for e in h.trace if e[0] == "track"])
    np.testing.assert_allclose(maps, fx["map_loss"], rtol=1e-6)
    np.testing.assert_allclose(tracks, fx["track_loss"], rtol=1e-6, atol=1e-9)
    assert [[e[1], e[2]] for e in h.trace[:n] if e[0] == "densify"] == fx["densify"].tolist()
    np.testing.assert_allclose(h.after_progressive["pose_t"], fx["pose_t"], atol=1e-7)
    assert h.after_progressive["final_P"] == int(fx["final_P"]) and int(fx["_xyz"].shape[0]) == 983
    # the global phase behind it
    gl = h.trace[n:]
    np.testing.assert_allclose(np.array([e[3] for e in gl if e[0] == "map"]), fx["global_map_loss"], rtol=1e-6)
    assert [e[2][0] for e in gl if e[0] == "map"] == fx["global_map_view"].tolist()
    assert [[e[1], e[2]] for e in gl if e[0] == "densify"] == fx["global_densify"].tolist()
    assert h.pc.num_points == int(fx["global_final_P"])
    # the stored cloud is the INITIAL one (the run must not have written through to the arrays)
    assert float(np.abs(fx["_rotation"][:, 1:]).max()) == 0.0


def test_the_reference_trajectory_is_only_that_reproducible_after_a_densification(oracle32):
    """What "reproduces the recorded trajectory" can mean: the SAME harness with another summation order of the oracle's
    own backward (its tile loop on several OpenMP threads, atomics in arrival order) against its own 1-thread fixture.
    Before the densification the per-iteration losses agree to rounding (1e-6); after it -- children with zero Adam
    moments, first steps of lr * sign(gradient) -- by up to several 1e-4.  tests/test_harness_pin_gpu.py holds the HIP
    harness to ref_harness.POST_DENSIFY_RTOL there, and this test keeps that number honest: the reference must stay
    inside ref_harness.REFERENCE_SELF_RTOL against itself, and the schedule / cloud size must not move at all."""
    import make_harness_golden as M
    from oracle.fsgs_oracle import usable_cores
    from tests import ref_harness

    fx = dict(np.load(os.path.join(HERE, "golden", "harness_pin.npz")))
    n = max(2, min(8, usable_cores()))
    oracle32.set_threads(n)
    try:
        h = M.run(fx, oracle32)
    finally:
        oracle32.set_threads(1)
    maps = np.array([e[3] for e in h.trace[:h.after_progressive["n_trace"]] if e[0] == "map"])
    n_pre = int((fx["map_iter"] < fx["densify"][0, 0]).sum())
    rel = np.abs(maps - fx["map_loss"]) / np.abs(fx["map_loss"])
    print("reference vs itself (%d threads): per-iteration loss off by %.1e before, %.1e after the densification" % (
        n, rel[:n_pre].max(), rel[n_pre:].max()))
    assert [[e[1], e[2]] for e in h.trace[:h.after_progressive["n_trace"]] if e[0] == "densify"] == fx["densify"].tolist()
    assert rel[:n_pre].max() <= 1e-5
    assert rel[n_pre:].max() <= ref_harness.REFERENCE_SELF_RTOL
