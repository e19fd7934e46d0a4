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
    maps = np.array([e[3] for e in h.trace if e[0] == "map"])
    tracks = np.array([[e[3], e[4], e[5]] for e in h.trace if e[0] == "track"])
    np.testing.assert_allclose(maps, fx["map_loss"], rtol=1e-6)
    np.testing.assert_allclose(tracks, fx["track_loss"], rtol=1e-6, atol=1e-9)
    assert [[e[1], e[2]] for e in h.trace if e[0] == "densify"] == fx["densify"].tolist()
    np.testing.assert_allclose(h.poses.t.detach().numpy(), fx["pose_t"], atol=1e-7)
    assert h.pc.num_points == int(fx["final_P"]) and int(fx["_xyz"].shape[0]) == 983
    # the stored cloud is the INITIAL one (the run must not have written through to the arrays)
    assert float(np.abs(fx["_rotation"][:, 1:]).max()) == 0.0
