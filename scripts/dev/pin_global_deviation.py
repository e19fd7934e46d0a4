"""Deviation of the pinned harness's GLOBAL phase from the CPU-oracle fixture over repeated runs (companion of
pin_deviation.py).   gpurun -- 'python scripts/dev/pin_global_deviation.py [runs]'"""
import os
import sys

sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "free-surgs_amd")]
import numpy as np  # noqa: E402

import tests.test_harness_pin_gpu as T  # noqa: E402

fx = dict(np.load(T.FX))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    run = T._run_gpu(fx, with_global=True)
    gl = [e for e in run.trace[run.after_progressive["n_trace"]:] if e[0] == "map"]
    got = np.array([e[3] for e in gl])
    rel = np.abs(got - fx["global_map_loss"]) / np.abs(fx["global_map_loss"])
    dens = [[e[1], e[2]] for e in run.trace[run.after_progressive["n_trace"]:] if e[0] == "densify"]
    print("P %d densify %s views %s  loss rel dev per iteration %s" % (run.pc.num_points, dens, [e[2][0] for e in gl],
                                                                    " ".join("%.1e" % r for r in rel)), flush=True)
