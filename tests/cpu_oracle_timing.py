"""Not a test: times the CPU oracle's rasteriser forward / backward (the `cpu_baseline` of SURVEY.md s8d) at C1 with one
thread and with all cores, and at C2 with all cores.  Run on the GPU box's host:  python tests/cpu_oracle_timing.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
from fsgs_amd import synth  # noqa: E402
from oracle.fsgs_oracle import Oracle  # noqa: E402


def knn_cpu(pts):
    o = Oracle(np.float32)
    return o.knn_meandist2(np.asarray(pts, np.float32)) if len(pts) <= 40000 else None


def scene(W, H, P, trained):
    # scales from a spacing estimate when the brute-force KNN oracle would take too long
    knn = lambda pts: (knn_cpu(pts) if len(pts) <= 40000 else np.full((len(pts),), (1.0 / 1035.0 * 1280 / W * 3.0) ** 2, np.float32))
    sc = (synth.trained_like_scene if trained else synth.init_scene)(W, H, P, seed=0, knn_fn=knn)
    return sc


def run(tag, W, H, P, trained, threads, reps):
    o = Oracle(np.float32)
    if not threads:  # "all cores" = what the cgroup quota really grants
        from oracle.fsgs_oracle import usable_cores

        threads = min(usable_cores(), o.max_threads())
    o.set_threads(threads)
    sc = scene(W, H, P, trained)
    cam = synth.make_camera(W, H)
    s, r, op = synth.activate(sc)
    col = np.clip(sc["_features_dc"][:, 0, :] * synth.SH_C0 + 0.5, 0, None).astype(np.float32)
    dL = (np.random.default_rng(0).uniform(-1, 1, (3, H, W)) / (3 * H * W)).astype(np.float32)
    tf = tb = 0.0
    for _ in range(reps):
        t0 = time.time()
        img, dep, radii, st = o.raster_forward(cam, sc["_xyz"], col, op.reshape(-1), s, r)
        t1 = time.time()
        o.raster_backward(st, dL)
        tf += t1 - t0
        tb += time.time() - t1
    print("%-28s threads %3d : forward %8.1f ms  backward %8.1f ms  (R = %d rect-based pairs)" % (
        tag, threads, 1e3 * tf / reps, 1e3 * tb / reps, st.num_rendered), flush=True)


if __name__ == "__main__":
    run("C1 640x512 P=20k init", 640, 512, 20000, False, 0, 5)
    run("C2 1280x1024 P=300k trained", 1280, 1024, 300000, True, 0, 2)
    run("C1 640x512 P=20k init", 640, 512, 20000, False, 1, 2)
