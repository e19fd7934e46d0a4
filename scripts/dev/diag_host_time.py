"""Where does the HOST spend a mapping step, and is it ever behind the GPU?  Wraps the step driver's C-ABI calls with
perf_counter stamps (steady-state asynchronous run, C2).  `fwd` contains the mailbox wait for R (host idle while the GPU
works): a large value there means the host is ahead of the GPU; the other segments are pure host work.
    gpurun -- 'python scripts/dev/diag_host_time.py [--tracking]'"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402


def main():
    tracking = "--tracking" in sys.argv
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    np.random.seed(0)
    from fsgs_amd import _lib
    from fsgs_amd.fast_step import FastStepper

    _lib.load()
    pc, poses, frames, cam, sc = bench.build_problem("C2", dev, 0, 1)
    st = FastStepper(pc, poses, frames)
    stamps = []
    lib = st.lib

    class Wrapped:
        """attribute proxy over the CDLL: every fsgs_* call leaves (name, t_entry, t_return)"""

        def __getattr__(self, name):
            f = getattr(lib, name)
            if not name.startswith("fsgs_"):
                return f

            def g(*a):
                t0 = time.perf_counter()
                r = f(*a)
                stamps.append((name, t0, time.perf_counter()))
                return r

            return g

    st.lib = Wrapped()
    n_frames = len(frames.colors)
    if tracking:
        from fsgs_amd.trainer import make_flow_targets  # noqa: F401
    steps = 300

    def run(n, it0=0):
        for it in range(it0, it0 + n):
            stamps.append(("step", time.perf_counter(), 0.0))
            st.mapping_step([it % n_frames])

    run(30)
    torch.cuda.synchronize()
    stamps.clear()
    t0 = time.perf_counter()
    run(steps, 30)
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("steps %d: wall %.1f us/step (host finished enqueueing %.1f us before the GPU finished)" % (
        steps, (t1 - t0) / steps * 1e6, (t1 - t_enq) * 1e6))
    # per-step segments
    seg = {}
    cur = None
    last = None
    for name, a, b in stamps:
        if name == "step":
            if cur is not None and last is not None:
                seg.setdefault("tail -> next step", []).append(a - last)
            cur = a
            last = a
            continue
        seg.setdefault("before " + name, []).append(a - last)
        seg.setdefault("inside " + name, []).append(b - a)
        last = b
    for k, v in seg.items():
        print("%-55s mean %7.1f us  median %7.1f us" % (k, np.mean(v) * 1e6, np.median(v) * 1e6))


if __name__ == "__main__":
    main()
