"""Premise test for moving the SH-rest Adam stream off the step's critical path (round 4): how much do the forward's
latency-bound kernels (per-Gaussian preprocess, scatter, tile sort) and a concurrent Adam-from-compact-gradient launch on a
second stream slow each other down?   gpurun -- 'python scripts/dev/overlap_probe.py'"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
from fsgs_amd import _lib  # noqa: E402
from fsgs_amd.fast_step import FastStepper  # noqa: E402

dev = torch.device("cuda", 0)
pc, poses, frames, cam, sc = bench.build_problem("C2", dev, 0, 1)
fs = FastStepper(pc, poses, frames)
for it in range(10):
    fs.mapping_step([it % 8])
torch.cuda.synchronize()
lib = _lib.load()
P = pc.num_points
gc = torch.zeros((P, 14), device=dev)
side = torch.cuda.Stream()
GROUPS = ["render_pre_fwd", "sort_depth", "sort_tile", "blend_fwd", "adam"]


def run(with_adam, n=60):
    _lib.profile_enable(GROUPS, stride=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(n):
        w2c = poses.get_pose_detached(it % 8)
        if with_adam:
            adam = fs._fused_adam_struct()
            args = fs._last_args
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                _lib.check(lib.fsgs_adam_step_compact(P, C.byref(args), _lib.ptr(gc), C.byref(adam), _lib.current_stream()), "adam")
        a = fs._render_forward(w2c, fs.buf)
        fs._last_args = a[0]
        torch.cuda.synchronize()  # one forward (+ one Adam) at a time: the question is the slowdown inside the overlap
    dt = (time.perf_counter() - t0) / n * 1e3
    pr = _lib.profile_read()
    _lib.profile_enable([])
    return dt, {k: pr[k][0] / pr[k][1] * 1e3 for k in GROUPS if k in pr and pr[k][1]}


fs._last_args = fs._render_forward(poses.get_pose_detached(0), fs.buf)[0]
torch.cuda.synchronize()
for rep in range(2):
    for mode in (False, True):
        dt, k = run(mode)
        print("%-22s wall %.3f ms/iter  %s" % ("forward + Adam beside it" if mode else "forward alone", dt,
                                            "  ".join("%s %.1f" % (n, v) for n, v in k.items())), flush=True)
