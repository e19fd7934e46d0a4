"""Host profile of the autograd-driven step with INTEGRATION.md s3's three edits (fused render + HIP losses + FusedAdam under
loss.backward()): the route is host-bound at C2 (0.94 ms/step against 0.64 for the autograd-free driver), so where the
interpreter spends the step.   gpurun -- 'python scripts/dev/profile_autograd_step.py'"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
from fsgs_amd.trainer import mapping_step  # noqa: E402

pc, poses, frames, cam, sc = bench.build_problem("C2", torch.device("cuda", 0), 0, 1)
n = len(frames.colors)
for it in range(10):
    mapping_step(pc, poses, frames, [it % n])
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(100):
    mapping_step(pc, poses, frames, [it % n])
torch.cuda.synchronize()
print("ms/step %.3f" % ((time.perf_counter() - t0) * 10))
pr = cProfile.Profile()
pr.enable()
for it in range(100):
    mapping_step(pc, poses, frames, [it % n])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
