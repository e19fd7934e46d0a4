"""How large is the densification statistic (xyz_gradient_accum / denom, train.py:298-311) on the bench's synthetic
problem, and what target texture makes part of the cloud cross the reference's 2e-4 threshold?  (GPU box)"""
import sys

sys.path[:0] = [".", "free-surgs_amd"]
import numpy as np
import torch

import bench
from fsgs_amd.fast_step import FastStepper

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
for amp in (0.0, 0.15, 0.3):
    torch.manual_seed(0)
    pc, poses, frames, cam, sc = bench.build_problem(cfg, torch.device("cuda", 0), 0, 1, texture=amp)
    fs = FastStepper(pc, poses, frames)
    for it in range(100):
        fs.mapping_step([it % 8])
    den = pc.variables["denom"].reshape(-1)
    g = (pc.variables["xyz_gradient_accum"].reshape(-1) / den)[den > 0]
    q = torch.quantile(g[torch.randperm(len(g), device=g.device)[:500_000]], torch.tensor([0.5, 0.9, 0.99, 0.999], device=g.device))
    print("%s texture %.2f: seen %d, grad quantiles 50/90/99/99.9 %% = %s, > 2e-4: %d (%.2f %%)" % (
        cfg, amp, len(g), ["%.2e" % v for v in q.tolist()], int((g > 2e-4).sum()), 100.0 * float((g > 2e-4).float().mean())))
    P0 = pc.num_points
    info = pc.densify_and_prune_device(2e-4, 0.05, None)
    print("   densify_and_prune(2e-4, 0.05, None): %d -> %d %s" % (P0, pc.num_points, info))
