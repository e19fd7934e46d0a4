"""Per-frame cost of the pose-independent half of projection_flow_loss (FlowTargets: back-projection + duplicate rejection)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "free-surgs_amd"))
import numpy as np, torch
from fsgs_amd import synth
from fsgs_amd.flow import FlowTargets
W, H = 1280, 1024
K = synth.intrinsics(W, H)
u = torch.arange(W, device="cuda").float()[None] / W
v = torch.arange(H, device="cuda").float()[:, None] / H
depth = (1.0 + 0.3 * torch.sin(6.28 * u) * torch.cos(6.28 * v)).reshape(1, H, W).contiguous()
fl = torch.randn(2, H, W, device="cuda")
rigid = torch.rand(H, W, device="cuda") > 0.05
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tg = FlowTargets(depth, np.eye(4, dtype=np.float32), K, fl, rigid)
    torch.cuda.synchronize(); print("FlowTargets %.2f ms, M = %d" % ((time.perf_counter() - t0) * 1e3, tg.pts.shape[0]))
