"""how many Gaussians of the bench scenes are invisible in a view (radius 0: zero gradient, their Adam update does not depend
on the blend), and how many 256-Gaussian blocks are invisible as a whole"""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", ".."), os.path.join(os.path.dirname(__file__), "..", "..", "free-surgs_amd")]
import torch, bench
from fsgs_amd.fast_step import FastStepper
for cfg in ("C2", "C4", "C1"):
    pc, poses, frames, cam, sc = bench.build_problem(cfg, torch.device("cuda", 0), 0, 1)
    fs = FastStepper(pc, poses, frames)
    for it in range(8):
        fs.mapping_step([it % 8])
        r = fs.last["radii"]
        P = r.numel()
        inv = (r <= 0)
        nb = (P + 255) // 256
        pad = torch.zeros(nb * 256, dtype=torch.bool, device=r.device); pad[:P] = inv; pad[P:] = True
        whole = pad.view(nb, 256).all(dim=1).float().mean().item()
        if it in (0, 3, 7):
            print("%s frame %d: P %d, invisible %.1f %%, wholly invisible 256-blocks %.1f %%" % (cfg, it, P, 100 * inv.float().mean().item(), 100 * whole))
