"""Tile-list length distribution of a bench scene (what the per-tile sort and the blend kernels see)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "free-surgs_amd"))
import numpy as np, torch
import bench
from fsgs_amd import synth, rasterizer
from fsgs_amd.trainer import settings_from_cam

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C2"
pc, poses, frames, cam, sc = bench.build_problem(cfg_name, "cuda", 0, 1)
s, r, op = synth.activate(sc)
col = np.clip(sc["_features_dc"][:, 0, :] * synth.SH_C0 + 0.5, 0, None).astype(np.float32)
T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device="cuda")
cfg = rasterizer.make_cfg(settings_from_cam(cam, "cuda"), 3)
out = rasterizer.raster_forward(cfg, T(sc["_xyz"]), T(col), T(op.reshape(-1, 1)), T(s), T(r))
st = out[-1] if isinstance(out, tuple) else out
v = rasterizer.state_views(st)
rg = v["ranges"].cpu().numpy()
n = rg[:, 1] - rg[:, 0]
print(cfg_name, "tiles", len(n), "R", int(n.sum()), "mean %.1f" % n.mean(), "max", int(n.max()))
print("percentiles 50/90/99/99.9:", [int(np.percentile(n, p)) for p in (50, 90, 99, 99.9)])
hist = np.bincount(np.minimum(np.ceil(np.log2(np.maximum(n, 1))).astype(int), 12))
print("padded size 2^k histogram:", {1 << k: int(c) for k, c in enumerate(hist) if c})
