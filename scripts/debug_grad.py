import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
import numpy as np, torch
from fsgs_amd import synth, rasterizer
from oracle.fsgs_oracle import Oracle
from fsgs_amd.trainer import settings_from_cam
from tests.util import sh0_colors
dev = "cuda:0"
W, H, P = 640, 512, 20000
o32, o64 = Oracle(np.float32), Oracle(np.float64)
sc = synth.init_scene(W, H, P, seed=0)
cam = synth.make_camera(W, H)
s, r, o = synth.activate(sc)
col = sh0_colors(sc)
T = lambda a: torch.tensor(a, device=dev)
cfg = rasterizer.make_cfg(settings_from_cam(cam, dev), 3)
m3, c, op, sc_t, r_t = T(sc["_xyz"]), T(col), T(o.reshape(-1)), T(s), T(r)
img, depth, radii, st = rasterizer.raster_forward(cfg, m3, c, op, sc_t, r_t)
dL = (np.random.default_rng(0).uniform(-1, 1, (3, H, W)) / (3 * H * W)).astype(np.float32)
g = rasterizer.raster_backward(st, m3, c, sc_t, r_t, radii, T(dL))
names = ["means2D", "colors", "opacities", "means3D", "scales", "rotations"]
hip = {k: v.cpu().numpy() for k, v in zip(names, g)}
hip["opacities"] = hip["opacities"].reshape(-1)
v = {k: t.cpu().numpy() for k, t in rasterizer.state_views(st).items()}
res = {}
for tag, oc in (("o32", o32), ("o64", o64)):
    oi, od, orad, ost = oc.raster_forward(cam, sc["_xyz"], col, o.reshape(-1), s, r)
    og = oc.raster_backward(ost, dL)
    res[tag] = (oi, ost, og)
    print(tag, "img err", np.abs(img.cpu().numpy() - oi).max(), "radii mism", (radii.cpu().numpy() != orad).sum(),
          "R", st.num_rendered, ost.num_rendered,
          "n_contrib mism", (v["n_contrib"] != ost.n_contrib()).sum(), "finalT err", np.abs(v["final_T"] - ost.final_T()).max(),
          "plist equal", np.array_equal(v["point_list"].astype(np.uint32), ost.point_list()))
    for k in names:
        a, b = hip[k], og[k]
        e = np.abs(a - b); i = np.unravel_index(np.argmax(e), e.shape)
        print("   %-10s err/inf %.3e at %s hip %.5e ref %.5e" % (k, e.max() / (np.abs(b).max() + 1e-30), i, a[i], b[i]))
og32, og64 = res["o32"][2], res["o64"][2]
for k in names:
    print("o32 vs o64 %-10s %.3e" % (k, np.abs(og32[k] - og64[k]).max() / (np.abs(og64[k]).max() + 1e-30)))
