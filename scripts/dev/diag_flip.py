"""Diagnose un-witnessed HIP-vs-oracle outliers on the C1 scene: which decision at that pixel is closest to its threshold?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
import numpy as np, torch
from oracle.fsgs_oracle import Oracle
from fsgs_amd import synth, rasterizer
from fsgs_amd.trainer import settings_from_cam
from tests.util import sh0_colors, to_camera_frame, c1_poses
o = Oracle(np.float32)
W, H, P = 640, 512, 20000
cam = synth.make_camera(W, H); sc = synth.init_scene(W, H, P, seed=0); s, r, op = synth.activate(sc); col = sh0_colors(sc)
T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device="cuda")
for pi, w2c in enumerate(c1_poses()):
    xyz = to_camera_frame(sc["_xyz"], w2c)
    cfg = rasterizer.make_cfg(settings_from_cam(cam, "cuda"), 3)
    img, depth, radii, st = rasterizer.raster_forward(cfg, T(xyz), T(col), T(op.reshape(-1)), T(s), T(r))
    img = img.cpu().numpy()
    v = {k: t.cpu().numpy() for k, t in rasterizer.state_views(st).items()}
    dL = np.zeros((3, H, W), np.float32)
    amp, (oi, od, orad, og, ost) = o.flip_amplitudes(cam, xyz, col, op.reshape(-1), s, r, dL)
    err = np.abs(img - oi)
    rogue = err > 1e-4 * (np.abs(oi).max() + 1) + 2 * amp["image"]
    print("pose", pi, "outliers", int((err > 2e-4).sum()), "rogue", int(rogue.sum()))
    oxy, oco = ost.xy(), ost.conic_opacity()
    for (c, y, x) in np.argwhere(rogue)[:3]:
        tile = (y // 16) * ((W + 15) // 16) + x // 16
        r0, r1 = ost.ranges()[tile]
        ids = ost.point_list()[r0:r1]
        Tt = 1.0
        print(" pixel", y, x, "err", err[c, y, x], "list", len(ids), "hip n_contrib", v["n_contrib"][y, x], "oracle", ost.n_contrib().reshape(H, W)[y, x])
        for k, g in enumerate(ids):
            dx = np.float32(oxy[g, 0]) - np.float32(x); dy = np.float32(oxy[g, 1]) - np.float32(y)
            A, B, Cc, oo = oco[g]
            power = np.float32(-0.5) * (A * dx * dx + Cc * dy * dy) - B * dx * dy
            if power > 0: continue
            a_raw = oo * np.exp(power)
            alpha = min(0.99, a_raw)
            hx, hy = v["xy"][g]; hA, hB, hC, ho = v["conic_opacity"][g]
            hdx = np.float32(hx) - np.float32(x); hdy = np.float32(hy) - np.float32(y)
            hp = np.float32(-0.5) * (hA * hdx * hdx + hC * hdy * hdy) - hB * hdx * hdy
            ha = ho * np.exp(hp)
            near = abs(a_raw * 255 - 1) < 5e-3 or abs(a_raw - 0.99) < 1e-3
            tt = Tt * (1 - alpha) if alpha >= 1 / 255 else Tt
            nearT = abs(tt / 1e-4 - 1) < 1e-2
            if near or nearT:
                print("   k", k, "g", g, "alpha*255", a_raw * 255, "hip-geom alpha*255", ha * 255, "T'", tt, "dxy", oxy[g] - v["xy"][g], "dconic rel", (oco[g] - v["conic_opacity"][g]) / (np.abs(oco[g]) + 1e-30), "radius", orad[g], "sigma-ish", 1 / np.sqrt(A))
            if alpha >= 1 / 255:
                if tt < 1e-4: break
                Tt = tt
