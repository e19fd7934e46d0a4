import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
import numpy as np, torch
from oracle.fsgs_oracle import Oracle
from fsgs_amd import synth, rasterizer
from fsgs_amd.trainer import settings_from_cam
from tests.util import sh0_colors, to_camera_frame, c1_poses
import tests.test_raster_gpu as T
o = Oracle(np.float32)
W, H, P = 640, 512, 20000
cam = synth.make_camera(W, H); sc = synth.init_scene(W, H, P, seed=0); s, r, op = synth.activate(sc); col = sh0_colors(sc)
for pi, w2c in enumerate(c1_poses()):
    xyz = to_camera_frame(sc["_xyz"], w2c)
    dL = (np.random.default_rng(pi).uniform(-1, 1, (3, H, W)) / (3 * H * W)).astype(np.float32)
    img, dep, radii, g = T._run_hip(cam, xyz, col, op.reshape(-1), s, r, dL)
    amp, (oi, od, orad, og, ost) = o.flip_amplitudes(cam, xyz, col, op.reshape(-1), s, r, dL)
    floor = 1e-3 * max(float(np.abs(v).max()) for v in og.values())
    for k in ("means3D", "means2D", "colors", "opacities", "scales"):
        a, b = g[k].reshape(P, -1), og[k].reshape(P, -1)
        scale = np.abs(b).max() + floor
        rogue = np.abs(a - b) > 1e-4 * scale + 2 * amp[k]
        for gi in np.unique(np.argwhere(rogue)[:, 0])[:2]:
            print("pose", pi, k, "rogue Gaussian", gi, "hip", a[gi], "oracle", b[gi], "amp", amp[k][gi], "radius", orad[gi], "opacity", op.reshape(-1)[gi])
            oxy, oco = ost.xy(), ost.conic_opacity()
            cx, cy, rad = oxy[gi, 0], oxy[gi, 1], orad[gi]
            x0, x1, y0, y1 = int(max(0, cx - rad - 1)), int(min(W, cx + rad + 2)), int(max(0, cy - rad - 1)), int(min(H, cy + rad + 2))
            d = np.abs(img - oi)[:, y0:y1, x0:x1].max(axis=0)
            ys, xs = np.nonzero(d > 2e-6)
            print("   footprint px with |img diff| > 2e-6:", [(int(y + y0), int(x + x0), float(d[y, x])) for y, x in zip(ys, xs)][:6], " amp there:", [float(amp["image"][:, y + y0, x + x0].max()) for y, x in zip(ys, xs)][:6])
            gxn = (W + 15) // 16
            for y, x in list(zip(ys + y0, xs + x0))[:3]:
                tile = (y // 16) * gxn + x // 16
                r0, r1 = ost.ranges()[tile]; ids = ost.point_list()[r0:r1]
                Tt = 1.0
                for kk, gg in enumerate(ids):
                    dx = np.float32(oxy[gg, 0]) - np.float32(x); dy = np.float32(oxy[gg, 1]) - np.float32(y)
                    A, B, Cc, oo = oco[gg]
                    power = np.float32(-0.5) * (A * dx * dx + Cc * dy * dy) - B * dx * dy
                    if power > 0: continue
                    a_raw = oo * np.exp(power)
                    if abs(a_raw * 255 - 1) < 2e-2: print("      px", y, x, "k", kk, "g", gg, "alpha*255 = %.6f" % (a_raw * 255), "T", Tt)
                    if a_raw >= 1 / 255:
                        tt = Tt * (1 - min(0.99, a_raw))
                        if abs(tt / 1e-4 - 1) < 2e-2: print("      px", y, x, "k", kk, "T' / 1e-4 = %.6f" % (tt / 1e-4))
                        if tt < 1e-4: break
                        Tt = tt
