"""Why does one image element of a sweep seed differ between the HIP rasteriser and the oracle without a witness?
For the worst un-witnessed pixel: every Gaussian of its tile list with the decisions (power > 0, alpha < 1/255, T < 1e-4)
evaluated in float64 from the ORACLE's fp32 state and from the HIP state, side by side.
   python scripts/dev/diag_pixel.py <seed>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
import numpy as np, torch
from oracle.fsgs_oracle import Oracle
from fsgs_amd import rasterizer
from fsgs_amd.trainer import settings_from_cam
import tests.test_raster_gpu as TR

seed = int(sys.argv[1])
cam, xyz, col, op, s, r = TR.sweep_scene(seed)
H, W, C = cam["image_height"], cam["image_width"], col.shape[1]
print("seed", seed, "W H P C", W, H, len(xyz), C)
o = Oracle(np.float32)
T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device="cuda")
cfg = rasterizer.make_cfg(settings_from_cam(cam, "cuda"), C)
img, depth, radii, st = rasterizer.raster_forward(cfg, T(xyz), T(col), T(op.reshape(-1)), T(s), T(r))
img = img.cpu().numpy()
v = {k: t.cpu().numpy() for k, t in rasterizer.state_views(st).items()}
dL = np.zeros((C, H, W), np.float32)
amp, (oi, od, orad, og, ost) = o.flip_amplitudes(cam, xyz, col, op.reshape(-1), s, r, dL)
err = np.abs(img - oi)
rogue = err > 1e-4 * (np.abs(oi).max() + 1) + 2 * amp["image"]
print("rogue elements", int(rogue.sum()), "radii differ at", np.nonzero(radii.cpu().numpy() != orad)[0][:10])
oxy, oco, odep = ost.xy().astype(np.float64), ost.conic_opacity().astype(np.float64), ost.depth()
hxy, hco, hdep = v["xy"].astype(np.float64), v["conic_opacity"].astype(np.float64), v["depth"]
gx = (W + 15) // 16
for (c, y, x) in np.argwhere(rogue)[:2]:
    tile = (y // 16) * gx + x // 16
    r0, r1 = ost.ranges()[tile]
    ids = ost.point_list()[r0:r1].astype(np.int64)
    h0, h1 = v["ranges"][tile]
    hids = v["point_list"][h0:h1].astype(np.int64)
    print("pixel", (c, y, x), "hip", img[c, y, x], "oracle", oi[c, y, x], "tile", tile, "oracle list", len(ids), "hip list", len(hids))
    print("  hip final_T %.7g n_contrib %d | oracle final_T %.7g n_contrib %d" % (
        v["final_T"][y, x], v["n_contrib"][y, x], ost.final_T()[y, x], ost.n_contrib()[y, x]))
    print("  ids only in oracle list:", sorted(set(ids) - set(hids))[:20], " only in hip list:", sorted(set(hids) - set(ids))[:20])
    order_same = [g for g in ids if g in set(hids)] == [g for g in hids if g in set(ids)]
    print("  common ids in the same order:", order_same)
    To = Th = 1.0
    for k, g in enumerate(ids):
        def ev(xy, co):
            dx, dy = xy[g, 0] - x, xy[g, 1] - y
            power = -0.5 * (co[g, 0] * dx * dx + co[g, 2] * dy * dy) - co[g, 1] * dx * dy
            a = co[g, 3] * np.exp(power)
            return power, a
        po, ao = ev(oxy, oco)
        ph, ah = ev(hxy, hco)
        near = []
        if abs(po) < 1e-4 or abs(ph) < 1e-4: near.append("power~0")
        if abs(ao * 255 - 1) < 2e-3: near.append("alpha~1/255 (%.2e)" % (ao * 255 - 1))
        if abs(min(ao, .99) - 0.99) < 1e-3 and ao > 0.98: near.append("alpha~0.99")
        co_ = po <= 0 and ao >= 1 / 255.0
        ch_ = ph <= 0 and ah >= 1 / 255.0
        if co_:
            tt = To * (1 - min(0.99, ao))
            if abs(tt / 1e-4 - 1) < 1e-2: near.append("T~1e-4 (%.3e)" % tt)
        mark = "" if (co_ == ch_ and not near) else "   <-- " + " ".join(near) + ("" if co_ == ch_ else " DECISION DIFFERS")
        if mark or k < 3:
            print("   #%d g=%d depth o/h %.7g/%.7g alpha o/h %.6g/%.6g (rel diff %.1e) contributes o/h %s/%s T_o %.4g%s" % (
                k, g, odep[g], hdep[g], ao, ah, (ah - ao) / max(ao, 1e-30), co_, ch_, To, mark))
        if co_:
            tt = To * (1 - min(0.99, ao))
            if tt < 1e-4: print("   oracle stops at #%d" % k); break
            To = tt
print("per-Gaussian state, oracle f32 / HIP / oracle f64:")
o64 = Oracle(np.float64)
st64 = o64.raster_forward(cam, xyz, col, op.reshape(-1), s, r)[3]
xy64, co64 = st64.xy(), st64.conic_opacity()
for g in range(min(len(xyz), 4)):
    A, B, Cc = co64[g, :3]
    det_c = A * Cc - B * B
    a, b, c = Cc / det_c, -B / det_c, A / det_c
    print(" g=%d scale %s op %.4g radius %d  cov2D (a,b,c)=(%.6g, %.6g, %.6g) det %.6g  a*c/det = %.3g" % (
        g, s[g], op[g], orad[g], a, b, c, a * c - b * b, a * c / (a * c - b * b)))
    print("    xy     o32 %s hip %s o64 %s" % (oxy[g], hxy[g], xy64[g]))
    print("    conic  o32 %s\n           hip %s\n           o64 %s" % (oco[g, :3], hco[g, :3], co64[g, :3]))
    print("    rel err of the conic vs f64: oracle-f32 %s hip %s" % (np.abs(oco[g, :3] - co64[g, :3]) / np.abs(co64[g, :3]),
                                                                       np.abs(hco[g, :3] - co64[g, :3]) / np.abs(co64[g, :3])))
