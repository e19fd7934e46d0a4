import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
import numpy as np, torch
from oracle.fsgs_oracle import Oracle
from fsgs_amd.model import PARAM_NAMES
from fsgs_amd.render import render, render_two_pass
from tests import ref_cpu
import tests.test_render_gpu as T
deg, gi = 3, 2588
W, H, P = 320, 256, 5000
pc, poses = T._setup(W, H, P, deg, seed=deg)
g = torch.Generator(device="cpu").manual_seed(1)
wi = (torch.rand(3, H, W, generator=g) - 0.5).to("cuda") / (H * W)
wd = (torch.rand(H, W, generator=g) - 0.5).to("cuda") / (H * W)
ws = (torch.rand(H, W, generator=g) - 0.5).to("cuda") / (H * W)
o = Oracle(np.float32)
ref_o, ref_g, amp_o, amp_g = ref_cpu.reference_render_with_amplitudes(o, pc, poses, 1, True, False, wi, wd, ws)
f_o, f_g = T._run(render, pc, poses, True, False, wi, wd, ws)
t_o, t_g = T._run(render_two_pass, pc, poses, True, False, wi, wd, ws)
for k in PARAM_NAMES + ("viewspace",):
    sc = np.abs(ref_g[k]).max()
    print(k, "norm", sc, "\n  ref  ", ref_g[k][gi].reshape(-1)[:6], "\n  fused", f_g[k][gi].reshape(-1)[:6], "\n  2pass", t_g[k][gi].reshape(-1)[:6])
print("radii", ref_o["radii"][gi], f_o["radii"][gi], t_o["radii"][gi])
print("xyz", pc.params["_xyz"][gi], "scal", pc.params["_scaling"][gi].exp(), "op", torch.sigmoid(pc.params["_opacity"][gi]))
# all elements: fused vs 2pass worst
for k in PARAM_NAMES:
    sc = np.abs(ref_g[k]).max()
    print(k, "max |fused-ref|/norm", np.abs(f_g[k] - ref_g[k]).max() / sc, "|2pass-ref|", np.abs(t_g[k] - ref_g[k]).max() / sc, "|fused-2pass|", np.abs(f_g[k] - t_g[k]).max() / sc)

# ---- where do the rasterisers decide differently around Gaussian gi? (3-channel RGB pass through the drop-in) ----
from fsgs_amd import rasterizer
from fsgs_amd.render import rendervars
from fsgs_amd.pose import transform_to_frame
with torch.no_grad():
    w2c = poses.get_pose(1)
    tr = transform_to_frame(pc.get_xyz, w2c, False, False)
    rv, dv = rendervars(pc, tr, torch.zeros_like(tr), poses.cam_center)
cfg = rasterizer.make_cfg(pc.cam, 3)
c = lambda t: t.detach().contiguous()
img, depth, radii, st = rasterizer.raster_forward(cfg, c(rv["means3D"]), c(rv["colors_precomp"]), c(rv["opacities"]).reshape(-1), c(rv["scales"]), c(rv["rotations"]))
v = {k: t.cpu().numpy() for k, t in rasterizer.state_views(st).items()}
n = lambda t: t.detach().cpu().numpy()
cam = ref_cpu.cam_from_settings(pc.cam)
oi, od, orad, ost = o.raster_forward(cam, n(rv["means3D"]), n(rv["colors_precomp"]), n(rv["opacities"]).reshape(-1), n(rv["scales"]), n(rv["rotations"]))
oxy, oco = ost.xy(), ost.conic_opacity()
print("geometry of gi: oracle xy", oxy[gi], "hip", v["xy"][gi], "conic", oco[gi], v["conic_opacity"][gi])
gx = (W + 15) // 16
x0, y0 = int(oxy[gi, 0]) - 17, int(oxy[gi, 1]) - 17
ndiff = 0
for y in range(max(0, y0), min(H, y0 + 35)):
    for x in range(max(0, x0), min(W, x0 + 35)):
        tile = (y // 16) * gx + x // 16
        r0, r1 = ost.ranges()[tile]
        ids = ost.point_list()[r0:r1]
        hl = set(v["point_list"][v["ranges"][tile, 0]:v["ranges"][tile, 1]].tolist())
        To = Th = 1.0; done_o = done_h = False
        for k, g in enumerate(ids):
            def ev(xy, co):
                dx = np.float32(xy[0]) - np.float32(x); dy = np.float32(xy[1]) - np.float32(y)
                p = np.float32(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
                return p, (co[3] * np.exp(np.float32(p)) if p <= 0 else 0.0)
            po, ao = ev(oxy[g], oco[g]); ph, ah = ev(v["xy"][g], v["conic_opacity"][g])
            in_h = g in hl
            co_ = ao >= 1 / 255 and not done_o; ch_ = ah >= 1 / 255 and in_h and not done_h
            if co_:
                if To * (1 - min(0.99, ao)) < 1e-4: done_o = True; co_ = False
            if ch_:
                if Th * (1 - min(0.99, ah)) < 1e-4: done_h = True; ch_ = False
            if co_ != ch_ and ndiff < 12:
                ndiff += 1
                print("pixel", y, x, "k", k, "g", g, "oracle alpha*255 %.6f hip-geom alpha*255 %.6f in_hip_list %s To %.3e Th %.3e done %s %s" % (ao * 255, ah * 255, in_h, To, Th, done_o, done_h))
            if co_: To *= 1 - min(0.99, ao)
            if ch_: Th *= 1 - min(0.99, ah)
print("decision differences found:", ndiff)

# ---- image-level differences in the footprint of gi (both passes) ----
cx, cy, rad = oxy[gi, 0], oxy[gi, 1], orad[gi]
x0, x1, y0, y1 = int(max(0, cx - rad - 1)), int(min(W, cx + rad + 2)), int(max(0, cy - rad - 1)), int(min(H, cy + rad + 2))
for name, a, b in (("render", f_o["render"], ref_o["render"]), ("depth", f_o["render_dep"][None], ref_o["render_dep"][None]), ("sil", f_o["sil"][None], ref_o["sil"][None])):
    d = np.abs(a - b)[:, y0:y1, x0:x1].max(axis=0)
    ys, xs = np.nonzero(d > 1e-6)
    print(name, "px with diff > 1e-6 in footprint:", [(int(y + y0), int(x + x0), float(d[y, x])) for y, x in zip(ys, xs)][:8])
# the depth/sil pass through the drop-in, decisions per pixel of the footprint: T termination
img2, depth2, radii2, st2 = rasterizer.raster_forward(cfg, c(dv["means3D"]), c(dv["colors_precomp"]), c(dv["opacities"]).reshape(-1), c(dv["scales"]), c(dv["rotations"]))
v2 = {k: t.cpu().numpy() for k, t in rasterizer.state_views(st2).items()}
on = ost.n_contrib().reshape(H, W); hn = v["n_contrib"].reshape(H, W)
def last_id(ranges, plist, n):
    yy, xx = np.mgrid[0:H, 0:W]; tile = (yy // 16) * gx + xx // 16
    pos = ranges[tile, 0].astype(np.int64) + n - 1
    return np.where(n > 0, plist[np.clip(pos, 0, len(plist) - 1)].astype(np.int64), -1)
a = last_id(v["ranges"], v["point_list"], hn.astype(np.int64)); b = last_id(ost.ranges(), ost.point_list(), on.astype(np.int64))
ys, xs = np.nonzero(a[y0:y1, x0:x1] != b[y0:y1, x0:x1])
print("last contributor differs at", [(int(y + y0), int(x + x0)) for y, x in zip(ys, xs)][:8])
fT = np.abs(v["final_T"].reshape(H, W) - ost.final_T().reshape(H, W))[y0:y1, x0:x1]
print("final_T max diff in footprint", fT.max(), "at", np.unravel_index(fT.argmax(), fT.shape), "values", ost.final_T().reshape(H, W)[y0:y1, x0:x1].min())

# ---- bisect: which pixel's dL produces a different dL/dcolour for Gaussian gi (HIP raster bwd vs oracle)? ----
m3, cc, oo, ss, rr = [c(rv[k]) for k in ("means3D", "colors_precomp", "opacities", "scales", "rotations")]
def hip_dcol(dLm):
    g = rasterizer.raster_backward(st, m3, cc, ss, rr, radii, torch.tensor(dLm, device="cuda"))
    return g[1][gi].cpu().numpy()
def ora_dcol(dLm):
    return o.raster_backward(ost, dLm)["colors"][gi]
bad_rows = []
for y in range(y0, y1):
    dLm = np.zeros((3, H, W), np.float32); dLm[:, y, x0:x1] = 1.0
    a, b = hip_dcol(dLm), ora_dcol(dLm)
    if abs(a[0] - b[0]) > 2e-6 * max(1.0, abs(b[0])): bad_rows.append((y, a[0], b[0]))
print("rows with a different sum alpha*T for gi:", bad_rows[:10])
for (y, _, _) in bad_rows[:3]:
    for x in range(x0, x1):
        dLm = np.zeros((3, H, W), np.float32); dLm[:, y, x] = 1.0
        a, b = hip_dcol(dLm), ora_dcol(dLm)
        if abs(a[0] - b[0]) > 1e-6:
            tile = (y // 16) * gx + x // 16
            r0, r1 = ost.ranges()[tile]; ids = ost.point_list()[r0:r1]
            pos_o = int(np.nonzero(ids == gi)[0][0]) if gi in ids else -1
            hl = v["point_list"][v["ranges"][tile, 0]:v["ranges"][tile, 1]]
            pos_h = int(np.nonzero(hl == gi)[0][0]) if gi in hl else -1
            print("  pixel", y, x, "hip alpha*T", a[0], "oracle", b[0], "pos in oracle list", pos_o, "of", len(ids), "n_contrib", on[y, x], "| pos in hip list", pos_h, "of", len(hl), "n_contrib", hn[y, x], "final_T", ost.final_T().reshape(H, W)[y, x])

# ---- stage by stage for gi: CPU glue vs GPU glue, oracle on either ----
cpc, cps = ref_cpu.cpu_cloud(pc), ref_cpu.cpu_poses(poses)
with torch.no_grad():
    w2c_c = cps.get_pose(1)
    tr_c = transform_to_frame(cpc.get_xyz, w2c_c, False, False)
    rv_c, dv_c = rendervars(cpc, tr_c, torch.zeros_like(tr_c), cps.cam_center)
print("w2c diff", (w2c_c - w2c.cpu()).abs().max().item())
for k in ("means3D", "colors_precomp", "opacities", "scales", "rotations"):
    d = (rv_c[k] - rv[k].cpu()).abs()
    print(k, "max diff over cloud", d.max().item(), "at", int(d.reshape(P, -1).max(1)[0].argmax()), "gi:", d[gi].reshape(-1).tolist()[:4])
dLw = np.stack([(wi.cpu().numpy())[i] for i in range(3)]).astype(np.float32)
oc = o.raster_forward(cam, n(rv_c["means3D"]), n(rv_c["colors_precomp"]), n(rv_c["opacities"]).reshape(-1), n(rv_c["scales"]), n(rv_c["rotations"]))
gc_ = o.raster_backward(oc[3], dLw)
gg_ = o.raster_backward(ost, dLw)
gh = rasterizer.raster_backward(st, m3, cc, ss, rr, radii, torch.tensor(dLw, device="cuda"))
print("RGB-pass raster grads of gi with the test's wi:  dcolors  oracle(CPU glue)", gc_["colors"][gi], " oracle(GPU glue)", gg_["colors"][gi], " HIP", gh[1][gi].cpu().numpy())
print("   dopac", gc_["opacities"][gi], gg_["opacities"][gi], gh[2][gi].cpu().numpy(), " dmeans3D", gc_["means3D"][gi], gg_["means3D"][gi], gh[3][gi].cpu().numpy())
