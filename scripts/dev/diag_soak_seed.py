"""What is behind one seed of the randomised rasteriser sweep (scripts/soak_raster.py): per tensor the elements beyond 1e-4 of
the inf-norm, where they are, how large, what the oracle's witness says there, and which near-tie mechanism moves them (each
FLIP_MARGINS family switched on alone).   gpurun -- 'python scripts/dev/diag_soak_seed.py 3779'"""
import sys

import numpy as np

sys.path.insert(0, "free-surgs_amd")
sys.path.insert(0, ".")
import tests.test_raster_gpu as TR  # noqa: E402
from oracle.fsgs_oracle import Oracle  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3779
o = Oracle(np.float32)
o.set_threads(8)
cam, xyz, col, op, s, r = TR.sweep_scene(seed)
H, W = cam["image_height"], cam["image_width"]
Cc, P = np.asarray(col).shape[1], len(xyz)
print("seed %d: %d x %d image, %d Gaussians, %d channels" % (seed, W, H, P, Cc))
dL = (np.random.default_rng(seed).uniform(-1, 1, (Cc, H, W)) / (Cc * H * W)).astype(np.float32)
img, dep, radii, g = TR._run_hip(cam, xyz, col, op, s, r, dL)
amp, (oi, od, orad, og, st) = o.flip_amplitudes(cam, xyz, col, op, s, r, dL)
print("num_rendered", st.num_rendered, " order ties found:", None if o.find_order_ties(st) is None else int((o.find_order_ties(st) != 0).sum()))
o._order_h = None
for name, a, b, am in (("image", img, oi, amp["image"]), ("depth", dep, od, amp["depth"])):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1.0
    err = np.abs(a - b)
    out = err > 1e-4 * scale
    print("%s: scale %.3g, %d of %d beyond 1e-4, max err %.3g of scale, all witnessed: %s" % (
        name, scale, out.sum(), a.size, err.max() / scale, bool((err[out] <= 1e-4 * scale + 2 * np.asarray(am)[out]).all())))
    if out.any():
        idx = np.argwhere(out)
        lo, hi = idx.min(axis=0), idx.max(axis=0)
        print("   bounding box of the outliers (last two axes = y, x):", lo.tolist(), hi.tolist())
        print("   err / scale quantiles of the outliers:", np.quantile(err[out] / scale, [0, 0.5, 0.9, 1.0]).round(6).tolist())
        print("   sign: %d above, %d below" % (((a - b)[out] > 0).sum(), ((a - b)[out] < 0).sum()))
# which pixels of the depth outliers share their last contributor / list neighbours: the Gaussians covering the outlier box
d = np.asarray(dep, np.float64) - np.asarray(od, np.float64)
yy, xx = np.nonzero(np.abs(d) > 1e-4 * (np.abs(od).max() + 1.0))
if len(yy):
    n_contrib_o = st.n_contrib().reshape(H, W)
    print("oracle n_contrib over the outliers: min %d max %d;  final_T there: min %.3g max %.3g" % (
        n_contrib_o[yy, xx].min(), n_contrib_o[yy, xx].max(), st.final_T().reshape(H, W)[yy, xx].min(), st.final_T().reshape(H, W)[yy, xx].max()))
    # oracle with the thresholds moved either way: does ITS depth move there by as much?
    for sign in (+1, -1):
        o.find_order_ties(st)
        o.set_thresholds(sign)
        i2, d2, r2, s2 = o.raster_forward(cam, xyz, col, op, s, r)
        o.set_thresholds(0)
        mv = np.abs(np.asarray(d2, np.float64) - np.asarray(od, np.float64))[yy, xx]
        print("   oracle's own depth with thresholds %+d: moves by %.3g .. %.3g of scale over the outliers (HIP differs by %.3g .. %.3g)" % (
            sign, mv.min() / (np.abs(od).max() + 1), mv.max() / (np.abs(od).max() + 1), np.abs(d[yy, xx]).min() / (np.abs(od).max() + 1),
            np.abs(d[yy, xx]).max() / (np.abs(od).max() + 1)))
    # the depth values of the Gaussians: any two within a few ulp?
    z = np.asarray(st.depth(), np.float64)
    vis = np.asarray(orad) > 0
    zs = np.sort(z[vis])
    rel = np.diff(zs) / np.maximum(np.abs(zs[:-1]), 1e-30)
    print("   visible Gaussians %d; closest depth pairs (relative gaps): %s" % (vis.sum(), np.sort(rel)[:5].tolist()))
    print("   FLIP_MARGINS:", o.FLIP_MARGINS)
    # who is closer to the truth?  the fp64 build of the oracle on the same inputs
    o64 = Oracle(np.float64)
    i64, d64, r64, s64 = o64.raster_forward(cam, xyz, col, op, s, r)
    sc_ = np.abs(od).max() + 1
    e_hip = np.abs(np.asarray(dep, np.float64) - d64)[yy, xx] / sc_
    e_o32 = np.abs(np.asarray(od, np.float64) - d64)[yy, xx] / sc_
    print("   distance to the fp64 oracle over the outliers: HIP median %.3g max %.3g;  fp32 oracle median %.3g max %.3g" % (
        np.median(e_hip), e_hip.max(), np.median(e_o32), e_o32.max()))
    print("   kappa (a c / det of the conic) of the Gaussians:", np.asarray(st.kappa(), np.float64).round(1).tolist())
    print("   radii:", np.asarray(orad).tolist())
