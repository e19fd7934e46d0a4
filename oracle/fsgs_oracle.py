"""ctypes front-end of the CPU oracle (oracle/raster_oracle.c, oracle/knn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- never by the product package.  Parity is
UNPINNED for the rasteriser (see the header of raster_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAXC = 8


def usable_cores():
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes expose
    256 logical CPUs under a 16-core quota; 256 OpenMP threads there run 5x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(np.ceil(float(quota) / float(period)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // pr)))
        except Exception:
            pass
    return n


def build(force=False):
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (oracle/Makefile)."""
    libs = [os.path.join(_HERE, n) for n in ("liboracle_f32.so", "liboracle_f64.so")]
    srcs = [os.path.join(_HERE, n) for n in ("raster_oracle.c", "knn_oracle.c", "Makefile")]
    stale = force or any(
        (not os.path.exists(l)) or any(os.path.getmtime(s) > os.path.getmtime(l) for s in srcs) for l in libs
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return libs


def _cfg_struct(real):
    class OracleCfg(C.Structure):
        _fields_ = [
            ("image_height", C.c_int),
            ("image_width", C.c_int),
            ("channels", C.c_int),
            ("reserved", C.c_int),
            ("tanfovx", real),
            ("tanfovy", real),
            ("scale_modifier", real),
            ("bg", real * MAXC),
            ("viewmatrix", real * 16),
            ("projmatrix", real * 16),
        ]

    return OracleCfg


class Oracle:
    """One precision of the oracle. dtype = np.float32 (parity) or np.float64 (finite differences)."""

    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        f64 = self.dtype == np.float64
        libs = build()
        self.lib = C.CDLL(libs[1] if f64 else libs[0])
        self.real = C.c_double if f64 else C.c_float
        self.Cfg = _cfg_struct(self.real)
        L = self.lib
        assert L.oracle_real_bytes() == self.dtype.itemsize
        rp = C.POINTER(self.real)
        ip = C.POINTER(C.c_int)
        L.oracle_raster_forward.restype = C.c_void_p
        L.oracle_raster_forward.argtypes = [C.POINTER(self.Cfg), C.c_int, rp, rp, rp, rp, rp, rp, rp, ip]
        L.oracle_raster_backward.restype = None
        L.oracle_raster_backward.argtypes = [C.POINTER(self.Cfg), C.c_void_p] + [rp] * 11
        L.oracle_raster_free.argtypes = [C.c_void_p]
        L.oracle_state_num_rendered.restype = C.c_int64
        L.oracle_state_num_rendered.argtypes = [C.c_void_p]
        for name, rt in (
            ("oracle_state_xy", rp),
            ("oracle_state_conic_op", rp),
            ("oracle_state_depth", rp),
            ("oracle_state_cov3D", rp),
            ("oracle_state_kappa", rp),
            ("oracle_state_final_T", rp),
            ("oracle_state_n_contrib", ip),
            ("oracle_state_tiles", ip),
            ("oracle_state_point_list", C.POINTER(C.c_uint32)),
            ("oracle_state_ranges", ip),
        ):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.oracle_knn_meandist2.restype = None
        L.oracle_knn_meandist2.argtypes = [C.c_int, rp, rp]
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_set_thresholds.argtypes = [C.c_double] * 9
        L.oracle_set_thresholds.restype = None
        L.oracle_reset_thresholds.restype = None
        L.oracle_set_depth_shift.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.oracle_set_depth_shift.restype = None
        self._order_h = None
        self._shift_buf = None
        self.last_state = None

    # -- helpers -------------------------------------------------------------
    def set_threads(self, n):
        self.lib.oracle_set_threads(int(n))

    def max_threads(self):
        return int(self.lib.oracle_max_threads())

    def _a(self, x, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def _p(self, a):
        return a.ctypes.data_as(C.POINTER(self.real))

    def make_cfg(self, cam, channels=3, bg=None):
        """cam: dict with image_height, image_width, tanfovx, tanfovy, viewmatrix[4,4], projmatrix[4,4]
        (both in the reference's transposed storage), optional scale_modifier, bg."""
        cfg = self.Cfg()
        cfg.image_height = int(cam["image_height"])
        cfg.image_width = int(cam["image_width"])
        cfg.channels = int(channels)
        cfg.tanfovx = float(cam["tanfovx"])
        cfg.tanfovy = float(cam["tanfovy"])
        cfg.scale_modifier = float(cam.get("scale_modifier", 1.0))
        if bg is None:
            bg = cam.get("bg", [1.0, 1.0, 1.0])
        bg = list(np.asarray(bg, dtype=np.float64).reshape(-1))
        if len(bg) < channels:  # same background for the extra (depth/sil/depth^2) channels
            bg = bg + [bg[-1]] * (channels - len(bg))
        for i in range(MAXC):
            cfg.bg[i] = float(bg[i]) if i < len(bg) else 0.0
        vm = np.asarray(cam["viewmatrix"], dtype=np.float64).reshape(16)
        pm = np.asarray(cam["projmatrix"], dtype=np.float64).reshape(16)
        for i in range(16):
            cfg.viewmatrix[i] = float(vm[i])
            cfg.projmatrix[i] = float(pm[i])
        return cfg

    # -- rasteriser ------------------------------------------------------------
    def raster_forward(self, cam, means3D, colors, opacities, scales, rotations, bg=None):
        means3D = self._a(means3D)
        P = means3D.shape[0]
        colors = self._a(colors)
        colors = colors.reshape(P, -1) if P > 0 else colors.reshape(0, colors.shape[-1] if colors.ndim > 1 else 3)
        Cc = colors.shape[1]
        cfg = self.make_cfg(cam, Cc, bg)
        opacities = self._a(opacities, (P,))
        scales = self._a(scales, (P, 3))
        rotations = self._a(rotations, (P, 4))
        H, W = cfg.image_height, cfg.image_width
        out_color = np.zeros((Cc, H, W), self.dtype)
        out_depth = np.zeros((H, W), self.dtype)
        radii = np.zeros((max(P, 1),), np.int32)
        st = self.lib.oracle_raster_forward(
            C.byref(cfg), P, self._p(means3D), self._p(colors), self._p(opacities), self._p(scales),
            self._p(rotations), self._p(out_color), self._p(out_depth), radii.ctypes.data_as(C.POINTER(C.c_int)),
        )
        state = OracleState(self, st, cfg, (means3D, colors, opacities, scales, rotations))
        self.last_state = state
        return out_color, out_depth, radii[:P], state

    def raster_backward(self, state, dL_dcolor):
        means3D, colors, opacities, scales, rotations = state.inputs
        P, Cc = means3D.shape[0], colors.shape[1]
        dL = self._a(dL_dcolor, (Cc, state.cfg.image_height, state.cfg.image_width))
        n = max(P, 1)
        out = {
            "means2D": np.zeros((n, 3), self.dtype),
            "colors": np.zeros((n, Cc), self.dtype),
            "opacities": np.zeros((n,), self.dtype),
            "means3D": np.zeros((n, 3), self.dtype),
            "scales": np.zeros((n, 3), self.dtype),
            "rotations": np.zeros((n, 4), self.dtype),
        }
        self.lib.oracle_raster_backward(
            C.byref(state.cfg), state.ptr, self._p(means3D), self._p(colors), self._p(scales), self._p(rotations),
            self._p(dL), self._p(out["means2D"]), self._p(out["colors"]), self._p(out["opacities"]),
            self._p(out["means3D"]), self._p(out["scales"]), self._p(out["rotations"]),
        )
        return {k: v[:P] for k, v in out.items()}

    # -- flip attribution ------------------------------------------------------------
    # How far the decision thresholds are moved (relative unless noted).  The HIP path and this restatement evaluate
    # the same comparisons in fp32 on differently rounded operands:
    #  * the projected centre is good to an ulp (6e-5 px between 512 and 1024 px), and alpha = o exp(-d' S^-1 d / 2)
    #    moves by |S^-1 d| per pixel of centre shift -- ~5.5 / px three sigma out on the smallest footprint the +0.3
    #    dilation allows (sigma = 0.55 px): ~3e-4 relative -> alpha_min;
    #  * the exponent is a sum of three products that cancel on elongated (correlated) footprints; each carries its
    #    own rounding (~2 ulp, evaluated in a different order on either side: plain vs pre-scaled fma chain), so the
    #    exponent is good to ~5e-7 * (|A dx^2| / 2 + |C dy^2| / 2 + |B dx dy|) absolute, which is the same relative
    #    amount in alpha -> alpha_cond (per evaluated pair, on top of alpha_min);
    #  * T inherits both through its factors (1 - alpha); 3 sigma is good to a few 1e-7; tile rects truncate
    #    (centre -+ radius) / 16.
    # All far inside "alpha within 1e-5 of 1/255" (2.5e-3 relative) / "T within 1e-7 of 1e-4" (1e-3 relative).
    #  * the depth ORDER inside a tile: view-space depths computed by different glue differ in the last bit or two,
    #    so list neighbours within depth_ulps ulp may be sorted either way (find_order_ties).
    #  * the conic = (c, -b, a) / (a c - b^2) of a needle-shaped footprint: det cancels, kappa = a c / det (kept per
    #    Gaussian by the oracle) and ANY fp32 evaluation of the conic is only good to ~6 kappa eps -- the exponent inherits
    #    that, so the alpha_cond term is multiplied by (1 + alpha_kappa * kappa).  Found by the 3000-seed soak of round 3
    #    (seed 1459: a 16 x 0.03 needle, kappa = 3e3, the oracle's and the HIP path's conic 1.1e-3 above / 0.8e-3 below
    #    the float64 value, alpha 6.6e-3 apart across the 1/255 line).  kappa ~ 1 for anything round: no effect there.
    FLIP_MARGINS = dict(alpha_min=4e-4, alpha_cond=5e-7, alpha_max_abs=4e-4, T_min=5e-4, power_abs=1e-5, radius=1e-6,
                        near_plane_abs=1e-6, rect_abs=2e-4, depth_ulps=3.0, alpha_kappa=2.0)

    def find_order_ties(self, state=None):
        """Per-Gaussian signs h in {-1, 0, +1} such that shifting every sort key by h * (a few ulp), one way and then
        the other, swaps every pair of NEIGHBOURS in a tile's depth-sorted list whose depths are within
        FLIP_MARGINS['depth_ulps'] ulp: h = +1 / -1 on the two members of such a pair (pairs are rare and almost always
        disjoint; a Gaussian in two pairs keeps its first sign).  Stored for set_thresholds(+-1); None = no ties."""
        st = state or self.last_state
        self._order_h = None
        if st is None:
            return None
        pl = st.point_list().astype(np.int64)
        if len(pl) < 2:
            return None
        d = np.asarray(st.depth(), np.float64)[pl]
        rg = st.ranges()
        tile_of = np.zeros(len(pl), np.int64)  # which tile a list position belongs to
        starts = rg[:, 0][rg[:, 1] > rg[:, 0]]
        tile_of[starts] = 1
        tile_of = np.cumsum(tile_of)
        near = (tile_of[1:] == tile_of[:-1]) & (np.abs(d[1:] - d[:-1]) <= self.FLIP_MARGINS["depth_ulps"] * 1.2e-7 * np.abs(d[:-1])) \
            & (pl[1:] != pl[:-1])
        idx = np.nonzero(near)[0]
        if len(idx) == 0:
            return None
        h = np.zeros(st.P, np.float64)
        for k in idx:
            i, j = pl[k], pl[k + 1]  # i sorts in front of j: moving i back and j forward swaps them
            if h[i] == 0 and h[j] == 0:
                h[i], h[j] = +1.0, -1.0
            elif h[i] == 0:
                h[i] = -h[j]
            elif h[j] == 0:
                h[j] = -h[i]
        self._order_h = h
        return h

    def set_thresholds(self, sign=0):
        """sign = 0: the published constants; +1 / -1: every decision moved by FLIP_MARGINS towards 'contributes more'
        / 'contributes less'.  Process-global in the C library: always restore with sign = 0."""
        m = self.FLIP_MARGINS
        s = float(sign)
        if sign == 0 or self._order_h is None:
            self.lib.oracle_set_depth_shift(None, 0)
            self._shift_buf = None
        else:  # swap the near-tie neighbours of the depth order (find_order_ties) one way, then the other
            self._shift_buf = np.ascontiguousarray(self._order_h * (s * m["depth_ulps"] * 1.2e-7), np.float64)
            self.lib.oracle_set_depth_shift(self._shift_buf.ctypes.data_as(C.POINTER(C.c_double)), len(self._shift_buf))
        if sign == 0:
            self.lib.oracle_reset_thresholds()
            return
        self.lib.oracle_set_thresholds(
            (1.0 / 255.0) * (1.0 - s * m["alpha_min"]), 0.99 + s * m["alpha_max_abs"], 1e-4 * (1.0 - s * m["T_min"]),
            s * m["power_abs"], 1.0 + s * m["radius"], 0.2 - s * m["near_plane_abs"], s * m["rect_abs"],
            s * m["alpha_cond"], m["alpha_kappa"])

    # pixel classes of the gradient decomposition below: the backward is linear in dL/dpixel, so the gradient is the
    # sum of the gradients of the four 2x2-interleaved pixel classes
    FLIP_CLASSES = 4

    @staticmethod
    def pixel_class_masks(H, W):
        yy, xx = np.mgrid[0:H, 0:W]
        cls = (yy % 2) * 2 + (xx % 2)
        return [(cls == c) for c in range(4)]

    _f64 = None

    def flip_amplitudes(self, cam, means3D, colors, opacities, scales, rotations, dL_dcolor, roundoff=True):
        """How much every output element can move when a near-tie decision flips: run the rasteriser (forward +
        backward) with the nominal thresholds, then with every threshold moved a hair one way, then the other way
        (FLIP_MARGINS).  amp[name] = elementwise max |difference| between any two of the three runs: zero wherever no
        decision is within rounding distance of its threshold -- there two correct fp32 implementations must agree to
        the parity tolerance with no exceptions; elsewhere they may differ by about amp.  -> (amp, nominal run) with
        amp: image [C,H,W], depth [H,W], final_T [H*W], the six gradients [P,k] (float64) and boolean masks
        radii [P], n_contrib [H*W]; nominal run = (image, depth, radii, grads, state).
        roundoff=True adds, element by element, this fp32 restatement's OWN distance from its fp64 build on the same
        inputs: a per-Gaussian gradient is a sum over thousands of pixels with cancellation (a needle covering the
        whole image: 2e-4 of the norm between the f32 and f64 oracles), and one correct fp32 implementation cannot be
        asked to sit closer to another than that one sits to the truth.  That distance is ONE sample of the summation
        noise and the other implementation (atomics in arrival order) contributes its own, so it enters twice."""
        # A Gaussian's gradient is a SUM over its pixels; with the thresholds moved, several of its pixels flip at
        # once and their signed contributions partly cancel in the sum, which would understate what ONE flip (all the
        # other implementation may differ by) can do.  So the backward runs per pixel class (it is linear in
        # dL/dpixel) and the amplitudes of the classes are added: flips only meet inside a class.
        dL = np.asarray(dL_dcolor)
        H, W = dL.shape[-2:]
        masks = self.pixel_class_masks(H, W)
        runs = []
        try:
            for sign in (0, +1, -1):
                self.set_thresholds(sign)
                img, dep, radii, st = self.raster_forward(cam, means3D, colors, opacities, scales, rotations)
                if sign == 0:
                    self.find_order_ties(st)
                gs = [self.raster_backward(st, dL * m) for m in masks]
                runs.append((img, dep, radii, gs, st, st.final_T().copy(), st.n_contrib().copy()))
        finally:
            self._order_h = None
            self.set_thresholds(0)
        nom = list(runs[0])
        P = len(nom[2])

        def amp(get):
            a = [np.asarray(get(r), np.float64) for r in runs]
            return np.maximum(np.maximum(np.abs(a[1] - a[0]), np.abs(a[2] - a[0])), np.abs(a[1] - a[2]))

        out = {
            "image": amp(lambda r: r[0]), "depth": amp(lambda r: r[1]), "final_T": amp(lambda r: r[5]),
            "radii": (runs[1][2] != nom[2]) | (runs[2][2] != nom[2]),
            "n_contrib": (runs[1][6] != nom[6]) | (runs[2][6] != nom[6]),
        }
        for k in ("means3D", "means2D", "colors", "opacities", "scales", "rotations"):
            out[k] = sum(amp(lambda r, k=k, c=c: r[3][c][k].reshape(P, -1)) for c in range(len(masks)))
        nom[3] = self.raster_backward(nom[4], dL)  # the nominal gradient itself: one backward over all pixels
        if roundoff and self.dtype != np.float64:
            if Oracle._f64 is None:
                Oracle._f64 = Oracle(np.float64)
            o64 = Oracle._f64
            i64, d64, r64, s64 = o64.raster_forward(cam, means3D, colors, opacities, scales, rotations)
            g64 = o64.raster_backward(s64, dL_dcolor)
            out["image"] += 2 * np.abs(i64 - nom[0])
            out["depth"] += 2 * np.abs(d64 - nom[1])
            out["final_T"] += 2 * np.abs(s64.final_T() - nom[5])
            # Gradients: every component of ONE Gaussian is a sum over the same pixels of terms of comparable conditioning
            # (a needle covering the image, a footprint cut by the near plane: sums that cancel to a small remainder),
            # and where on that Gaussian's row the fp32 build happens to land close to the fp64 one is chance.  So the
            # round-off LEVEL of a Gaussian -- its worst |f64 - f32| over all six tensors, each relative to its tensor's
            # norm (floored like the comparison itself) -- is granted to all of its components, not element by element
            # (3000-seed soak of round 3: two elements 1.3e-4 / 1.6e-4 off on Gaussians whose other components were
            # 2e-4 / 6e-4 off IN THE ORACLE).  ~1e-6 for any well-conditioned Gaussian: no effect there.
            names = ("means3D", "means2D", "colors", "opacities", "scales", "rotations")
            rd = {k: np.abs(g64[k].reshape(P, -1) - np.asarray(nom[3][k], np.float64).reshape(P, -1)) for k in names}
            norm = {k: float(np.max(np.abs(nom[3][k]))) if P else 0.0 for k in names}
            floor = 1e-3 * max(norm.values()) if P else 0.0
            level = np.zeros(P)
            for k in names:
                if P:
                    level = np.maximum(level, rd[k].max(axis=1) / (norm[k] + floor + 1e-300))
            # ... and the level has a floor set by the conditioning of the Gaussian itself, relative to its OWN gradient row:
            #  * the conic of a needle-shaped footprint is only good to ~6 kappa eps in ANY fp32 evaluation (kappa = a c / det,
            #    OracleState.kappa; FLIP_MARGINS['alpha_kappa'] is the same statement for the alpha threshold), and
            #    dL/dcov2D = -conic dL/dconic conic inherits that;
            #  * dL/dscale and dL/drotation come from dL/dSigma through Sigma = R S^2 R^T: terms of size s_max^2 |dL/dSigma| cancel
            #    down to what the small axes leave, kappa3 = (s_max / s_min)^2 of rounding.
            # The oracle's fp32-vs-fp64 distance is ONE draw of that error and can come out small by luck: seed 2148 of the
            # soak (a 0.58 x 0.019 x 0.009 needle, kappa3 = 4.1e3, radius 62 px in a 27 x 52 image) sits 7e-5 of the norm from
            # its fp64 value in the oracle and 5.5e-4 in the HIP path on one rotation component, 8 % beyond the allowance
            # that one draw gave.  Both factors are ~1 for anything round (4e-7 / 1e-7 of the row): no effect there.
            eps32 = float(np.finfo(np.float32).eps)
            kap = np.asarray(nom[4].kappa(), np.float64)
            sc = np.abs(np.asarray(scales, np.float64).reshape(P, -1)) if P else np.zeros((0, 3))
            kap3 = (sc.max(axis=1) / np.maximum(sc.min(axis=1), 1e-30)) ** 2 if P else np.zeros(0)
            for k in names:
                own = np.abs(g64[k].reshape(P, -1)).max(axis=1) if P else np.zeros(0)
                cond = 6.0 * np.maximum(kap, 0.0) * eps32 + (kap3 * eps32 if k in ("scales", "rotations") else 0.0)
                out[k] += 2 * np.maximum(np.maximum(rd[k], (level * (norm[k] + floor))[:, None]), (cond * own)[:, None])
        return out, nom[:5]

    # -- KNN ---------------------------------------------------------------------
    def knn_meandist2(self, pts):
        pts = self._a(pts, (-1, 3))
        out = np.zeros((max(pts.shape[0], 1),), self.dtype)
        with np.errstate(over="ignore"):
            self.lib.oracle_knn_meandist2(pts.shape[0], self._p(pts), self._p(out))
        return out[: pts.shape[0]]


class OracleState:
    def __init__(self, oracle, ptr, cfg, inputs):
        self.oracle, self.ptr, self.cfg, self.inputs = oracle, ptr, cfg, inputs

    @property
    def num_rendered(self):
        return int(self.oracle.lib.oracle_state_num_rendered(self.ptr))

    def _arr(self, fn, shape, dtype):
        p = getattr(self.oracle.lib, fn)(self.ptr)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype)
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(shape).astype(dtype, copy=True)

    @property
    def P(self):
        return self.inputs[0].shape[0]

    def xy(self):
        return self._arr("oracle_state_xy", (self.P, 2), self.oracle.dtype)

    def conic_opacity(self):
        return self._arr("oracle_state_conic_op", (self.P, 4), self.oracle.dtype)

    def depth(self):
        return self._arr("oracle_state_depth", (self.P,), self.oracle.dtype)

    def cov3D(self):
        """[P,6] = (xx, xy, xz, yy, yz, zz) of Sigma; zero rows for Gaussians behind the near plane."""
        return self._arr("oracle_state_cov3D", (self.P, 6), self.oracle.dtype)

    def kappa(self):
        """[P] a c / det of the dilated 2D covariance (1 for a round footprint, thousands for a needle); 0 where culled"""
        return self._arr("oracle_state_kappa", (self.P,), self.oracle.dtype)

    def tiles_touched(self):
        return self._arr("oracle_state_tiles", (self.P,), np.int32)

    def final_T(self):
        return self._arr("oracle_state_final_T", (self.cfg.image_height, self.cfg.image_width), self.oracle.dtype)

    def n_contrib(self):
        return self._arr("oracle_state_n_contrib", (self.cfg.image_height, self.cfg.image_width), np.int32)

    def ranges(self):
        gx = (self.cfg.image_width + 15) // 16
        gy = (self.cfg.image_height + 15) // 16
        return self._arr("oracle_state_ranges", (gx * gy, 2), np.int32)

    def point_list(self):
        return self._arr("oracle_state_point_list", (self.num_rendered,), np.uint32)

    def __del__(self):
        try:
            if self.ptr:
                self.oracle.lib.oracle_raster_free(self.ptr)
                self.ptr = None
        except Exception:
            pass
