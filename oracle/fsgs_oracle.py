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
