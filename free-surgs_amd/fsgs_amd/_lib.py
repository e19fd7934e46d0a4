"""ctypes binding of libfsgs_hip.so (C ABI declared in include/fsgs.h).

There is NO fallback: if the HIP library is missing or a call fails, the op raises.
PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# FSGS_LIB_PATH: an experiment build (free-surgs_amd/build.py, FSGS_LIB_TAG) instead of the product library -- A/B runs only
LIB_PATH = os.environ.get("FSGS_LIB_PATH") or os.path.join(_HERE, "lib", "libfsgs_hip.so")

FSGS_OK = 0
FSGS_ERR_INVALID = -1
FSGS_ERR_CAPACITY = -2
FSGS_FLAG_XCD_BANDED_ORDER, FSGS_FLAG_DEPTH_GRAD_ONLY, FSGS_FLAG_SCRATCH_ZEROED, FSGS_FLAG_RGB_DEPTH_ONLY = 1, 2, 4, 8
FSGS_FLAG_SCRATCH_SELF_CLEAN = 16
FSGS_FLAG_BLEND_ONE_WAVE, FSGS_FLAG_BLEND_QUAD_WAVES = 32, 64
FSGS_FLAG_DETERMINISTIC = 128
FSGS_ERR_HIP = -3
FSGS_ERR_STATE = -4
MAX_CHANNELS = 8

_ERR_NAMES = {
    FSGS_ERR_INVALID: "FSGS_ERR_INVALID (bad argument)",
    FSGS_ERR_CAPACITY: "FSGS_ERR_CAPACITY (buffer too small)",
    FSGS_ERR_HIP: "FSGS_ERR_HIP (HIP runtime / launch failure)",
    FSGS_ERR_STATE: "FSGS_ERR_STATE (state buffer does not match)",
}


class FsgsRasterCfg(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("channels", C.c_int32),
        ("flags", C.c_int32),
        ("tanfovx", C.c_float),
        ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float),
        ("reserved0", C.c_float),
        ("bg", C.c_float * MAX_CHANNELS),
        ("viewmatrix", C.c_float * 16),
        ("projmatrix", C.c_float * 16),
    ]


class FsgsRenderArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation",
                                          "w2c", "cam_center")] + [("active_sh_degree", C.c_int32),
                                                                   ("max_sh_degree", C.c_int32)]


class FsgsRenderGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation",
                                          "means2D", "w2c")]


class FsgsFusedAdam(C.Structure):
    _fields_ = [("exp_avg", C.c_void_p * 6), ("exp_avg_sq", C.c_void_p * 6), ("lr", C.c_float * 6),
                ("step", C.c_int32 * 6), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("next_colors", C.c_void_p)]


class FsgsStepTail(C.Structure):
    _fields_ = [("max_radii2D", C.c_void_p), ("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p),
                ("loss_terms", C.c_void_p), ("loss_weights", C.c_void_p), ("n_terms", C.c_int32),
                ("loss_total", C.c_void_p)]


class FsgsDensifyGroup(C.Structure):
    _fields_ = [("in_param", C.c_void_p), ("in_exp_avg", C.c_void_p), ("in_exp_avg_sq", C.c_void_p),
                ("out_param", C.c_void_p), ("out_exp_avg", C.c_void_p), ("out_exp_avg_sq", C.c_void_p),
                ("row", C.c_int32), ("role", C.c_int32)]


class FsgsAdamGroup(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("step", C.c_int32)]


class FsgsError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = "%s failed: %s" % (where, _ERR_NAMES.get(code, "error %d" % code))
        if detail:
            msg += " -- " + detail
        super().__init__(msg)


_lock = threading.Lock()
_lib = None

# every exported symbol of include/fsgs.h: (restype, argtypes)
_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_PROTOTYPES = {
    "fsgs_version": (C.c_char_p, []),
    "fsgs_last_error": (C.c_char_p, []),
    "fsgs_selftest_transpose_reduce": (_i, [_vp, _vp, _vp]),
    "fsgs_selftest_transpose_reduce_n": (_i, [_vp, _vp, _i, _vp]),
    "fsgs_selftest_splat_alpha": (_i, [_i, _vp, _vp, _vp]),
    "fsgs_profile_enable": (_i, [C.c_uint64]),
    "fsgs_profile_stride": (_i, [C.c_int]),
    "fsgs_profile_count": (_i, []),
    "fsgs_profile_name": (C.c_char_p, [_i]),
    "fsgs_profile_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "fsgs_blend_waves_per_tile": (_i, [_i, _i, _i, _i, _i]),
    "fsgs_deterministic_scratch_bytes": (_sz, [_i, _i64]),
    "fsgs_raster_sizes": (_i, [_i, _i, _i, _i64, C.POINTER(_sz), C.POINTER(_sz)]),
    "fsgs_raster_state_layout": (_i, [_i, _i, _i, _i64, C.POINTER(_sz)]),
    "fsgs_raster_forward": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i64,
         C.POINTER(_i64), _vp],
    ),
    "fsgs_raster_backward": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
         _vp, _vp, _sz, _vp],
    ),
    "fsgs_render_sizes": (_i, [_i, _i, _i, _i64, C.POINTER(_sz), C.POINTER(_sz)]),
    "fsgs_render_state_layout": (_i, [_i, _i, _i, _i64, C.POINTER(_sz)]),
    "fsgs_render_forward": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, C.POINTER(FsgsRenderArgs), _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i64,
         C.POINTER(_i64), _vp],
    ),
    "fsgs_render_forward_reuse_colors": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, C.POINTER(FsgsRenderArgs), _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i64,
         C.POINTER(_i64), _vp, _sz, _i64, _vp],
    ),
    "fsgs_render_forward_cached_colors": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, C.POINTER(FsgsRenderArgs), _vp, _vp, _vp, _vp, _sz, _vp, _sz, _i64,
         C.POINTER(_i64), _vp, _vp],
    ),
    "fsgs_render_backward": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, C.POINTER(FsgsRenderArgs), _vp, _vp, _sz, _i64, _i64, _vp, _vp, _i, _i, _i,
         C.POINTER(FsgsRenderGrads), _vp, _sz, _vp],
    ),
    "fsgs_render_backward_compact": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, C.POINTER(FsgsRenderArgs), _vp, _vp, _sz, _i64, _i64, _vp, _vp, _vp, _vp,
         C.POINTER(FsgsStepTail), _vp, _sz, _vp],
    ),
    "fsgs_render_backward_compact_rows": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, C.POINTER(FsgsRenderArgs), _vp, _vp, _sz, _i64, _i64, _vp, _vp, _vp, _vp,
         C.POINTER(FsgsStepTail), _vp, _sz, _i, _i, _i, _vp],
    ),
    "fsgs_adam_step_compact": (_i, [_i, C.POINTER(FsgsRenderArgs), _vp, C.POINTER(FsgsFusedAdam), _vp]),
    "fsgs_adam_step_compact_sum": (_i, [_i, C.POINTER(FsgsRenderArgs), _vp, _vp, C.POINTER(FsgsFusedAdam), _vp]),
    "fsgs_render_backward_adam": (
        _i,
        [C.POINTER(FsgsRasterCfg), _i, C.POINTER(FsgsRenderArgs), _vp, _vp, _sz, _i64, _i64, _vp, _vp,
         C.POINTER(FsgsFusedAdam), _vp, C.POINTER(FsgsStepTail), _vp, _sz, _vp],
    ),
    "fsgs_knn_meandist2": (_i, [_i, _vp, _vp, _vp, C.POINTER(_sz), _vp]),
    "fsgs_photometric_scratch_bytes": (_sz, [_i, _i, _i]),
    "fsgs_photometric_loss_forward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp]),
    "fsgs_photometric_loss_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp]),
    "fsgs_photometric_loss_forward_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_view_losses_forward_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp,
                                               _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_pearson_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "fsgs_pearson_forward": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_flow_pose_loss_forward": (_i, [_i64, _vp, _vp, _vp, C.POINTER(C.c_float), _vp, _i, _i, C.c_float, _vp, _vp,
                                         _vp]),
    "fsgs_flow_pose_loss_backward": (_i, [_i64, _vp, _vp, _vp, C.POINTER(C.c_float), _vp, _i, _i, C.c_float, _vp, _vp,
                                          _vp, _vp]),
    "fsgs_flow_scratch_bytes": (_sz, [_i64]),
    "fsgs_flow_pose_loss_fused": (_i, [_i64, _vp, _vp, _vp, C.POINTER(C.c_float), _vp, _i, _i, C.c_float, C.c_float,
                                       C.c_float, _vp, _vp, _vp, _vp]),
    "fsgs_flow_targets_keys": (_i, [_i, _i, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp, _vp, _vp]),
    "fsgs_flow_targets_flag": (_i, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_flow_targets_gather": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_sampson_scratch_bytes": (_sz, [_i, _i]),
    "fsgs_sampson_rigid_mask": (_i, [_i, _i, _vp, C.POINTER(C.c_float), C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_pose_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "fsgs_pose_backward": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "fsgs_pose_adam_step": (_i, [_vp, _vp, _i, _i, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i, _i,
                                 C.c_double, C.c_double, C.c_double, _vp, _vp]),
    "fsgs_pose_frame_begin": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_adam_step": (_i, [_i, C.POINTER(FsgsAdamGroup), C.c_double, C.c_double, C.c_double, _vp]),
    "fsgs_densify_plan": (_i, [_i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _i, _vp, _vp]),
    "fsgs_densify_apply": (_i, [_i, _vp, _vp, C.POINTER(C.c_int32), _i, C.POINTER(FsgsDensifyGroup), _vp, _vp, _vp,
                                _vp]),
    "fsgs_densify_stats": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsgs_pearson_backward": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "fsgs_event_create": (_i, [C.POINTER(C.c_void_p)]),
    "fsgs_event_destroy": (_i, [_vp]),
    "fsgs_stream_wait_event": (_i, [_vp, _vp]),
    "fsgs_event_record": (_i, [_vp, _vp]),
    "fsgs_forward_done_event": (_i, [_vp]),
    "fsgs_pose_step_done_event": (_i, [_vp]),
}


def exported_symbols():
    return sorted(_PROTOTYPES)


def load():
    """Load the HIP library; raises if it has not been built (python free-surgs_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libfsgs_hip.so is missing (%s). Build it with `python free-surgs_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for this path." % LIB_PATH
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code, where):
    if code != FSGS_OK:
        detail = ""
        if code == FSGS_ERR_HIP:
            detail = (load().fsgs_last_error() or b"").decode("utf-8", "replace")
        raise FsgsError(code, where, detail)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream():
    """raw handle of the calling thread's current stream on the current device (torch.cuda.current_stream() builds a Stream
    object and resolves the device three times over: ~10 us a call, six calls per step; the private accessor underneath it
    is ~1 us -- with the public route as the fallback should it ever go away)"""
    import torch

    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def profile_enable(names=None, stride=1):
    """Enable HIP-event timing for the named kernels (None = all, [] = off); stride n times every n-th launch."""
    lib = load()
    lib.fsgs_profile_stride(int(stride))
    n = lib.fsgs_profile_count()
    all_names = [lib.fsgs_profile_name(i).decode() for i in range(n)]
    mask = 0
    for i, nm in enumerate(all_names):
        if names is None or nm in names:
            mask |= 1 << i
    lib.fsgs_profile_enable(mask)
    return all_names


def profile_read():
    """{kernel name: (total_ms, launches)} for every kernel that ran while enabled."""
    lib = load()
    out = {}
    for i in range(lib.fsgs_profile_count()):
        ms, cnt = C.c_double(0), C.c_int64(0)
        lib.fsgs_profile_read(i, C.byref(ms), C.byref(cnt))
        if cnt.value:
            out[lib.fsgs_profile_name(i).decode()] = (ms.value, cnt.value)
    return out
