"""distCUDA2 on the HIP C ABI (reference: submodules/simple-knn/ext.cpp:15-17, spatial.cu:15-26)."""
import ctypes as C

import torch

from . import _lib


def distCUDA2(points):
    """points [P,3] float32 cuda -> mean squared distance to the 3 nearest neighbours, [P] float32."""
    lib = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs a CUDA/HIP tensor; there is no CPU fallback")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("distCUDA2 expects points of shape [P, 3], got %s" % (tuple(points.shape),))
    pts = points.detach().contiguous().to(torch.float32)
    P = int(pts.shape[0])
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)  # spatial.cu:20: zeros
    nbytes = C.c_size_t(0)
    _lib.check(lib.fsgs_knn_meandist2(P, None, None, None, C.byref(nbytes), None), "fsgs_knn_meandist2(size)")
    scratch = torch.empty((nbytes.value,), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = lib.fsgs_knn_meandist2(P, _lib.ptr(pts), _lib.ptr(out), _lib.ptr(scratch), C.byref(nbytes),
                                    _lib.current_stream())
    _lib.check(rc, "fsgs_knn_meandist2")
    return out
