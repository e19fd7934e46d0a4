"""The C ABI's error behaviour on the device box: bad arguments come back as negative FSGS_ERR_* codes -- never a crash,
never a launch on garbage (include/fsgs.h: "Return value: 0 on success, negative FSGS_ERR_*")."""
import ctypes as C

import numpy as np
import pytest
import torch

from fsgs_amd import _lib, rasterizer, synth
from fsgs_amd.trainer import settings_from_cam

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _problem(P=50, W=64, H=48):
    cam = synth.make_camera(W, H)
    xyz, col, op, s, r = synth.random_small_scene(P, cam, seed=0)
    T = lambda a: torch.tensor(np.asarray(a, np.float32), device=DEV).contiguous()
    cfg = rasterizer.make_cfg(settings_from_cam(cam, DEV), 3)
    return cfg, P, W, H, T(xyz), T(col), T(op), T(s), T(r)


def test_raster_entry_points_reject_bad_arguments():
    lib = _lib.load()
    cfg, P, W, H, xyz, col, op, s, r = _problem()
    cap = 1 << 16
    sb, xb = C.c_size_t(0), C.c_size_t(0)
    assert lib.fsgs_raster_sizes(P, W, H, cap, C.byref(sb), C.byref(xb)) == _lib.FSGS_OK
    assert lib.fsgs_raster_sizes(-1, W, H, cap, C.byref(sb), C.byref(xb)) < 0
    assert lib.fsgs_raster_sizes(P, 0, H, cap, C.byref(sb), C.byref(xb)) < 0
    state = torch.empty(sb.value, dtype=torch.uint8, device=DEV)
    scratch = torch.empty(xb.value, dtype=torch.uint8, device=DEV)
    out, dep = torch.empty(3, H, W, device=DEV), torch.empty(1, H, W, device=DEV)
    radii = torch.empty(P, dtype=torch.int32, device=DEV)
    nr = C.c_int64(0)
    p = _lib.ptr
    stream = _lib.current_stream()

    def fwd(cfg_=cfg, P_=P, xyz_=xyz, state_bytes=sb.value, scratch_bytes=xb.value, cap_=cap, out_=out):
        return lib.fsgs_raster_forward(C.byref(cfg_) if cfg_ is not None else None, P_, p(xyz_), p(col), p(op), p(s),
                                       p(r), p(out_), p(dep), p(radii), p(state), state_bytes, p(scratch),
                                       scratch_bytes, cap_, C.byref(nr), stream)

    assert fwd() == _lib.FSGS_OK and nr.value > 0
    good = out.clone()
    assert fwd(cfg_=None) == _lib.FSGS_ERR_INVALID
    assert fwd(P_=-3) == _lib.FSGS_ERR_INVALID
    assert fwd(xyz_=None) == _lib.FSGS_ERR_INVALID
    assert fwd(out_=None) == _lib.FSGS_ERR_INVALID
    assert fwd(state_bytes=sb.value // 2) < 0          # too small a state buffer
    assert fwd(scratch_bytes=16) < 0
    assert fwd(cap_=-1) == _lib.FSGS_ERR_INVALID
    bad = _lib.FsgsRasterCfg()
    C.memmove(C.byref(bad), C.byref(cfg), C.sizeof(bad))
    for ch in (0, 2, 4, 5, 7, 9):
        bad.channels = ch
        assert fwd(cfg_=bad) == _lib.FSGS_ERR_INVALID, ch
    bad.channels, bad.image_width = 3, 0
    assert fwd(cfg_=bad) == _lib.FSGS_ERR_INVALID
    # capacity: one pair per segment is not enough -> reported with a sufficient max_pairs, and a retry with it works
    tiny = ((W + 15) // 16) * ((H + 15) // 16) * 8
    sb2, xb2 = C.c_size_t(0), C.c_size_t(0)
    assert lib.fsgs_raster_sizes(P, W, H, tiny, C.byref(sb2), C.byref(xb2)) == _lib.FSGS_OK
    rc = fwd(cap_=tiny)
    assert rc == _lib.FSGS_ERR_CAPACITY and nr.value > tiny
    need = int(nr.value)
    assert lib.fsgs_raster_sizes(P, W, H, need, C.byref(sb2), C.byref(xb2)) == _lib.FSGS_OK
    state = torch.empty(sb2.value, dtype=torch.uint8, device=DEV)
    scratch = torch.empty(xb2.value, dtype=torch.uint8, device=DEV)
    assert fwd(cap_=need, state_bytes=sb2.value, scratch_bytes=xb2.value) == _lib.FSGS_OK
    torch.cuda.synchronize()
    assert torch.equal(out, good)  # and none of the rejected calls disturbed anything


def test_fused_entry_points_reject_bad_arguments():
    lib = _lib.load()
    assert lib.fsgs_adam_step_compact(10, None, None, None, _lib.current_stream()) == _lib.FSGS_ERR_INVALID
    assert lib.fsgs_adam_step(-1, None, 0.9, 0.999, 1e-8, _lib.current_stream()) < 0
    assert lib.fsgs_densify_stats(5, None, None, None, None, None, _lib.current_stream()) == _lib.FSGS_ERR_INVALID
    assert lib.fsgs_photometric_loss_forward(3, 8, 8, None, None, None, None, 0.2, None, None, None,
                                             _lib.current_stream()) == _lib.FSGS_ERR_INVALID
    assert lib.fsgs_photometric_loss_forward(0, 8, 8, None, None, None, None, 0.2, None, None, None,
                                             _lib.current_stream()) == _lib.FSGS_ERR_INVALID
    n = C.c_size_t(0)
    assert lib.fsgs_knn_meandist2(-1, None, None, None, C.byref(n), None) < 0
    assert lib.fsgs_selftest_transpose_reduce_n(None, None, 32, _lib.current_stream()) == _lib.FSGS_ERR_INVALID
    x = torch.zeros(64 * 64, device=DEV)
    o = torch.zeros(64, device=DEV)
    assert lib.fsgs_selftest_transpose_reduce_n(_lib.ptr(x), _lib.ptr(o), 7, _lib.current_stream()) == _lib.FSGS_ERR_INVALID
    assert _lib.check(0, "ok") is None
    with pytest.raises(_lib.FsgsError):
        _lib.check(_lib.FSGS_ERR_INVALID, "demo")


def test_current_stream_handle_follows_torch_stream_contexts():
    """_lib.current_stream() reads the raw handle through torch's private accessor (1 us instead of 10): it must name the
    stream the calling thread has current, inside and outside `with torch.cuda.stream(...)`, on the current device."""
    from fsgs_amd import _lib

    assert (_lib.current_stream().value or 0) == torch.cuda.current_stream().cuda_stream  # (the default stream: handle 0 = NULL)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert (_lib.current_stream().value or 0) == side.cuda_stream == torch.cuda.current_stream().cuda_stream
        inner = torch.cuda.Stream()
        with torch.cuda.stream(inner):
            assert (_lib.current_stream().value or 0) == inner.cuda_stream
        assert (_lib.current_stream().value or 0) == side.cuda_stream
    assert (_lib.current_stream().value or 0) == torch.cuda.current_stream().cuda_stream


def test_a_kept_event_orders_a_side_stream_behind_and_back_into_the_main_stream():
    """fsgs_event_create / fsgs_event_record / fsgs_stream_wait_event (include/fsgs.h): the join the step driver makes every
    step with ONE event it keeps -- work enqueued on the main stream behind the wait sees what the side stream wrote before
    the record, over many re-records of the same event; a NULL event is FSGS_ERR_INVALID, not a crash."""
    lib = _lib.load()
    ev = C.c_void_p()
    assert lib.fsgs_event_create(C.byref(ev)) == _lib.FSGS_OK and ev.value
    assert lib.fsgs_event_record(None, None) == _lib.FSGS_ERR_INVALID
    assert lib.fsgs_stream_wait_event(None, None) == _lib.FSGS_ERR_INVALID
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    a = torch.zeros(1 << 22, device=DEV)
    out = []
    for k in range(1, 21):
        side.wait_stream(main)  # (the side stream's writer must not overtake the previous round's reader)
        with torch.cuda.stream(side):
            for _ in range(8):  # enough queued work that an unordered reader would run too early
                a.add_(1.0)
        assert lib.fsgs_event_record(ev, C.c_void_p(side.cuda_stream)) == _lib.FSGS_OK
        assert lib.fsgs_stream_wait_event(_lib.current_stream(), ev) == _lib.FSGS_OK
        out.append(a[::65536].clone())
    torch.cuda.synchronize()
    for k, o in enumerate(out, 1):
        assert torch.equal(o, torch.full_like(o, 8.0 * k)), k
    assert lib.fsgs_event_destroy(ev) == _lib.FSGS_OK
