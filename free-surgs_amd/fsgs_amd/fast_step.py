"""Autograd-free mapping / tracking iterations: the same arithmetic as trainer.mapping_step /
tracking_step (train.py:166-200,236-272), but every stage is ONE C-ABI call into libfsgs_hip.so and the
chain rule between stages is written out by hand, so a step costs ~15 kernel launches on two streams and no
autograd graph (the torch-autograd path spends ~0.5 ms/step of pure host time in ~100 tiny kernels).

    mapping :  render fwd -> [rgb loss fwd, bwd (x5)  ||  patch draws, pearson(global + patches) fwd, bwd (x0.05, x0.15/n)]
               -> render bwd feeding Adam directly (1 view, 1 rank)
                  or render bwd -> compact [P,14] gradient (summed over views, all-reduced over ranks) -> Adam from it
               -> densification statistics
    tracking:  [flow loss fwd+bwd (one pass)  ||  render fwd] -> masked rgb loss fwd, bwd -> render bwd (pose only)
               -> quaternion/translation chain (12 floats) -> Adam

Equivalence with the autograd path is asserted in tests/test_fast_step_gpu.py.
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _lib, losses, optim, rasterizer
from .model import PARAM_NAMES
from .render_ops import _args_struct

LOSS_W_MAPPING = {"rgb": 5.0, "pearson": 0.05, "local_pearson": 0.15}  # train.py:254-258
LOSS_W_TRACKING = {"rgb": 1.0, "flow": 0.1}  # train.py:180-184
BOX, P_CORR = 128, 0.5  # train.py:257


class _Buffers:
    """Per-shape device buffers reused across steps (nothing here survives a change of P, H or W)."""

    def __init__(self, P, H, W, n_patches, dev):
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.key = (P, H, W, n_patches, str(dev))
        self.image, self.depth_sil = f(3, H, W), f(3, H, W)
        self.radii = torch.empty((P,), dtype=torch.int32, device=dev)
        self.maps = f(3, 3, H, W)
        self.sums = torch.empty((int(_lib.load().fsgs_photometric_scratch_bytes(3, H, W)),), dtype=torch.uint8,
                                device=dev)
        # every scalar loss term of a view in ONE buffer: [rgb loss, L1 mean, SSIM mean, pearson, local pearson, -, -, -]
        # so that the weighted sum is a single dot product
        self.terms = torch.zeros((8,), dtype=torch.float32, device=dev)
        self.rgb_out = self.terms[0:3]
        self.stats = torch.empty((int(_lib.load().fsgs_pearson_scratch_bytes(H, W, n_patches, BOX)),),
                                 dtype=torch.uint8, device=dev)
        self.coef = f(8 * (n_patches + 1))
        self.pe_out = self.terms[3:5]
        tw = np.zeros((8,), np.float32)
        tw[0], tw[3], tw[4] = LOSS_W_MAPPING["rgb"], LOSS_W_MAPPING["pearson"], LOSS_W_MAPPING["local_pearson"]
        self.term_w = torch.tensor(tw, device=dev)
        self.d_image = f(3, H, W)
        self.d_depth_sil = torch.zeros((3, H, W), dtype=torch.float32, device=dev)  # planes 1,2 stay zero
        self.up_rgb = torch.tensor([LOSS_W_MAPPING["rgb"]], dtype=torch.float32, device=dev)
        w = np.full((n_patches + 1,), LOSS_W_MAPPING["local_pearson"] / max(n_patches, 1), np.float32)
        w[0] = LOSS_W_MAPPING["pearson"]
        self.pe_w = torch.tensor(w, device=dev)
        tk = np.zeros((8,), np.float32)
        tk[0], tk[5] = LOSS_W_TRACKING["rgb"], LOSS_W_TRACKING["flow"]
        self.trk_w = torch.tensor(tk, device=dev)
        self.flow_out = self.terms[5:7]  # {flow loss, #valid points}
        self.flow_scratch = None
        self.d_flow = f(4, 4)
        self.gc = None  # this view's compact gradient [P,14] (lazily: only multi-view / multi-rank steps)
        self.zero = torch.zeros((), dtype=torch.float32, device=dev)
        self.means2D_grad = f(P, 3)
        # the backward's accumulator rows: zero here and zero again behind every backward (FSGS_FLAG_SCRATCH_SELF_CLEAN)
        self.bwd_scratch = torch.zeros((P * 64 + 512,), dtype=torch.uint8, device=dev)
        # True while a backward is under way: raised before its first launch, lowered when its LAST row chunk has been
        # enqueued.  A backward that raises in between (a rejected argument, an exchange that failed between two row chunks)
        # leaves it up, and the next backward of this buffer set runs without FSGS_FLAG_SCRATCH_ZEROED, i.e. clears the rows
        # itself instead of trusting them
        self.scratch_dirty = False
        self.sizes = {}


class FastStepper:
    def __init__(self, pc, poses, frames):
        self.pc, self.poses, self.frames = pc, poses, frames
        self.buf = None
        self.cfg = None
        self.cfg_key = None
        self.lib = _lib.load()
        self.last = {}
        self._check_frames()
        self.pairs_total = self.forward_calls = 0
        self.fuse_adam = True  # single-view steps on one rank: Adam inside the render backward
        self.compact = True    # multi-view / multi-rank steps: [P,14] gradient + fsgs_adam_step_compact
        self.fuse_pose = True  # tracking: pose adjoint + Adam + next pose in one launch
        # (FSGS_OVERLAP_VIEWS=0 / FSGS_CACHE_COLORS=0: bisection switches, diagnostics only)
        # FSGS_LOSS_STREAMS=1: the loss stage as two launches on ONE stream (fsgs_view_losses_forward_backward, round 6).  Measured
        # equal at 1280x1024 (0.5785 vs 0.5776 ms / step: the fork / join gaps a rocprofv3 timeline shows between the two streams
        # are the tracer's, not the step's) and slower at 640x512 (0.190 vs 0.172): the default stays two streams
        # (profiles/r06_ab_loss_streams.txt)
        self.one_stream_losses = os.environ.get("FSGS_LOSS_STREAMS", "2") == "1"
        self.overlap_views = os.environ.get("FSGS_OVERLAP_VIEWS", "1") != "0"  # multi-view mapping steps: views >= 1 on their own streams beside view 0
        self.cache_colors = os.environ.get("FSGS_CACHE_COLORS", "1") != "0"    # the Adam kernels leave the next forward's per-Gaussian colours behind (16 B instead of 192 B)
        self.cache_hits = 0        # forwards served from that cache (tests / diagnostics)
        # True: mapping forwards blend the planes the step reads -- image + depth, like the tracking forward (train.py:250-258:
        # rgb_loss on `render`, the Pearson losses on `render_dep`; the silhouette and depth^2 planes feed render()'s
        # `render_opacity` / `uncertainty`, which no training step reads).  Measured at C2: blend_fwd 137.8 -> 136.2 us (the
        # colour FMAs are packed two channels an instruction either way), so the default keeps all six planes.
        self.mapping_planes4 = False

    def _check_frames(self):
        """the per-frame targets are handed to the kernels as raw pointers: float32, contiguous, on the cloud's device,
        [3,H,W] colours and [H,W] mono-depths of one size (checked once here, not per step)"""
        dev = self.pc.params["_xyz"].device
        hw = None
        for name, seq, lead in (("colors", self.frames.colors, (3,)), ("monodeps", self.frames.monodeps, ())):
            if hasattr(seq, "prefetch"):  # a staged lane (staging.py): contiguous float32 device buffers of one shape by construction
                if seq.device != dev or (seq.shape is not None and len(seq.shape) != len(lead) + 2):
                    raise ValueError("frames.%s is staged on %s with item shape %s" % (name, seq.device, seq.shape))
                continue
            for i, t in enumerate(seq or []):
                if t is None:
                    continue
                ok = (torch.is_tensor(t) and t.dtype == torch.float32 and t.is_contiguous() and t.device == dev
                      and t.dim() == len(lead) + 2 and tuple(t.shape[:len(lead)]) == lead)
                if ok and hw is None:
                    hw = tuple(t.shape[-2:])
                if not ok or tuple(t.shape[-2:]) != hw:
                    raise ValueError("frames.%s[%d] must be a contiguous float32 tensor of shape %s on %s" % (
                        name, i, lead + (hw or ("H", "W")), dev))

    # ---- helpers -----------------------------------------------------------------------------------------
    def _buffers(self, P, H, W, n_patches, dev, view=0):
        """buffer set of view `view` of a step (view 0 = self.buf; the further views of a multi-view step own theirs,
        so that their pipelines can run beside view 0's on another stream)"""
        key = (P, H, W, n_patches, str(dev))
        if view == 0:
            if self.buf is None or self.buf.key != key:
                self.buf = _Buffers(P, H, W, n_patches, dev)
            return self.buf
        extra = self.__dict__.setdefault("_view_bufs", {})
        b = extra.get(view)
        if b is None or b.key != key:
            b = extra[view] = _Buffers(P, H, W, n_patches, dev)
        return b

    def _cfg(self):
        cam = self.pc.cam
        # the settings object itself is part of the key and kept alive by it (an id() alone can be recycled)
        # (+ the process-wide switches make_cfg folds into the flags: toggling them re-makes the struct here too)
        key = (cam, cam.viewmatrix._version, cam.projmatrix._version, cam.bg._version, rasterizer.blend_variant(),
               rasterizer.deterministic())
        if (self.cfg_key is None or self.cfg_key[0] is not cam or self.cfg_key[1:] != key[1:]):
            self.cfg = rasterizer.make_cfg(cam, 6)
            # this driver only ever hands the backward a gradient of the depth plane (Pearson losses): the silhouette and
            # depth^2 planes of d_depth_sil stay zero, and the flag lets the backward blend drop their terms
            self.cfg.flags |= _lib.FSGS_FLAG_DEPTH_GRAD_ONLY
            # every backward of this driver leaves its accumulator rows zero (the per-Gaussian kernel stores zeros over what
            # it has read): no fill launch, and no stream bookkeeping to keep a fill clear of the previous backward
            self.cfg.flags |= _lib.FSGS_FLAG_SCRATCH_SELF_CLEAN
            self.cfg_key = key
        return self.cfg

    def _cfg_zeroed(self):
        """the same configuration with FSGS_FLAG_SCRATCH_ZEROED: the scratch of this driver's buffer sets starts out zero and
        every backward cleans up behind itself (FSGS_FLAG_SCRATCH_SELF_CLEAN, _cfg)"""
        base = self._cfg()
        if getattr(self, "_cfgz_of", None) is not base:
            z = _lib.FsgsRasterCfg()
            C.memmove(C.byref(z), C.byref(base), C.sizeof(z))
            z.flags |= _lib.FSGS_FLAG_SCRATCH_ZEROED
            self._cfgz, self._cfgz_of = z, base
        return self._cfgz

    def _cfg_backward(self, b, trust=True, cap=None):
        """configuration of a backward on buffer set `b`: the accumulator rows are known to be zero (SCRATCH_ZEROED) unless
        the previous backward on them did not run to its end (or trust=False: the call clears them itself); marks the set
        dirty until _backward_done(b).  cap: the forward's pair capacity -- the deterministic backward (tests) keeps a row
        per pair slot behind the accumulator rows, so the scratch grows with it"""
        if cap is not None and (self._cfg().flags & _lib.FSGS_FLAG_DETERMINISTIC):
            need = rasterizer.backward_scratch_bytes(self._cfg(), self.pc.num_points, cap, b.bwd_scratch.numel())
            if b.bwd_scratch.numel() < need:
                b.bwd_scratch = torch.zeros((need,), dtype=torch.uint8, device=b.bwd_scratch.device)
                b.scratch_dirty = False
        cfg = self._cfg() if (b.scratch_dirty or not trust) else self._cfg_zeroed()
        b.scratch_dirty = True
        return cfg

    @staticmethod
    def _backward_done(b):
        b.scratch_dirty = False

    def _cfg_tracking(self):
        """the same configuration with FSGS_FLAG_RGB_DEPTH_ONLY: the tracking iteration reads the image and the depth
        plane (`render_dep > 0`) and nothing else, so the silhouette and depth^2 planes are not blended"""
        base = self._cfg()
        if getattr(self, "_cfgt_of", None) is not base:
            z = _lib.FsgsRasterCfg()
            C.memmove(C.byref(z), C.byref(base), C.sizeof(z))
            z.flags |= _lib.FSGS_FLAG_RGB_DEPTH_ONLY
            self._cfgt, self._cfgt_of = z, base
        return self._cfgt

    def _render_forward(self, w2c, b, tracking=False, allow_reuse=True, done_event=None):
        # tracking = True: FSGS_FLAG_RGB_DEPTH_ONLY, the forward blends image + depth plane only (tracking AND mapping steps)
        pc, lib = self.pc, self.lib
        p = pc.params
        for name in PARAM_NAMES:  # raw pointers go to the kernels: no silent reinterpretation of other layouts
            t_ = p[name]
            if t_.dtype != torch.float32 or not t_.is_contiguous() or not t_.is_cuda:
                raise ValueError("%s must be a contiguous float32 device tensor for the step driver (got %s, %s)"
                                 % (name, t_.dtype, "contiguous" if t_.is_contiguous() else "strided"))
        cfg = self._cfg_tracking() if tracking else self._cfg()
        P, H, W = pc.num_points, cfg.image_height, cfg.image_width
        dev = p["_xyz"].device
        args = _args_struct(p["_xyz"], p["_features_dc"], p["_features_rest"], p["_opacity"], p["_scaling"],
                            p["_rotation"], w2c, self.poses.cam_center, pc.active_sh_degree, pc.max_sh_degree)
        cap = rasterizer._capacity_for(P, W, H)
        nr = C.c_int64(0)
        stream = _lib.current_stream()
        # The per-Gaussian colours depend on the parameters and the frame-0 camera centre only: while neither changes
        # (the 50 tracking iterations of a frame, the second view of a two-view mapping step) a forward copies them from
        # the previous forward's state instead of evaluating 48 SH coefficients per Gaussian again.
        ckey = self._color_key()
        # (the tensors themselves are part of the entry: ids alone can be recycled by the allocator)
        owners = tuple(p[n] for n in PARAM_NAMES) + (self.poses.cam_center,)
        prev = self.__dict__.get("_color_src") if (allow_reuse and getattr(self, "reuse_colors", True)) else None
        if prev is not None and (prev[0] != ckey or len(prev[4]) != len(owners) or any(a is not b for a, b in zip(prev[4], owners))):
            prev = None
        # ... and a MAPPING step's forward takes them from the colour cache the previous step's Adam kernel filled right
        # after it updated the parameters (FsgsFusedAdam.next_colors), valid under the same identity / version test
        # (allow_reuse = False only rules out the copy from ANOTHER forward's state -- a view on its own stream must not
        # wait for view 0's forward; the cache was complete before the step began)
        cached = self.__dict__.get("_color_cache") if getattr(self, "reuse_colors", True) else None
        if cached is not None and (cached[0] != ckey or len(cached[2]) != len(owners) or
                                   any(a_ is not b_ for a_, b_ in zip(cached[2], owners))):
            cached = None
        if cached is not None:
            prev = None
        for _attempt in range(3):
            sz = b.sizes.get(cap)
            if sz is None:
                sb, xb = C.c_size_t(0), C.c_size_t(0)
                _lib.check(lib.fsgs_render_sizes(P, W, H, cap, C.byref(sb), C.byref(xb)), "fsgs_render_sizes")
                sz = b.sizes[cap] = (sb.value, xb.value)
            state = torch.empty((sz[0],), dtype=torch.uint8, device=dev)
            scratch = torch.empty((sz[1],), dtype=torch.uint8, device=dev)
            if done_event is not None:  # signalled by the forward blend's own completion (one-shot: set before every attempt)
                _lib.check(lib.fsgs_forward_done_event(done_event), "fsgs_forward_done_event")
            if cached is not None:
                cached[1].record_stream(torch.cuda.current_stream())
                rc = lib.fsgs_render_forward_cached_colors(
                    C.byref(cfg), P, C.byref(args), _lib.ptr(b.image), _lib.ptr(b.depth_sil), _lib.ptr(b.radii),
                    _lib.ptr(state), sz[0], _lib.ptr(scratch), sz[1], cap, C.byref(nr), _lib.ptr(cached[1]), stream)
            elif prev is not None:
                # (the source state may belong to another stream's pool: view 0's forward, read by view 1's on its own)
                prev[1].record_stream(torch.cuda.current_stream())
                rc = lib.fsgs_render_forward_reuse_colors(
                    C.byref(cfg), P, C.byref(args), _lib.ptr(b.image), _lib.ptr(b.depth_sil), _lib.ptr(b.radii),
                    _lib.ptr(state), sz[0], _lib.ptr(scratch), sz[1], cap, C.byref(nr), _lib.ptr(prev[1]), prev[2], prev[3],
                    stream)
            else:
                rc = lib.fsgs_render_forward(C.byref(cfg), P, C.byref(args), _lib.ptr(b.image), _lib.ptr(b.depth_sil),
                                             _lib.ptr(b.radii), _lib.ptr(state), sz[0], _lib.ptr(scratch), sz[1], cap,
                                             C.byref(nr), stream)
            if rc == _lib.FSGS_ERR_CAPACITY and nr.value > cap:
                cap = int(nr.value * 1.25) + 1024
                rasterizer._capacity[(P, W, H)] = cap
                continue
            _lib.check(rc, "fsgs_render_forward")
            break
        else:
            raise _lib.FsgsError(_lib.FSGS_ERR_CAPACITY, "fsgs_render_forward")
        if cached is not None:
            self.cache_hits += 1
        if (prev is not None or cached is not None) and os.environ.get("FSGS_CHECK_REUSE") == "1":
            self._check_reused_colors(cfg, P, W, H, args, state, sz, cap, b)
        self._color_src = (ckey, state, sz[0], cap, owners)
        rasterizer.last_num_rendered = int(nr.value)
        self.pairs_total += int(nr.value)  # (bench.py reports the mean pair count of the steps it timed)
        self.forward_calls += 1
        return args, state, sz[0], cap, int(nr.value)

    def _color_key(self):
        """what the per-Gaussian colours depend on: the cloud's size, the SH degrees, the frame-0 camera centre and the
        parameter tensors (object ids here; the objects themselves are compared by the callers) with their versions"""
        pc, p = self.pc, self.pc.params
        cfg = self._cfg()
        return (pc.num_points, cfg.image_width, cfg.image_height, pc.active_sh_degree, pc.max_sh_degree,
                self.poses.cam_center._version) + tuple((id(p[n]), p[n]._version) for n in PARAM_NAMES)

    def _next_colors_buffer(self):
        """[P,4] the Adam kernel of this step fills for the next forward (FsgsFusedAdam.next_colors); None = not wanted"""
        if not getattr(self, "cache_colors", True):
            return None
        P = self.pc.num_points
        buf = self.__dict__.get("_next_colors")
        dev = self.pc.params["_xyz"].device
        if buf is None or buf.shape[0] != P or buf.device != dev:
            buf = self._next_colors = torch.empty((P, 4), dtype=torch.float32, device=dev)
        return buf

    def _colors_cached(self, buf):
        """call right after optim.mark_updated(): `buf` holds the colours of the parameters as they stand now"""
        if buf is None:
            self._color_cache = None
            return
        p = self.pc.params
        self._color_cache = (self._color_key(), buf, tuple(p[n] for n in PARAM_NAMES) + (self.poses.cam_center,))

    def _check_reused_colors(self, cfg, P, W, H, args, state, sz, cap, b):
        """FSGS_CHECK_REUSE=1 (debugging): the reuse above rests on every raw-pointer writer bumping the version
        counters (optim.mark_updated); here the colours are evaluated afresh into a throw-away state and compared with
        the ones the forward just copied -- a writer that forgot the bump shows up as a mismatch instead of as a
        silently stale image."""
        lib = self.lib
        dev = state.device
        st2 = torch.empty_like(state)
        scr = torch.empty((sz[1],), dtype=torch.uint8, device=dev)
        img, ds, rad = torch.empty_like(b.image), torch.empty_like(b.depth_sil), torch.empty_like(b.radii)
        nr = C.c_int64(0)
        _lib.check(lib.fsgs_render_forward(C.byref(cfg), P, C.byref(args), _lib.ptr(img), _lib.ptr(ds), _lib.ptr(rad),
                                           _lib.ptr(st2), sz[0], _lib.ptr(scr), sz[1], cap, C.byref(nr),
                                           _lib.current_stream()), "fsgs_render_forward (FSGS_CHECK_REUSE)")
        off = (C.c_size_t * 9)()
        _lib.check(lib.fsgs_render_state_layout(P, W, H, cap, off), "fsgs_render_state_layout")
        rec = lambda s_: s_[off[7]:off[7] + 64 * P].view(torch.float32).reshape(P, 16)[:, 8:14]
        vis = rad > 0
        if not torch.equal(rec(state)[vis], rec(st2)[vis]):
            raise RuntimeError("FSGS_CHECK_REUSE: the per-Gaussian colours copied from the previous forward differ from "
                               "freshly evaluated ones -- a parameter or cam_center was written without a version bump")

    def _fused_adam_struct(self):
        """FsgsFusedAdam for the six groups of pc.optimizer (state created on first use, step counters advanced:
        exactly what FusedAdam.step() does before its launch)."""
        opt = self.pc.optimizer
        adam = _lib.FsgsFusedAdam()
        by_name = {g["name"]: g for g in opt.param_groups}
        for k, name in enumerate(PARAM_NAMES):
            g = by_name[name]
            p = g["params"][0]
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] = int(st["step"]) + 1
            adam.exp_avg[k], adam.exp_avg_sq[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            adam.lr[k], adam.step[k] = float(g["lr"]), int(st["step"])
            adam.beta1, adam.beta2, adam.eps = float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])
        nc = self._next_colors_buffer()
        adam.next_colors = None if nc is None else nc.data_ptr()
        self._adam_next_colors = nc
        return adam

    @staticmethod
    def _offset_structs(args, adam, lo):
        """the same FsgsRenderArgs / FsgsFusedAdam with every per-Gaussian pointer advanced by `lo` Gaussians"""
        if lo == 0:
            return args, adam
        rows = (3, 3, 45, 1, 3, 4)  # floats per Gaussian: xyz, f_dc, f_rest, opacity, scaling, rotation
        a2 = _lib.FsgsRenderArgs()
        C.memmove(C.byref(a2), C.byref(args), C.sizeof(a2))
        rest_row = ((int(args.max_sh_degree) + 1) ** 2 - 1) * 3
        for name, r in zip(("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"),
                           (3, 3, rest_row, 1, 3, 4)):
            v = getattr(args, name)
            setattr(a2, name, None if not v else v + 4 * r * lo)
        ad2 = _lib.FsgsFusedAdam()
        C.memmove(C.byref(ad2), C.byref(adam), C.sizeof(ad2))
        for g, r in enumerate((3, 3, rest_row, 1, 3, 4)):
            if adam.exp_avg[g]:
                ad2.exp_avg[g] = adam.exp_avg[g] + 4 * r * lo
                ad2.exp_avg_sq[g] = adam.exp_avg_sq[g] + 4 * r * lo
        if adam.next_colors:
            ad2.next_colors = adam.next_colors + 16 * lo
        return a2, ad2

    def _side_stream(self, dev, view=0):
        """the second stream of a view's pipeline (Pearson chain / flow loss beside the photometric kernels)"""
        pool = self.__dict__.setdefault("_sides", {})
        st = pool.get(view)
        if st is None or st.device != torch.device(dev):
            st = pool[view] = torch.cuda.Stream(device=dev)
        return st

    def _view_stream(self, dev, view):
        """the stream the whole pipeline of view `view` >= 1 of a multi-view step runs on"""
        pool = self.__dict__.setdefault("_view_streams", {})
        st = pool.get(view)
        if st is None or st.device != torch.device(dev):
            st = pool[view] = torch.cuda.Stream(device=dev)
        return st

    def _render_backward(self, args, state, sbytes, cap, nr, b, d_image, d_depth_sil, grads, gs_grad, cam_grad,
                         param_grads, zeroed=False):
        cfg = self._cfg_backward(b, trust=zeroed, cap=cap)
        rc = self.lib.fsgs_render_backward(C.byref(cfg), self.pc.num_points, C.byref(args), _lib.ptr(b.radii),
                                           _lib.ptr(state), sbytes, cap, nr, _lib.ptr(d_image), _lib.ptr(d_depth_sil),
                                           int(gs_grad), int(cam_grad), int(param_grads), C.byref(grads),
                                           _lib.ptr(b.bwd_scratch), b.bwd_scratch.numel(), _lib.current_stream())
        _lib.check(rc, "fsgs_render_backward")
        self._backward_done(b)

    @staticmethod
    def _grad_struct(tensors, means2D, w2c):
        g = _lib.FsgsRenderGrads()
        (g.xyz, g.features_dc, g.features_rest, g.opacity, g.scaling, g.rotation) = [
            None if t is None else t.data_ptr() for t in tensors]
        g.means2D = None if means2D is None else means2D.data_ptr()
        g.w2c = None if w2c is None else w2c.data_ptr()
        return g

    def _step_tail(self, b, weights, with_stats):
        """FsgsStepTail for the per-Gaussian backward launch: the iteration's scalar loss (sum of the loss kernels'
        terms times `weights`) into a fresh 0-d tensor and, with_stats, the densification statistics of view 0.
        -> (struct, total tensor, tensors to keep alive until the launch is enqueued)"""
        pc = self.pc
        t = _lib.FsgsStepTail()
        total = torch.empty((), dtype=torch.float32, device=b.terms.device)
        t.loss_terms, t.loss_weights, t.n_terms, t.loss_total = (b.terms.data_ptr(), weights.data_ptr(),
                                                                int(b.terms.numel()), total.data_ptr())
        keep = []
        if with_stats:
            v = pc.variables
            for k in ("max_radii2D", "xyz_gradient_accum", "denom"):
                if not (v[k].is_contiguous() and v[k].dtype == torch.float32):
                    v[k] = v[k].contiguous().float()
            t.max_radii2D = v["max_radii2D"].data_ptr()
            keep = [v["max_radii2D"]]
            if with_stats != "radii":  # "radii": a further view of the step raises max_radii2D only (FsgsStepTail)
                t.xyz_gradient_accum, t.denom = v["xyz_gradient_accum"].data_ptr(), v["denom"].data_ptr()
                keep += [v["xyz_gradient_accum"], v["denom"]]
        return t, total, keep

    # ---- mapping (train.py:236-272) ------------------------------------------------------------------------
    def _view_forward(self, b, ts, corners, dev, H, W, view, allow_reuse=True):
        """First part of one view's pipeline on the CURRENT stream (+ the view's side stream): the patch draws on the side
        stream, then the render forward (returns when the pair count has arrived, i.e. as the forward blend starts).
        -> context for _view_losses / the backward"""
        w2c = (self.poses.get_pose_detached(ts) if hasattr(self.poses, "get_pose_detached")
               else self.poses.get_pose(ts).detach().contiguous())
        # The photometric chain (LDS / VALU bound) and the Pearson chain (bandwidth / latency bound) are independent
        # until the render backward: they run on two HIP streams.  What the side stream can do without the render
        # -- the two randint launches of the patch corners (same position in the RNG sequence as at loss time:
        # nothing else draws in between) -- is queued before the forward, so that the chain behind the forward is the
        # three Pearson launches only.  (Until round 3 the side stream also cleared the backward's accumulators here,
        # behind an event that kept the fill clear of the previous backward: the per-Gaussian backward now leaves them
        # zero itself, FSGS_FLAG_SCRATCH_SELF_CLEAN, and the event packet in front of the forward is gone.)
        side = self._side_stream(dev, view)
        with torch.cuda.stream(side):  # (nothing here depends on the current stream: no event in front of the forward)
            cr = corners if corners is not None else losses.draw_patch_corners(H, W, BOX, P_CORR, dev)
        b.corners_drawn = None
        if corners is None and self.one_stream_losses:
            # the one-stream loss stage reads the corners on the CURRENT stream: an event behind the draws, waited for at the
            # loss stage only if it has not fired by then (it has: the draws run at once on the idle side stream, a forward ahead)
            if getattr(b, "draw_event", None) is None:
                b.draw_event = torch.cuda.Event()
            b.draw_event.record(side)
            b.corners_drawn = b.draw_event
        # the side stream's Pearson chain waits for the forward's outputs: an event that rides on the forward blend's launch
        # (fsgs_forward_done_event) instead of a marker recorded behind it, which would sit between the blend and the first
        # loss kernel on this stream (~6 us)
        if getattr(b, "fwd_event", None) is None:
            ev = C.c_void_p()
            _lib.check(self.lib.fsgs_event_create(C.byref(ev)), "fsgs_event_create")
            b.fwd_event = ev
            weakref.finalize(b, self.lib.fsgs_event_destroy, ev).atexit = False  # (at interpreter exit the runtime cleans up itself)
        args, state, sbytes, cap, nr = self._render_forward(w2c, b, tracking=getattr(self, "mapping_planes4", False),
                                                            allow_reuse=allow_reuse, done_event=b.fwd_event)
        return {"args": args, "state": state, "sbytes": sbytes, "cap": cap, "nr": nr, "fwd_done": b.fwd_event, "cr": cr,
                "side": side, "ts": ts}

    def _view_losses(self, b, ctx, H, W, n_patches):
        """Second part, same streams: L1+SSIM forward / backward on the current stream, Pearson forward / backward on
        the side stream.  Leaves dL/dimage in b.d_image, dL/ddepth in b.d_depth_sil[0], the loss terms in b.terms."""
        lib = self.lib
        stream = _lib.current_stream()
        ts, cr, side = ctx["ts"], ctx["cr"], ctx["side"]
        gt, mono = self.frames.colors[ts], self.frames.monodeps[ts]
        if self.one_stream_losses and n_patches <= 63 and BOX <= 128:
            # Round 6: the whole stage in two launches on THIS stream (fsgs_view_losses_forward_backward): the Pearson
            # statistics ride in the photometric forward's launch, the Pearson gradient in the backward's.  The fork / join
            # events of the two-stream layout below cost 10 + 15 us of idle GPU per iteration at 1280x1024.
            drawn = getattr(b, "corners_drawn", None)
            if drawn is not None and not drawn.query():
                torch.cuda.current_stream().wait_event(drawn)
            _lib.check(lib.fsgs_view_losses_forward_backward(
                3, H, W, _lib.ptr(b.image), _lib.ptr(gt), None, None, 0.2, _lib.ptr(b.maps), _lib.ptr(b.sums),
                _lib.ptr(b.rgb_out), _lib.ptr(b.up_rgb), _lib.ptr(b.d_image), n_patches, BOX, _lib.ptr(cr[0]), _lib.ptr(cr[1]),
                _lib.ptr(mono), _lib.ptr(b.depth_sil[0]), _lib.ptr(b.stats), _lib.ptr(b.coef), _lib.ptr(b.pe_out),
                _lib.ptr(b.pe_w), _lib.ptr(b.d_depth_sil[0]), stream), "fsgs_view_losses_forward_backward")
            if drawn is not None:  # drawn on the side stream, last read on this one
                cur = torch.cuda.current_stream()
                for t_ in cr:
                    t_.record_stream(cur)
            return
        # forward + backward in two launches (the loss value is finished by an extra workgroup of the backward)
        _lib.check(lib.fsgs_photometric_loss_forward_backward(
            3, H, W, _lib.ptr(b.image), _lib.ptr(gt), None, None, 0.2, _lib.ptr(b.maps), _lib.ptr(b.sums),
            _lib.ptr(b.rgb_out), _lib.ptr(b.up_rgb), _lib.ptr(b.d_image), stream),
            "fsgs_photometric_loss_forward_backward")
        sstream = C.c_void_p(side.cuda_stream)
        _lib.check(lib.fsgs_stream_wait_event(sstream, ctx["fwd_done"]), "fsgs_stream_wait_event")
        ready = getattr(self.frames.monodeps, "ready", None)  # a staged lane: the side stream reads the mono-depth too
        if ready is not None and ready(ts) is not None:
            side.wait_event(ready(ts))
        # the two Pearson launches go to the side stream by its raw handle and the join back is an event this buffer set keeps
        # (round 5: a `with torch.cuda.stream(...)` block, a fresh torch.cuda.Event and torch.cuda.current_stream() per step
        # were ~25 us of host time, scripts/dev/host_microbench.py -- at 640x512 the step is as much host- as GPU-bound)
        dep = b.depth_sil[0]
        _lib.check(lib.fsgs_pearson_forward(H, W, n_patches, BOX, _lib.ptr(cr[0]), _lib.ptr(cr[1]),
                                            _lib.ptr(mono), _lib.ptr(dep), _lib.ptr(b.stats), _lib.ptr(b.coef),
                                            _lib.ptr(b.pe_out), sstream), "fsgs_pearson_forward")
        _lib.check(lib.fsgs_pearson_backward(H, W, n_patches, BOX, _lib.ptr(cr[0]), _lib.ptr(cr[1]),
                                             _lib.ptr(mono), _lib.ptr(dep), _lib.ptr(b.coef), _lib.ptr(b.pe_w),
                                             0, _lib.ptr(b.d_depth_sil[0]), sstream), "fsgs_pearson_backward")
        if getattr(b, "side_event", None) is None:
            ev = C.c_void_p()
            _lib.check(lib.fsgs_event_create(C.byref(ev)), "fsgs_event_create")
            b.side_event = ev
            weakref.finalize(b, _lib.load().fsgs_event_destroy, ev).atexit = False
        _lib.check(lib.fsgs_event_record(b.side_event, sstream), "fsgs_event_record")
        for t_ in cr:  # last used on the side stream (drawn there, or handed in by the caller from another stream)
            t_.record_stream(side)
        _lib.check(lib.fsgs_stream_wait_event(stream, b.side_event), "fsgs_stream_wait_event")

    def _view_forward_and_losses(self, b, ts, corners, dev, H, W, n_patches, view):
        """Front half of one view's pipeline on the CURRENT stream (+ the view's side stream): render forward, L1+SSIM
        forward / backward, patch draws + Pearson forward / backward.  -> (args, state, sbytes, cap, nr, event behind
        the forward)"""
        ctx = self._view_forward(b, ts, corners, dev, H, W, view)
        self._view_losses(b, ctx, H, W, n_patches)
        return ctx["args"], ctx["state"], ctx["sbytes"], ctx["cap"], ctx["nr"], ctx["fwd_done"]

    def mapping_step(self, timesteps, step_optimizer=True, grad_sync=None, corners=None, reduce_compact=None,
                     collect_stats=True):
        """One mapping iteration over `timesteps` (summed loss).  Gradient routes:
          * one view, no reduction, optimizer stepped here  -> Adam inside the render backward (no gradient tensors);
          * several views and / or `reduce_compact(tensor)` (the data-parallel all-reduce) -> one compact [P,14]
            gradient per view; fsgs_adam_step_compact_sum consumes the sum of two views' directly, an exchange gets
            their sum in one buffer.  The views' pipelines (forward -> losses -> backward) are independent until Adam:
            view k >= 1 runs on a stream of its own with its own buffers (`overlap_views`), its latency-bound
            preprocess / binning kernels and the tails of its blend kernels filling the SIMDs beside view 0's
            issue-bound blend kernels (train.py:236-259 runs them one after the other);
          * step_optimizer=False or the legacy `grad_sync(pc)` -> full gradients in the parameters' .grad.
        collect_stats=False (only with the first two routes): the densification statistics are not accumulated --
        for iterations past the last densification (train.py:305: `iteration < 15000`), where nothing reads them."""
        pc, lib = self.pc, self.lib
        dev = pc.params["_xyz"].device
        H, W = int(pc.cam.image_height), int(pc.cam.image_width)
        n_patches = int(P_CORR * (H // BOX) * (W // BOX))
        total = None
        stats_done = False  # the statistics of view 0 went into the backward launch itself
        with torch.no_grad(), torch.cuda.device(dev):
            b = self._buffers(pc.num_points, H, W, n_patches, dev)
            stream = _lib.current_stream()
            fused_ok = step_optimizer and grad_sync is None and isinstance(pc.optimizer, optim.FusedAdam)
            # single view, single rank: the backward feeds Adam directly (no gradient tensors at all)
            fuse_adam = self.fuse_adam and fused_ok and len(timesteps) == 1 and reduce_compact is None
            if fuse_adam:
                ts = timesteps[0]
                args, state, sbytes, cap, nr, _ = self._view_forward_and_losses(b, ts, corners, dev, H, W, n_patches, 0)
                adam = self._fused_adam_struct()
                cfg = self._cfg_backward(b, cap=cap)
                # statistics and the scalar loss ride in the same launch (no densify_stats / dot kernels)
                tail, total, _keep = self._step_tail(b, b.term_w, collect_stats)
                _lib.check(lib.fsgs_render_backward_adam(C.byref(cfg), pc.num_points, C.byref(args), _lib.ptr(b.radii),
                                                         _lib.ptr(state), sbytes, cap, nr, _lib.ptr(b.d_image),
                                                         _lib.ptr(b.d_depth_sil), C.byref(adam),
                                                         _lib.ptr(b.means2D_grad) if collect_stats else None,
                                                         C.byref(tail), _lib.ptr(b.bwd_scratch),
                                                         b.bwd_scratch.numel(), stream), "fsgs_render_backward_adam")
                self._backward_done(b)
                optim.mark_updated([pc.params[n_] for n_ in PARAM_NAMES])
                self._colors_cached(self._adam_next_colors)
                step_optimizer = False  # done
                stats_done = True
                radii0, last_b = b.radii, b
            elif self.compact and fused_ok:
                radii0, last_b, state, cap = self._mapping_compact(timesteps, corners, reduce_compact, collect_stats,
                                                                   dev, H, W, n_patches)
                total = self._compact_total
                stats_done = collect_stats
                step_optimizer = False  # done
            else:
                for k, ts in enumerate(timesteps):
                    args, state, sbytes, cap, nr, _ = self._view_forward_and_losses(b, ts, corners, dev, H, W, n_patches, 0)
                    # render backward straight into the parameters' .grad (view 0) or a scratch set that is added
                    first = k == 0
                    tgt = []
                    for name in PARAM_NAMES:
                        p = pc.params[name]
                        if first:
                            if p.grad is None:
                                p.grad = torch.empty_like(p)
                            tgt.append(p.grad)
                        else:
                            tgt.append(torch.empty_like(p))
                    m2 = b.means2D_grad if first else torch.empty_like(b.means2D_grad)
                    grads = self._grad_struct(tgt, m2, None)
                    self._render_backward(args, state, sbytes, cap, nr, b, b.d_image, b.d_depth_sil, grads, True, False, True)
                    if not first:
                        for name, t in zip(PARAM_NAMES, tgt):
                            pc.params[name].grad.add_(t)
                        # render() itself raises max_radii2D for EVERY rendered view (gaussian_renderer/__init__.py:79)
                        pc.variables["max_radii2D"] = torch.maximum(pc.variables["max_radii2D"], b.radii.float())
                    loss_k = torch.dot(b.terms, b.term_w)
                    total = loss_k if total is None else total + loss_k
                    if first:  # densification statistics come from view 0 only (train.py:260-263)
                        radii0 = b.radii if len(timesteps) == 1 else b.radii.clone()
                last_b = b
            if grad_sync is not None:
                grad_sync(pc)
            if collect_stats and not stats_done:
                optim.densify_stats(radii0, b.means2D_grad, pc.variables["max_radii2D"],
                                    pc.variables["xyz_gradient_accum"], pc.variables["denom"])
            self.last = {"radii": radii0, "viewspace_grad": b.means2D_grad, "image": last_b.image,
                         "depth_sil": last_b.depth_sil, "state": state, "max_pairs": cap, "radii_last": last_b.radii,
                         "P": pc.num_points}  # LAST view (debugging / statistics)
            if step_optimizer:
                pc.optimizer.step()
                # gradients are overwritten by the next step's backward; nothing to zero
        return total

    def _mapping_compact(self, timesteps, corners, reduce_compact, collect_stats, dev, H, W, n_patches):
        """The compact-gradient route of mapping_step: every view's backward leaves its [P,14] gradient in the view's own
        buffer set; views >= 1 run on their own streams when `overlap_views`; Adam consumes the (reduced) sum."""
        pc, lib = self.pc, self.lib
        P = pc.num_points
        main = torch.cuda.current_stream()
        overlap = getattr(self, "overlap_views", True) and len(timesteps) > 1
        views, joins, totals, ctxs, streams = [], [], [], [], []
        # Overlapped: ALL forwards are enqueued first, every view on its own stream, then every view's losses + backward.
        # A forward call returns when its pair count has arrived (its binning is done and its blend enqueued), so the
        # host hands view 1's binning over while view 0's forward blend runs, and view 0's loss kernels while view 1's
        # does: no stream waits for the ~0.2 ms of host work a view's calls take.  Every view reads its per-Gaussian colours
        # from the cache the previous step's Adam kernel left (without one, view k >= 1 evaluates its own: copying view 0's
        # would make its first kernel wait for view 0's whole forward).
        # Serial (overlap_views = False): one view after the other on the current stream, colours reused.
        step_begun = None
        if overlap:
            step_begun = torch.cuda.Event()
            step_begun.record(main)  # the parameters are final behind the previous Adam
        for k, ts in enumerate(timesteps):
            b = self._buffers(P, H, W, n_patches, dev, view=k)
            vstream = main if (k == 0 or not overlap) else self._view_stream(dev, k)
            if vstream is not main:
                vstream.wait_event(step_begun)
            views.append(b)
            streams.append(vstream)
            if overlap:
                with torch.cuda.stream(vstream):
                    ctxs.append(self._view_forward(b, ts, corners, dev, H, W, k, allow_reuse=(k == 0)))
        for k, ts in enumerate(timesteps):
            b, vstream, first = views[k], streams[k], k == 0
            with torch.cuda.stream(vstream):
                stream = _lib.current_stream()
                ctx = ctxs[k] if overlap else self._view_forward(b, ts, corners, dev, H, W, k)
                self._view_losses(b, ctx, H, W, n_patches)
                args, state, sbytes, cap, nr = ctx["args"], ctx["state"], ctx["sbytes"], ctx["cap"], ctx["nr"]
                if b.gc is None:
                    b.gc = torch.empty((P, 14), dtype=torch.float32, device=dev)
                # the densification statistic comes from view 0 only (train.py:260-263); the other views of the step raise
                # max_radii2D, as every render() does (gaussian_renderer/__init__.py:79) -- both inside the backward launch
                m2 = b.means2D_grad if (first and collect_stats) else None
                cfg = self._cfg_backward(b, cap=cap)
                tail, loss_k, _keep = self._step_tail(b, b.term_w, (True if first else "radii") if collect_stats else False)
                # one view per step and a producer-side reducer (N > 1): the per-Gaussian backward goes out in row
                # chunks, the all-reduce of each chunk starts while the next one is produced (dist.py)
                produce = (len(timesteps) == 1 and getattr(reduce_compact, "producer", False))
                if produce:
                    for ci, (lo, hi) in enumerate(reduce_compact.bounds(P)):
                        _lib.check(lib.fsgs_render_backward_compact_rows(
                            C.byref(cfg), P, C.byref(args), _lib.ptr(b.radii), _lib.ptr(state), sbytes, cap, nr,
                            _lib.ptr(b.d_image), _lib.ptr(b.d_depth_sil), _lib.ptr(b.gc),
                            None if m2 is None else _lib.ptr(m2), C.byref(tail), _lib.ptr(b.bwd_scratch),
                            b.bwd_scratch.numel(), lo, hi, int(ci == 0), stream), "fsgs_render_backward_compact_rows")
                        reduce_compact.produced(b.gc, lo, hi)
                    self._backward_done(b)  # (every row chunk went out: each one cleaned its own rows)
                else:
                    _lib.check(lib.fsgs_render_backward_compact(C.byref(cfg), P, C.byref(args), _lib.ptr(b.radii),
                                                                _lib.ptr(state), sbytes, cap, nr, _lib.ptr(b.d_image),
                                                                _lib.ptr(b.d_depth_sil), _lib.ptr(b.gc),
                                                                None if m2 is None else _lib.ptr(m2), C.byref(tail),
                                                                _lib.ptr(b.bwd_scratch), b.bwd_scratch.numel(), stream),
                               "fsgs_render_backward_compact")
                    self._backward_done(b)
                if vstream is not main:
                    done = torch.cuda.Event()
                    done.record()
                    joins.append(done)
            totals.append(loss_k)
        for e in joins:
            main.wait_event(e)
        b0 = views[0]
        # Adam reads at most two gradient buffers: further views (the reference never has more than two) are added up
        for extra in views[2:]:
            views[1].gc.add_(extra.gc)
        second = views[1].gc if len(views) > 1 else None
        if reduce_compact is not None and second is not None:  # an exchange moves ONE buffer
            b0.gc.add_(second)
            second = None
        adam = self._fused_adam_struct()

        def adam_rows(lo, hi, args=args, adam=adam):
            """Adam for Gaussians [lo, hi) from the (reduced) compact gradient"""
            a2, ad2 = self._offset_structs(args, adam, lo)
            with torch.cuda.device(dev):
                _lib.check(lib.fsgs_adam_step_compact_sum(hi - lo, C.byref(a2), b0.gc.data_ptr() + lo * 56,
                                                          None if second is None else second.data_ptr() + lo * 56,
                                                          C.byref(ad2), _lib.current_stream()),
                           "fsgs_adam_step_compact_sum")

        if reduce_compact is None:
            adam_rows(0, P)
        elif produce:
            reduce_compact.finish(adam_rows)  # Adam per chunk, each behind its own all-reduce
        elif getattr(reduce_compact, "pipelined", False):
            reduce_compact(b0.gc, adam_rows)  # chunked: all-reduce of chunk i+1 beside Adam of chunk i
        else:
            reduce_compact(b0.gc)  # ONE all-reduce of 56 B / Gaussian
            adam_rows(0, P)
        optim.mark_updated([pc.params[n_] for n_ in PARAM_NAMES])
        self._colors_cached(self._adam_next_colors)
        # the step's scalar loss (reporting only): summed BEHIND the Adam launch -- in front of it the little add kernel and its
        # dispatch gap sat on the step's critical path (two-view step: ~8 us)
        total = totals[0]
        for t_ in totals[1:]:
            total = total + t_
        self._compact_total = total
        return b0.radii, views[-1], state, cap

    # ---- tracking (train.py:166-200) -----------------------------------------------------------------------
    def tracking_step(self, t, targets, rigid_mask, want_losses=True):
        """One pose-tracking iteration of frame t.  want_losses=False skips the two launches that only form the weighted
        loss values for logging (returns (None, None, None)); the update itself is unaffected."""
        pc, lib, poses = self.pc, self.lib, self.poses
        dev = pc.params["_xyz"].device
        H, W = int(pc.cam.image_height), int(pc.cam.image_width)
        with torch.cuda.device(dev):
            fused_pose = self.fuse_pose and hasattr(poses, "fused_step") and isinstance(poses.optimizer, optim.FusedAdam)
            # fused: the pose the previous iteration's update kernel already produced (no launch); otherwise the
            # tiny quaternion -> matrix autograd graph for the 12-float chain rule
            w2c = poses.get_pose_detached(t) if fused_pose else poses.get_pose(t)
            with torch.no_grad():
                b = self._buffers(pc.num_points, H, W, int(P_CORR * (H // BOX) * (W // BOX)), dev)
                stream = _lib.current_stream()
                wd = w2c.detach().contiguous()
                # flow loss forward + backward in ONE pass, on a second stream (dflow = w_flow * dloss/dw2c).  It only needs the
                # pose, but it is ENQUEUED behind the render forward call: that call returns when the pair count arrives,
                # i.e. as the forward blend starts, and the bandwidth-bound flow kernels then share the GPU with the
                # issue-bound blend instead of with the latency-bound binning kernels (which they slowed by ~10 us)
                side = self._side_stream(dev)
                # the second stream needs the pose.  From the second iteration of a frame on that is the w2c the previous
                # iteration's pose update produced, and the update's launch carried an event (fsgs_pose_step_done_event):
                # wait for that -- everything the previous iteration enqueued on this stream lies in front of it -- instead
                # of recording a marker here, in front of this iteration's first kernel (~6 us)
                riding = self.__dict__.get("_pose_event_for")
                if fused_pose and riding is not None and riding[0] is w2c and riding[1] is targets:
                    _lib.check(lib.fsgs_stream_wait_event(C.c_void_p(side.cuda_stream), self._pose_event), "fsgs_stream_wait_event")
                else:
                    pose_ready = torch.cuda.Event()
                    pose_ready.record()
                    side.wait_event(pose_ready)
                self._pose_event_for = None
                M = int(targets.pts.shape[0])
                need = int(lib.fsgs_flow_scratch_bytes(M))
                if b.flow_scratch is None or b.flow_scratch.numel() < need:
                    b.flow_scratch = torch.empty((need,), dtype=torch.uint8, device=dev)
                args, state, sbytes, cap, nr = self._render_forward(wd, b, tracking=True)
                # (side stream by its raw handle, join through an event this buffer set keeps: see _view_losses)
                sstream = C.c_void_p(side.cuda_stream)
                _lib.check(lib.fsgs_flow_pose_loss_fused(M, _lib.ptr(targets.pts), _lib.ptr(targets.vu), _lib.ptr(wd),
                                                         targets.K9, _lib.ptr(targets.flow), W, H, 20.0,
                                                         float(LOSS_W_TRACKING["flow"]), 0.0,
                                                         _lib.ptr(b.flow_scratch), _lib.ptr(b.flow_out),
                                                         _lib.ptr(b.d_flow), sstream),
                           "fsgs_flow_pose_loss_fused")
                if getattr(b, "flow_event", None) is None:
                    ev = C.c_void_p()
                    _lib.check(lib.fsgs_event_create(C.byref(ev)), "fsgs_event_create")
                    b.flow_event = ev
                    weakref.finalize(b, _lib.load().fsgs_event_destroy, ev).atexit = False
                _lib.check(lib.fsgs_event_record(b.flow_event, sstream), "fsgs_event_record")
                wd.record_stream(side)
                # mask = [rendered depth > 0] * rigid mask (train.py:176-178): the presence test is evaluated inside the
                # loss kernels from the depth plane; the rigid mask (None = every pixel rigid) is handed over as floats,
                # converted once per mask object -- a frame's 50 iterations share it
                rigid_f = None
                if rigid_mask is not None:
                    hit = getattr(self, "_rigid_f", None)
                    if hit is None or hit[0] is not rigid_mask or hit[1] != rigid_mask._version:
                        hit = self._rigid_f = (rigid_mask, rigid_mask._version,
                                               rigid_mask.to(torch.float32).reshape(H, W).contiguous())
                    rigid_f = hit[2]
                presence = b.depth_sil[0]
                gt = self.frames.colors[t]
                _lib.check(lib.fsgs_photometric_loss_forward_backward(
                    3, H, W, _lib.ptr(b.image), _lib.ptr(gt), _lib.ptr(rigid_f), _lib.ptr(presence), 0.2,
                    _lib.ptr(b.maps), _lib.ptr(b.sums), _lib.ptr(b.rgb_out), None, _lib.ptr(b.d_image), stream),
                    "fsgs_photometric_loss_forward_backward")
                d_total = torch.empty((4, 4), dtype=torch.float32, device=dev)
                # (no dL/dmeans2D: with gs_grad = False the reference's viewspace_points carries no gradient either)
                grads = self._grad_struct([None] * 6, None, d_total)
                _lib.check(lib.fsgs_stream_wait_event(stream, b.flow_event), "fsgs_stream_wait_event")
                self._render_backward(args, state, sbytes, cap, nr, b, b.d_image, None, grads, False, True, False,
                                      zeroed=True)
                total = rgb = flow = None
                if want_losses:
                    weighted = b.terms * b.trk_w  # [w_rgb * rgb, ., ., ., ., w_flow * flow, ., .]
                    rgb, flow = weighted[0], weighted[5]
                    total = torch.dot(b.terms, b.trk_w)
                if fused_pose:
                    # scheduler first, as train.py:189,194; then ONE launch: w_rgb dW_rgb + dW_flow -> pose adjoint ->
                    # Adam -> the next iteration's w2c
                    poses.scheduler.step()
                    if self.__dict__.get("_pose_event") is None:
                        ev = C.c_void_p()
                        _lib.check(lib.fsgs_event_create(C.byref(ev)), "fsgs_event_create")
                        self._pose_event = ev
                        weakref.finalize(self, lib.fsgs_event_destroy, ev).atexit = False
                    _lib.check(lib.fsgs_pose_step_done_event(self._pose_event), "fsgs_pose_step_done_event")
                    poses.fused_step(t, d_total, float(LOSS_W_TRACKING["rgb"]), b.d_flow)
                    # (valid for the next call only if it tracks the same frame with the pose this update produced and
                    # the same flow targets, which were formed on this stream before)
                    self._pose_event_for = (poses.pred_w2c[int(t)], targets)
                    return total, rgb, flow
                # d_total = w_rgb * dL_rgb/dw2c + w_flow * dL_flow/dw2c
                if float(LOSS_W_TRACKING["rgb"]) != 1.0:
                    d_total.mul_(float(LOSS_W_TRACKING["rgb"]))
                d_total.add_(b.d_flow)
            w2c.backward(d_total)  # LearnPose.forward's backward: normalize + q2rot, 12 floats
            poses.scheduler.step()
            with torch.no_grad():
                poses.optimizer.step()
                poses.optimizer.zero_grad(set_to_none=True)
        return total, rgb, flow
