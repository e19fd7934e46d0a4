"""Optical-flow reprojection loss of the tracking step (scene/pose_optimizer.py:42-73,164-218).

The loss splits into a pose-INDEPENDENT part (back-projection of the previous frame's depth,
duplicate/origin rejection) that the reference recomputes in each of the 50 tracking iterations,
and a pose-dependent part (transform, project, border mask, L1 against the forward flow).
`FlowTargets` caches the first per frame; `flow_pose_loss` is the second.
"""
import numpy as np
import torch


def backproject_previous(depth_prev, K, w2c_prev, rigid_mask=None):
    """get_pointcloud over the valid pixels of the previous depth (scene/pose_optimizer.py:42-73,
    171-181).  Returns world points [M,3] and their pixel indices [M,2] = (v, u)."""
    depth = depth_prev.float()
    dm = depth[0]
    if rigid_mask is not None:
        dm = dm * rigid_mask
    idx = torch.stack(torch.where(dm > 0), dim=1)  # (v, u)
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    xx = (idx[:, 1] - cx) / fx
    yy = (idx[:, 0] - cy) / fy
    z = depth[0, idx[:, 0], idx[:, 1]]
    pts_cam = torch.stack((xx * z, yy * z, z), dim=-1)
    pts4 = torch.cat([pts_cam, torch.ones_like(pts_cam[:, :1])], dim=1)
    w2c_prev = torch.as_tensor(np.asarray(w2c_prev), dtype=torch.float32, device=depth.device) \
        if not torch.is_tensor(w2c_prev) else w2c_prev.float()
    pts = (torch.inverse(w2c_prev) @ pts4.T).T[:, :3]
    # drop rows whose |round(.,4)| equals another row's or the origin (scene/pose_optimizer.py:60-68)
    A = torch.abs(torch.round(pts, decimals=4)).float()
    B = torch.zeros((1, 3), dtype=A.dtype, device=A.device)
    _, inv, counts = torch.cat([A, B], dim=0).unique(dim=0, return_inverse=True, return_counts=True)
    dup = torch.isin(inv, torch.where(counts.gt(1))[0])[: len(A)]
    keep = ~dup
    return pts[keep], idx[keep]


def backproject_previous_hip(depth_prev, K, w2c_prev, rigid_mask=None):
    """product path of backproject_previous (csrc/flow.hip flow_targets_*): three launches around one sort of 64-bit
    hashes and one scan, instead of torch.unique(dim=0)'s lexicographic row sort; one host read (M)."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    depth = depth_prev.detach().float().contiguous()
    if not depth.is_cuda:
        raise RuntimeError("fsgs flow targets need CUDA/HIP tensors; there is no CPU fallback")
    dev = depth.device
    H, W = int(depth.shape[-2]), int(depth.shape[-1])
    HW = H * W
    w2c = torch.as_tensor(np.asarray(w2c_prev), dtype=torch.float32, device=dev) if not torch.is_tensor(w2c_prev) \
        else w2c_prev.detach().float().to(dev)
    c2w = torch.inverse(w2c).cpu().numpy().astype(np.float32).reshape(16)  # the reference inverts on the device too
    Kn = np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float32).reshape(9)
    K9 = (C.c_float * 9)(*[float(v) for v in Kn])
    M16 = (C.c_float * 16)(*[float(v) for v in c2w])
    rig = None if rigid_mask is None else rigid_mask.detach().to(torch.uint8).contiguous()
    world = torch.empty((HW, 3), dtype=torch.float32, device=dev)
    rounded = torch.empty((HW, 3), dtype=torch.float32, device=dev)
    keys = torch.empty((HW,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        stream = _lib.current_stream()
        _lib.check(lib.fsgs_flow_targets_keys(H, W, _lib.ptr(depth), _lib.ptr(rig), K9, M16, _lib.ptr(world),
                                              _lib.ptr(rounded), _lib.ptr(keys), stream), "fsgs_flow_targets_keys")
        skeys, sidx = torch.sort(keys)
        keep = torch.empty((HW,), dtype=torch.int32, device=dev)
        _lib.check(lib.fsgs_flow_targets_flag(HW, _lib.ptr(skeys), _lib.ptr(sidx), _lib.ptr(rounded), _lib.ptr(keep),
                                              stream), "fsgs_flow_targets_flag")
        incl = torch.cumsum(keep, dim=0, dtype=torch.int32)
        M = int(incl[-1].item())  # the one host read: the outputs' size
        pts = torch.empty((M, 3), dtype=torch.float32, device=dev)
        vu = torch.empty((M, 2), dtype=torch.int64, device=dev)
        if M > 0:
            _lib.check(lib.fsgs_flow_targets_gather(H, W, _lib.ptr(keep), _lib.ptr(incl), _lib.ptr(world), _lib.ptr(pts),
                                                    _lib.ptr(vu), stream), "fsgs_flow_targets_gather")
    return pts, vu


def flow_pose_loss_torch(pts_world, pix_vu, w2c_cur, K, flow_fw, width, height, edge=20):
    """transform by the current (differentiable) pose, project with K, keep the border-safe points in
    front of the camera, L1 between (projection - pixel) and the forward flow
    (scene/pose_optimizer.py:183-216).  flow_fw [2,H,W]: channel 0 = du, 1 = dv."""
    dev = pts_world.device
    pts4 = torch.cat([pts_world, torch.ones_like(pts_world[:, :1])], dim=1)
    cam = (w2c_cur @ pts4.T).T[:, :3]
    Kt = torch.as_tensor(np.asarray(K), dtype=torch.float32, device=dev) if not torch.is_tensor(K) else K.float()
    p = (Kt @ cam.T).T
    pz = p[:, 2:] + 1e-5
    p = p / pz
    uv = p[:, :2]
    m = (uv[:, 0] < width - edge) & (uv[:, 0] > edge) & (uv[:, 1] < height - edge) & (uv[:, 1] > edge) & (pz[:, 0] > 0)
    uv = uv[m]
    vu = pix_vu[m]
    if uv.numel() == 0 or vu.numel() == 0:
        return torch.tensor(0.0, device=dev)
    uv_src = vu[:, [1, 0]]
    proj_flow = uv - uv_src.float()
    gt = flow_fw[:, uv_src[:, 1], uv_src[:, 0]].permute(1, 0)
    if torch.isnan(proj_flow).any() or torch.isnan(gt).any():
        return torch.tensor(0.0, device=dev)
    return (proj_flow - gt).abs().mean()


def projection_flow_loss_torch(depth_prev, w2c_prev, w2c_cur, K, flow_fw, rigid_mask=None):
    """the whole reference function in one call (scene/pose_optimizer.py:164-218)."""
    H, W = depth_prev.shape[1], depth_prev.shape[2]
    pts, vu = backproject_previous(depth_prev, K, w2c_prev, rigid_mask)
    return flow_pose_loss_torch(pts, vu, w2c_cur, K, flow_fw.float(), W, H)


# ---- product path: per-frame cache + HIP kernels (csrc/flow.hip) ------------------------------------------
class FlowTargets:
    """The pose-independent half of projection_flow_loss, prepared ONCE per tracked frame (the reference
    redoes it in each of the 50 iterations, incl. a 1.3M-row unique): world points of the previous frame's
    valid pixels and their pixel indices, plus the forward flow, all resident on the device."""

    def __init__(self, depth_prev, w2c_prev, K, flow_fw, rigid_mask=None):
        self.H, self.W = int(depth_prev.shape[1]), int(depth_prev.shape[2])
        if depth_prev.is_cuda:
            pts, vu = backproject_previous_hip(depth_prev, K, w2c_prev, rigid_mask)
        else:
            pts, vu = backproject_previous(depth_prev, K, w2c_prev, rigid_mask)  # CPU: the torch statement
        self.pts = pts.detach().contiguous().float()
        self.vu = vu.detach().contiguous().to(torch.int64)
        self.flow = flow_fw.detach().contiguous().float()
        Kn = np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float32).reshape(9)
        import ctypes as C

        self.K9 = (C.c_float * 9)(*[float(v) for v in Kn])


class _FlowPoseLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w2c, targets, edge):
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        if not w2c.is_cuda:
            raise RuntimeError("fsgs flow loss needs CUDA/HIP tensors; there is no CPU fallback")
        w = w2c.detach().contiguous().float()
        acc = torch.empty((3,), dtype=torch.float64, device=w.device)
        out = torch.empty((2,), dtype=torch.float32, device=w.device)
        M = int(targets.pts.shape[0])
        with torch.cuda.device(w.device):
            rc = lib.fsgs_flow_pose_loss_forward(M, _lib.ptr(targets.pts), _lib.ptr(targets.vu), _lib.ptr(w), targets.K9,
                                                 _lib.ptr(targets.flow), targets.W, targets.H, float(edge),
                                                 _lib.ptr(acc), _lib.ptr(out), _lib.current_stream())
        _lib.check(rc, "fsgs_flow_pose_loss_forward")
        ctx.save_for_backward(w, acc)
        ctx.targets, ctx.edge = targets, float(edge)
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        from . import _lib

        lib = _lib.load()
        w, acc = ctx.saved_tensors
        t = ctx.targets
        up = grad_out.detach().contiguous().float().reshape(1)
        dw = torch.empty((4, 4), dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            rc = lib.fsgs_flow_pose_loss_backward(int(t.pts.shape[0]), _lib.ptr(t.pts), _lib.ptr(t.vu), _lib.ptr(w), t.K9,
                                                  _lib.ptr(t.flow), t.W, t.H, ctx.edge, _lib.ptr(acc), _lib.ptr(up),
                                                  _lib.ptr(dw), _lib.current_stream())
        _lib.check(rc, "fsgs_flow_pose_loss_backward")
        return dw, None, None


def flow_pose_loss(w2c_cur, targets, edge=20):
    """projection_flow_loss given the cached per-frame targets (scene/pose_optimizer.py:183-216)."""
    return _FlowPoseLoss.apply(w2c_cur, targets, edge)


def projection_flow_loss(depth_prev, w2c_prev, w2c_cur, K, flow_fw, rigid_mask=None):
    """Same signature role as the reference function (one-shot: builds the targets, then the HIP loss)."""
    return flow_pose_loss(w2c_cur, FlowTargets(depth_prev, w2c_prev, K, flow_fw, rigid_mask))
