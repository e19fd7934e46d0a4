"""Sampson-distance rigid mask of the tracking step (train.py:157-165).

The reference computes, once per tracked frame t > 1, the squared Sampson distance of the dense matches given by
the forward flow of frame t-2 under the fundamental matrix of the (already optimised) poses t-2 and t-1
(PoseModel.compute_epipolar_loss / get_matches / get_fundamental_matrix, scene/pose_optimizer.py:640-648,
700-746) and thresholds it adaptively (utils/general_utils.py:96-116).  The three geometry helpers come from
kornia, which is not part of the reference tree; their public definitions are restated here
(essential_from_Rt, fundamental_from_essential, sampson_epipolar_distance with squared=True): parity unpinned.
"""
import ctypes as C

import numpy as np
import torch


def fundamental_from_w2c(w2c_1, w2c_2, K):
    """F = K^-T [t]x R K^-1 with the relative motion R = R2 R1^T, t = t2 - R t1 (both poses world-to-camera)."""
    A = np.asarray(w2c_1.detach().cpu() if torch.is_tensor(w2c_1) else w2c_1, dtype=np.float32)
    B = np.asarray(w2c_2.detach().cpu() if torch.is_tensor(w2c_2) else w2c_2, dtype=np.float32)
    Kn = np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float32)
    R1, t1, R2, t2 = A[:3, :3], A[:3, 3], B[:3, :3], B[:3, 3]
    R = R2 @ R1.T
    t = t2 - R @ t1
    Tx = np.array([[0.0, -t[2], t[1]], [t[2], 0.0, -t[0]], [-t[1], t[0], 0.0]], dtype=np.float32)
    E = Tx @ R
    Kinv = np.linalg.inv(Kn).astype(np.float32)
    return (Kinv.T @ E @ Kinv).astype(np.float32)


def sampson_distance_torch(flow_fw, F):
    """plain torch statement: flow_fw [2,H,W] (du, dv), F [3,3] -> squared Sampson distance [H,W]."""
    H, W = int(flow_fw.shape[1]), int(flow_fw.shape[2])
    dev = flow_fw.device
    Ft = torch.as_tensor(np.asarray(F), dtype=torch.float32, device=dev)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    x1 = torch.stack([xx.float(), yy.float(), torch.ones_like(xx, dtype=torch.float32)], dim=-1).reshape(-1, 3)
    x2 = x1.clone()
    x2[:, 0] += flow_fw[0].reshape(-1).float()
    x2[:, 1] += flow_fw[1].reshape(-1).float()
    l1 = x1 @ Ft.T   # F x1
    l2 = x2 @ Ft     # F^T x2
    num = (x2 * l1).sum(1) ** 2
    den = l1[:, 0] ** 2 + l1[:, 1] ** 2 + l2[:, 0] ** 2 + l2[:, 1] ** 2
    return (num / den).reshape(H, W)


def rigid_mask_torch(dist, factor=2.0):
    """`sampson_dist < adaptive_thresholding(sampson_dist)` exactly as written in the reference: the right-hand side is a
    BOOL mask (dist <= mean + factor * std), promoted to 0/1 by the comparison."""
    thr = dist.mean().item() + factor * dist.std().item()
    return dist < (dist <= thr)


def rigid_mask(flow_fw, F, factor=2.0):
    """product path (csrc/flow.hip): returns (rigid bool [H,W], dist float [H,W], stats {mean, std, threshold} on
    the device); two launches, no host synchronisation."""
    from . import _lib

    lib = _lib.load()
    if not flow_fw.is_cuda:
        raise RuntimeError("fsgs rigid mask needs CUDA/HIP tensors; there is no CPU fallback")
    fl = flow_fw.detach().contiguous().float()
    H, W = int(fl.shape[1]), int(fl.shape[2])
    dev = fl.device
    F9 = (C.c_float * 9)(*[float(v) for v in np.asarray(F, dtype=np.float32).reshape(9)])
    scratch = torch.empty((int(lib.fsgs_sampson_scratch_bytes(H, W)),), dtype=torch.uint8, device=dev)
    dist = torch.empty((H, W), dtype=torch.float32, device=dev)
    rigid = torch.empty((H, W), dtype=torch.uint8, device=dev)
    stats = torch.empty((3,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.fsgs_sampson_rigid_mask(H, W, _lib.ptr(fl), F9, float(factor), _lib.ptr(scratch), _lib.ptr(dist),
                                               _lib.ptr(rigid), _lib.ptr(stats), _lib.current_stream()),
                   "fsgs_sampson_rigid_mask")
    return rigid.bool(), dist, stats
