"""Fused optimiser pieces: `FusedAdam` (drop-in for the torch.optim.Adam instances the reference
builds at scene/gaussian_model.py:372-378,408 and scene/pose_optimizer.py:489-493 -- same param_groups
and state layout, so densification's state surgery keeps working) and the densification statistics."""
import ctypes as C

import torch

from . import _lib


def mark_updated(tensors):
    """The HIP kernels write parameters behind autograd's back; bump the version counters like an in-place torch op
    would (no kernel is launched), so version-keyed caches and autograd's saved-tensor checks stay truthful."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (betas, eps, no weight decay / amsgrad / maximize) with every group
    updated by ONE HIP kernel launch.  state[p] = {'step' (python int), 'exp_avg', 'exp_avg_sq'}."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        lib = _lib.load()
        by_hyper = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam needs CUDA/HIP tensors; there is no CPU fallback")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (group["betas"][0], group["betas"][1], group["eps"], p.device)
                by_hyper.setdefault(key, []).append((p, g, st, float(group["lr"])))
        for (b1, b2, eps, dev), items in by_hyper.items():
            with torch.cuda.device(dev):
                stream = _lib.current_stream()
                for i in range(0, len(items), 8):
                    chunk = items[i:i + 8]
                    arr = (_lib.FsgsAdamGroup * len(chunk))()
                    for k, (p, g, st, lr) in enumerate(chunk):
                        arr[k].param, arr[k].grad = p.data_ptr(), g.data_ptr()
                        arr[k].exp_avg, arr[k].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                        arr[k].n, arr[k].lr, arr[k].step = p.numel(), lr, st["step"]
                    _lib.check(lib.fsgs_adam_step(len(chunk), arr, b1, b2, eps, stream), "fsgs_adam_step")
                mark_updated([it[0] for it in items])
        return None


def densify_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    """add_densification_stats + max_radii2D update in one launch (scene/gaussian_model.py:678-681)."""
    lib = _lib.load()
    P = int(radii.shape[0])
    g = viewspace_grad.detach().contiguous()
    # raw pointers cross the C ABI: the kernel reads int32 radii and fp32 [P,3] / [P,1] / [P] rows, nothing else
    assert radii.dtype == torch.int32 and radii.is_contiguous() and radii.is_cuda, "radii: contiguous int32 on the GPU"
    for name, t, n in (("viewspace_grad", g, 3 * P), ("max_radii2D", max_radii2D, P),
                       ("xyz_gradient_accum", xyz_gradient_accum, P), ("denom", denom, P)):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n and t.device == radii.device, \
            "%s: contiguous fp32 with %d elements on %s" % (name, n, radii.device)
    with torch.cuda.device(radii.device):
        rc = lib.fsgs_densify_stats(P, _lib.ptr(radii), _lib.ptr(g), _lib.ptr(max_radii2D),
                                    _lib.ptr(xyz_gradient_accum), _lib.ptr(denom), _lib.current_stream())
    _lib.check(rc, "fsgs_densify_stats")
