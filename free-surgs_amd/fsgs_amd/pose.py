"""Camera-pose glue on the hot path (reference: scene/pose_optimizer.py:822-877, 960-989)."""
import torch
import torch.nn.functional as F


def quat_to_rotmat(q):
    """q [4] = (r,x,y,z), normalised inside like LearnPose.q2rot (scene/pose_optimizer.py:840-860)."""
    q = q / torch.sqrt((q * q).sum())
    r, x, y, z = q[0], q[1], q[2], q[3]
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)]),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)]),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]),
    ])


def pose_to_w2c(r_param, t_param, cam_id):
    """LearnPose.forward: r_param [1,4,N], t_param [3,N] -> w2c [4,4] (scene/pose_optimizer.py:822-877).
    The quaternion is normalised twice in the reference (F.normalize then q2rot); both are kept."""
    q = F.normalize(r_param[..., cam_id])[0]
    t = t_param[..., cam_id]
    R = quat_to_rotmat(q)
    top = torch.cat([R, t.reshape(3, 1)], dim=1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=top.dtype, device=top.device)
    return torch.cat([top, bottom], dim=0)


def transform_to_frame(means3D, w2c, gaussians_grad=True, camera_grad=True):
    """x_cam = (w2c @ [x;1])[:3] with either operand detached (scene/pose_optimizer.py:960-989)."""
    m = w2c if camera_grad else w2c.detach()
    p = means3D if gaussians_grad else means3D.detach()
    return p @ m[:3, :3].T + m[:3, 3]


class _PoseToW2C(torch.autograd.Function):
    """LearnPose.forward as ONE launch forward and ONE backward (csrc/pose.hip)."""

    @staticmethod
    def forward(ctx, r_param, t_param, cam_id):
        from . import _lib

        lib = _lib.load()
        if not r_param.is_cuda:
            raise RuntimeError("fsgs pose op needs CUDA/HIP tensors; there is no CPU fallback")
        r = r_param.detach().contiguous().float()
        t = t_param.detach().contiguous().float()
        N = int(r.shape[-1])
        w2c = torch.empty((4, 4), dtype=torch.float32, device=r.device)
        with torch.cuda.device(r.device):
            _lib.check(lib.fsgs_pose_forward(_lib.ptr(r), _lib.ptr(t), N, int(cam_id), _lib.ptr(w2c),
                                             _lib.current_stream()), "fsgs_pose_forward")
        ctx.save_for_backward(r)
        ctx.meta = (N, int(cam_id), tuple(r_param.shape), tuple(t_param.shape))
        return w2c

    @staticmethod
    def backward(ctx, dw):
        from . import _lib

        lib = _lib.load()
        (r,) = ctx.saved_tensors
        N, cam_id, rs, ts = ctx.meta
        g = dw.detach().contiguous().float()
        dr = torch.empty(rs, dtype=torch.float32, device=r.device)
        dt = torch.empty(ts, dtype=torch.float32, device=r.device)
        with torch.cuda.device(r.device):
            _lib.check(lib.fsgs_pose_backward(_lib.ptr(r), N, cam_id, _lib.ptr(g), _lib.ptr(dr), _lib.ptr(dt),
                                              _lib.current_stream()), "fsgs_pose_backward")
        return dr, dt, None


def pose_to_w2c_hip(r_param, t_param, cam_id):
    return _PoseToW2C.apply(r_param, t_param, cam_id)
