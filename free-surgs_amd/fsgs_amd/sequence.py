"""Synthetic SCARED-like video sequence rendered with the HIP path itself (no dataset ships, SURVEY.md
Appendix B): a hidden ground-truth Gaussian scene, a smooth ground-truth camera trajectory, and per frame
the colour image, a mono-depth map (affine-normalised to [0.5,1.5] like scene/pose_optimizer.py:406-407)
and the forward optical flow implied by depth + relative pose (convention of the reference's flow loss:
pixel index = K coordinate, scene/pose_optimizer.py:49-50,189-216)."""
import numpy as np
import torch

from . import synth
from .model import GaussianCloud
from .render import render
from .trainer import FrameData, PoseTrack, settings_from_cam


def gt_trajectory(n, step_t=0.004, step_r=0.002, seed=0):
    rng = np.random.default_rng(seed)
    poses = [np.eye(4)]
    v_t = np.array([step_t, -0.5 * step_t, 0.3 * step_t])
    v_r = np.array([0.5 * step_r, step_r, -0.3 * step_r])
    for i in range(1, n):
        v_t = v_t + 0.1 * step_t * rng.standard_normal(3)
        v_r = v_r + 0.1 * step_r * rng.standard_normal(3)
        q = np.array([1.0, *(0.5 * v_r * i)])
        poses.append(synth.pose_matrix(q, v_t * i))
    return poses


def make_sequence(W, H, n_frames, P, device="cuda", seed=0):
    """-> (frames: FrameData with gt_w2c, init_params for the learner's first-frame cloud)."""
    from simple_knn._C import distCUDA2

    knn = lambda pts: distCUDA2(torch.tensor(pts, device=device)).cpu().numpy()
    cam = synth.make_camera(W, H)
    sc = synth.init_scene(W, H, P, seed=seed, knn_fn=knn)
    # make the hidden scene opaque enough to be a well-defined surface
    sc = dict(sc)
    sc["_opacity"] = np.full_like(sc["_opacity"], 3.0)
    sc["_scaling"] = sc["_scaling"] + 0.35
    gt = GaussianCloud(sc, sh_degree=3, device=device)
    gt.cam = settings_from_cam(cam, device)
    w2cs = gt_trajectory(n_frames, seed=seed)
    poses = PoseTrack(n_frames, device)
    for i, m in enumerate(w2cs):
        q = _rot_to_quat(m[:3, :3])
        poses.set_pose(i, q, m[:3, 3])
    K = torch.tensor(cam["K"], dtype=torch.float32, device=device)
    colors, depths = [], []
    with torch.no_grad():
        for i in range(n_frames):
            pkg = render(poses, i, gt, gs_grad=False, cam_grad=False)
            colors.append(pkg["render"].clamp(0, 1).contiguous())
            depths.append(pkg["render_dep"].contiguous())
    monodeps = []
    for d in depths:
        lo, hi = d.min(), d.max()
        monodeps.append(((d - lo) / (hi - lo) + 0.5).contiguous())
    flows = []
    vv, uu = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    for i in range(n_frames - 1):
        d = depths[i]
        x = (uu - K[0, 2]) / K[0, 0] * d
        y = (vv - K[1, 2]) / K[1, 1] * d
        cam_i = torch.stack([x, y, d, torch.ones_like(d)], 0).reshape(4, -1)
        rel = torch.tensor(w2cs[i + 1] @ np.linalg.inv(w2cs[i]), dtype=torch.float32, device=device)
        cam_j = (rel @ cam_i)[:3]
        p = K @ cam_j
        u2, v2 = p[0] / (p[2] + 1e-5), p[1] / (p[2] + 1e-5)
        flows.append(torch.stack([u2.reshape(H, W) - uu, v2.reshape(H, W) - vv], 0).contiguous())
    frames = FrameData(colors, monodeps, flows_fw=flows, K=cam["K"], gt_w2c=[np.asarray(m, np.float32) for m in w2cs])
    frames.gt_depths = depths
    return frames, cam


def _rot_to_quat(R):
    """rotation matrix -> (r,x,y,z), positive scalar part."""
    t = np.trace(R)
    r = np.sqrt(max(0.0, 1.0 + t)) / 2.0
    x = (R[2, 1] - R[1, 2]) / (4 * r)
    y = (R[0, 2] - R[2, 0]) / (4 * r)
    z = (R[1, 0] - R[0, 1]) / (4 * r)
    return np.array([r, x, y, z])


def learner_from_first_frame(frames, cam, ratio=0.1, device="cuda", seed=0):
    """GaussianModel.initialize_first_timestep (scene/gaussian_model.py:237-258): random 10 % pixel mask,
    back-projection of the first mono-depth with the first colours, distCUDA2 scales."""
    from simple_knn._C import distCUDA2

    H, W = frames.colors[0].shape[-2:]
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = int(ratio * H * W)
    perm = torch.randperm(H * W, generator=g)[:n].sort().values.to(device)
    vv, uu = perm // W, perm % W
    K = cam["K"]
    depth = frames.monodeps[0]
    z = depth[vv, uu]
    x = (uu.float() - K[0, 2]) / K[0, 0] * z
    y = (vv.float() - K[1, 2]) / K[1, 1] * z
    xyz = torch.stack([x, y, z], 1).contiguous()
    rgb = frames.colors[0][:, vv, uu].T
    dist2 = torch.clamp_min(distCUDA2(xyz), 1e-7)
    P = xyz.shape[0]
    params = {
        "_xyz": xyz, "_features_dc": ((rgb - 0.5) / synth.SH_C0).reshape(P, 1, 3).contiguous(),
        "_features_rest": torch.zeros((P, 15, 3), device=device),
        "_opacity": torch.full((P, 1), float(np.log(0.1 / 0.9)), device=device),
        "_scaling": torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3),
        "_rotation": torch.tensor([[1.0, 0, 0, 0]], device=device).repeat(P, 1),
    }
    pc = GaussianCloud(params, sh_degree=3, device=device, scene_radius=float(depth.max()) / 2.0)
    pc.cam = settings_from_cam(cam, device)
    pc.training_setup()
    return pc


def write_frames(root, frames, scene="1", data="5"):
    """Lay a synthetic sequence down in the reference's on-disk layout (dataset.write_sequence):
    8-bit PNG colours, mono-depth as disparity (the reader inverts and re-normalises to [0.5,1.5]), forward flows,
    zero backward flows, the ground-truth poses and the intrinsics referred back to 1280x1024."""
    from . import dataset

    H, W = frames.colors[0].shape[-2:]
    u8 = [(c.detach().clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy() for c in frames.colors]
    disp = [(1.0 / m.detach()).cpu().numpy() for m in frames.monodeps]
    fw = [f.detach().cpu().numpy() for f in frames.flows_fw]
    KL = np.array(frames.K, dtype=np.float64)
    KL[0, :] *= dataset.REF_W / W
    KL[1, :] *= dataset.REF_H / H
    return dataset.write_sequence(root, u8, disp, fw, [np.zeros_like(f) for f in fw], frames.gt_w2c, KL,
                                  scene=scene, data=data)
