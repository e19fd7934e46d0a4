"""The on-disk sequence layout the reference's caller consumes (PoseModel.__init__,
scene/pose_optimizer.py:355-460) -> the HBM-resident FrameData the step drivers work on, and the writer that lays a
(synthetic) sequence down in that layout.

    <src>/input/<scene>_<data>_<x>_<img>.png|.jpeg|.jpg   RGB frames, sorted by file name
    <src>/poses/<scene>_<data>/frame_<img>.json           "camera-pose" 4x4, "camera-calibration"."KL" 3x3 @1280x1024
    <src>/flow/flow_fw_<name>.npz, flow_bw_<name>.npz     'pred' [1,2,H,W], frame i -> i+1 (none for the last frame)
    <src>/monodep/depth_<name>.npz                        'pred' [H,W] disparity; depth = 1/pred, min-max -> [0.5,1.5]

Everything a frame needs is decoded ONCE and kept on the device (s8f #4: no per-iteration H2D / D2H)."""
import glob
import json
import os

import numpy as np
import torch

from .trainer import FrameData

REF_W, REF_H = 1280, 1024  # resolution the stored calibration refers to (scene/pose_optimizer.py:413-414)


def _frame_keys(path):
    """<scene>_<data>_<x>_<img>.<ext> -> (scene, data, img, name-without-extension) (:381-384,395)."""
    name = os.path.basename(path)
    parts = name.split("_")
    if len(parts) < 4:
        raise ValueError("frame name %r does not follow <scene>_<data>_<x>_<img>.<ext>" % name)
    return parts[0], parts[1], parts[3].split(".")[0], name.split(".")[0]


def list_frames(source_path, frame_start=0, frame_end=-1):
    """sorted RGB paths; frame_end == -1 keeps all of them, otherwise the [frame_start:frame_end] slice (:360-372)."""
    paths = []
    for ext in ("png", "jpeg", "jpg"):
        paths += glob.glob(os.path.join(source_path, "input", "*." + ext))
    if not paths:
        raise FileNotFoundError("no frames under %s" % os.path.join(source_path, "input"))
    paths = sorted(paths)
    if frame_end != -1:
        paths = paths[frame_start:frame_end]
    return paths


def _read_image(path):
    from PIL import Image

    a = np.asarray(Image.open(path))
    a = a[..., None] if a.ndim == 2 else a
    return np.ascontiguousarray(np.transpose(a, (2, 0, 1))).astype(np.float32) / 255.0  # PILtoTorch, CHW in [0,1]


def normalise_monodepth(disparity):
    """depth = 1 / disparity, affinely mapped to [0.5, 1.5] per frame (:406-407); float64 like the numpy statement."""
    d = 1.0 / np.asarray(disparity)
    return (d - d.min()) / (d.max() - d.min()) * 1.0 + 0.5


def read_sequence(source_path, frame_start=0, frame_end=-1, device="cuda", sample_rate=8, staged_capacity=None):
    """-> FrameData with colours / mono-depths / forward flows on `device` (with `staged_capacity` = n: a
    staging.StagedFrames instead -- the inputs in pinned host memory, n frames per lane resident on the device, the next
    frame copied while the current one is optimised; for sequences beyond HBM), the rescaled intrinsics, the ground-truth
    "camera-pose" matrices (grouped per <data> run: data_ind offsets and weights as eval_pose uses them,
    train.py:492-506), backward flows, image names and the i_train / i_test split."""
    paths = list_frames(source_path, frame_start, frame_end)
    n = len(paths)
    colors, monodeps, flows_fw, flows_bw, gt, runs = [], [], [], [], [], {}
    KL = None
    for i, p in enumerate(paths):
        scene, data, img, name = _frame_keys(p)
        with open(os.path.join(source_path, "poses", "%s_%s" % (scene, data), "frame_%s.json" % img)) as f:
            meta = json.load(f)
        pose = np.array(meta["camera-pose"], dtype=np.float64)
        KL = np.array(meta["camera-calibration"]["KL"], dtype=np.float64)  # the last frame's wins, as upstream
        runs.setdefault(data, []).append(pose)
        gt.append(pose.astype(np.float32))
        if i < n - 1:
            flows_fw.append(np.load(os.path.join(source_path, "flow", "flow_fw_%s.npz" % name))["pred"])
            flows_bw.append(np.load(os.path.join(source_path, "flow", "flow_bw_%s.npz" % name))["pred"])
        monodeps.append(normalise_monodepth(np.load(os.path.join(source_path, "monodep", "depth_%s.npz" % name))["pred"]))
        colors.append(_read_image(p))
    C, H, W = colors[0].shape
    K = KL.copy()
    K[0, :] *= W / REF_W
    K[1, :] *= H / REF_H
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).contiguous()
    fw = [(t if staged_capacity is None else np.ascontiguousarray)(f.reshape(-1, 2, H, W)[0]) for f in flows_fw]
    if staged_capacity is None:
        frames = FrameData([t(c) for c in colors], [t(m) for m in monodeps], flows_fw=fw, K=K.astype(np.float32), gt_w2c=gt)
    else:
        from .staging import StagedFrames

        frames = StagedFrames(colors, monodeps, flows_fw=fw, K=K.astype(np.float32), gt_w2c=gt, device=dev,
                              capacity=staged_capacity)
    if sample_rate != 8:
        idx = np.arange(n)
        frames.i_test = idx[int(sample_rate / 2)::sample_rate]
        frames.i_train = np.array([i for i in idx if i not in frames.i_test])
    frames.flows_bw = [t(f.reshape(-1, 2, H, W)[0]) for f in flows_bw]
    frames.image_names = paths
    frames.scene = "scared_%s" % _frame_keys(paths[-1])[0]
    frames.data_ind = [0]
    frames.weights = []
    for key, value in runs.items():  # insertion order = first appearance, as the reference's dict
        frames.data_ind.append(frames.data_ind[-1] + len(value))
        frames.weights.append(len(value) / n * 1.0)
    frames.gt_poses = {k: np.stack(v).astype(np.float32) for k, v in runs.items()}
    frames.W, frames.H = W, H
    return frames


def camera_from_frames(frames, znear=0.01, zfar=100.0):
    """the static camera dict synth.make_camera returns, from a loaded sequence's intrinsics
    (scene/pose_optimizer.py:421-426,600-633)."""
    from . import synth

    return synth.make_camera(frames.W, frames.H, K=np.asarray(frames.K, np.float64), near=znear, far=zfar)


def write_sequence(root, colors_u8, disparity, flows_fw, flows_bw, camera_poses, KL, scene="1", data="5", tag="left",
                   first_index=0, ext="png"):
    """Lay n frames down in the layout above.  colors_u8 [n,H,W,3] uint8, disparity [n,H,W] (what the mono-depth
    network emits: 1 / depth up to an affine map), flows_* [n-1,2,H,W], camera_poses [n,4,4], KL [3,3] at 1280x1024.
    `data` may be a list (one entry per frame) to write several runs."""
    from PIL import Image

    n = len(colors_u8)
    datas = list(data) if isinstance(data, (list, tuple)) else [data] * n
    for sub in ("input", "flow", "monodep"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    names = []
    for i in range(n):
        img = "%06d" % (first_index + i)
        name = "%s_%s_%s_%s" % (scene, datas[i], tag, img)
        names.append(name)
        Image.fromarray(np.asarray(colors_u8[i], np.uint8)).save(os.path.join(root, "input", name + "." + ext))
        pdir = os.path.join(root, "poses", "%s_%s" % (scene, datas[i]))
        os.makedirs(pdir, exist_ok=True)
        with open(os.path.join(pdir, "frame_%s.json" % img), "w") as f:
            json.dump({"camera-pose": np.asarray(camera_poses[i], np.float64).tolist(),
                       "camera-calibration": {"KL": np.asarray(KL, np.float64).tolist()}}, f)
        np.savez(os.path.join(root, "monodep", "depth_%s.npz" % name), pred=np.asarray(disparity[i], np.float32))
        if i < n - 1:
            np.savez(os.path.join(root, "flow", "flow_fw_%s.npz" % name), pred=np.asarray(flows_fw[i], np.float32)[None])
            np.savez(os.path.join(root, "flow", "flow_bw_%s.npz" % name), pred=np.asarray(flows_bw[i], np.float32)[None])
    return names
