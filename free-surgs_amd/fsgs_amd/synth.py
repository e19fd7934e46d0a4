"""Synthetic SCARED-like cameras and Gaussian scenes (no dataset ships).

Host-side numpy only.  The scene is built the way the reference initialises it:
  * intrinsics scaled like the loader does   (scene/pose_optimizer.py:413-414)
  * raster settings like PoseModel.setup_camera (scene/pose_optimizer.py:600-633)
  * first-frame cloud: random pixel mask + back-projection of a mono-depth map
    normalised to [0.5, 1.5]                   (scene/gaussian_model.py:237-258,
    utils/geometry_utils.py:276-331, scene/pose_optimizer.py:406-407)
  * scales = log sqrt(max(knn_meandist2, 1e-7)), rot = (1,0,0,0),
    opacity = logit(0.1), colours -> RGB2SH     (scene/gaussian_model.py:335-357)
and a "trained-like" variant (anisotropic, rotated, mixed opacity, SH rest != 0).
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814  # utils/sh_utils.py:24


def intrinsics(W, H):
    """SCARED-like K at 1280x1024, rescaled by W/1280, H/1024 (scene/pose_optimizer.py:413-414)."""
    fx = 1035.0 * W / 1280.0
    fy = 1035.0 * H / 1024.0
    cx = 596.5 * W / 1280.0
    cy = 520.5 * H / 1024.0
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)


def make_camera(W, H, w2c=None, K=None, near=0.01, far=100.0, bg=(1.0, 1.0, 1.0)):
    """Raster settings exactly as PoseModel.setup_camera builds them
    (scene/pose_optimizer.py:600-633): matrices in TRANSPOSED storage."""
    if w2c is None:
        w2c = np.eye(4)
    w2c = np.asarray(w2c, dtype=np.float64)
    if K is None:
        K = intrinsics(W, H)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    opengl_proj = np.array(
        [
            [2 * fx / W, 0.0, -(W - 2 * cx) / W, 0.0],
            [0.0, 2 * fy / H, -(H - 2 * cy) / H, 0.0],
            [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
            [0.0, 0.0, 1.0, 0.0],
        ]
    )
    # the reference forms the product in fp32 (`w2c.bmm(opengl_proj)` on .float() tensors, :617-618): fp32 operands
    # and a k-ordered fp32 accumulation reproduce its matrix bit for bit (tests/golden/camera.npz)
    view_t = w2c.T.astype(np.float32)
    proj_t = opengl_proj.T.astype(np.float32)
    full_proj = np.zeros((4, 4), np.float32)
    for k in range(4):
        full_proj = full_proj + np.outer(view_t[:, k], proj_t[k, :])
    return {
        "image_height": int(H),
        "image_width": int(W),
        "tanfovx": W / (2 * fx),
        "tanfovy": H / (2 * fy),
        "bg": np.asarray(bg, dtype=np.float32),
        "scale_modifier": 1.0,
        "viewmatrix": view_t,
        "projmatrix": full_proj,
        "sh_degree": 0,
        "campos": np.linalg.inv(w2c)[:3, 3].astype(np.float32),
        "prefiltered": False,
        "debug": False,
        "K": K,
    }


def quat_to_rot(q):
    """(r,x,y,z) -> R, normalising first (scene/pose_optimizer.py:840-860)."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q)
    r, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
            [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
            [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def pose_matrix(q=(1.0, 0.0, 0.0, 0.0), t=(0.0, 0.0, 0.0)):
    m = np.eye(4)
    m[:3, :3] = quat_to_rot(q)
    m[:3, 3] = np.asarray(t, dtype=np.float64)
    return m


PERTURBED_POSE = dict(q=(1.0, 0.01, -0.02, 0.015), t=(0.02, -0.01, 0.03))


def synth_depth(W, H, rng, noise=0.002):
    """Smooth mono-depth in [0.5,1.5]: 1 + 0.4 sin(2 pi u/W) cos(2 pi v/H) + noise*N(0,1), clipped."""
    u = np.arange(W, dtype=np.float64)[None, :]
    v = np.arange(H, dtype=np.float64)[:, None]
    d = 1.0 + 0.4 * np.sin(2 * math.pi * u / W) * np.cos(2 * math.pi * v / H)
    d = d + noise * rng.standard_normal((H, W))
    return np.clip(d, 0.5, 1.5)


def synth_image(W, H, rng):
    """Smooth colourful RGB image in [0,1], [3,H,W]."""
    u = np.arange(W, dtype=np.float64)[None, :] / W
    v = np.arange(H, dtype=np.float64)[:, None] / H
    img = np.stack(
        [
            0.5 + 0.4 * np.sin(6.0 * u + 2.0 * v),
            0.5 + 0.4 * np.cos(4.0 * v - 3.0 * u),
            0.5 + 0.4 * np.sin(5.0 * (u + v)),
        ]
    )
    img = img + 0.03 * rng.standard_normal(img.shape)
    return np.clip(img, 0.0, 1.0)


def _knn_meandist2_cpu(pts):
    """Exact 3-NN mean squared distance on the host (scipy cKDTree); only used to BUILD scenes."""
    from scipy.spatial import cKDTree

    tree = cKDTree(pts)
    d, _ = tree.query(pts, k=4)
    return (d[:, 1:] ** 2).mean(axis=1)


def init_scene(W, H, P, seed=0, knn_fn=None, max_sh_degree=3):
    """First-frame initialisation as GaussianModel.initialize_first_timestep does it.
    Returns dict of RAW parameters in the GaussianModel.params layout (float32)."""
    rng = np.random.default_rng(seed)
    K = intrinsics(W, H)
    depth = synth_depth(W, H, rng)
    img = synth_image(W, H, rng)
    perm = rng.permutation(W * H)[:P]
    perm.sort()
    vv, uu = np.divmod(perm, W)
    z = depth[vv, uu]
    x = (uu - K[0, 2]) / K[0, 0] * z
    y = (vv - K[1, 2]) / K[1, 1] * z
    xyz = np.stack([x, y, z], axis=1)
    rgb = img[:, vv, uu].T
    knn = (knn_fn or _knn_meandist2_cpu)(xyz.astype(np.float32))
    knn = np.maximum(np.asarray(knn, dtype=np.float64), 1e-7)
    scaling = np.repeat(np.log(np.sqrt(knn))[:, None], 3, axis=1)
    nsh = (max_sh_degree + 1) ** 2
    f_dc = ((rgb - 0.5) / SH_C0)[:, None, :]
    f_rest = np.zeros((P, nsh - 1, 3))
    rot = np.zeros((P, 4))
    rot[:, 0] = 1.0
    opacity = np.full((P, 1), math.log(0.1 / 0.9))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {
        "_xyz": f32(xyz),
        "_features_dc": f32(f_dc),
        "_features_rest": f32(f_rest),
        "_opacity": f32(opacity),
        "_scaling": f32(scaling),
        "_rotation": f32(rot),
        "depth_map": f32(depth),
        "image": f32(img),
        "K": K,
    }


def trained_like_scene(W, H, P, seed=0, knn_fn=None, max_sh_degree=3, base_ratio=0.1):
    """C2/C4 stress variant: the init cloud replicated k times with jitter,
    log-normal anisotropic scales, random rotations, mixed opacities, non-zero SH rest."""
    rng = np.random.default_rng(seed + 1)
    P0 = min(P, max(1, int(base_ratio * W * H)))
    base = init_scene(W, H, P0, seed=seed, knn_fn=knn_fn, max_sh_degree=max_sh_degree)
    reps = int(math.ceil(P / P0))
    idx = np.tile(np.arange(P0), reps)[:P]
    spacing = np.exp(base["_scaling"][idx, :1].astype(np.float64))
    xyz = base["_xyz"][idx].astype(np.float64) + 0.5 * spacing * rng.standard_normal((P, 3))
    scaling = base["_scaling"][idx].astype(np.float64) - 0.5 * math.log(max(reps, 1)) / 1.0
    scaling = scaling + 0.5 * rng.standard_normal((P, 3))
    rot = rng.standard_normal((P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opacity = 1.5 * rng.standard_normal((P, 1))
    nsh = (max_sh_degree + 1) ** 2
    f_dc = base["_features_dc"][idx].astype(np.float64) + 0.1 * rng.standard_normal((P, 1, 3))
    f_rest = 0.05 * rng.standard_normal((P, nsh - 1, 3))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    out = dict(base)
    out.update(
        {
            "_xyz": f32(xyz),
            "_features_dc": f32(f_dc),
            "_features_rest": f32(f_rest),
            "_opacity": f32(opacity),
            "_scaling": f32(scaling),
            "_rotation": f32(rot),
        }
    )
    return out


def activate(params):
    """Host mirror of the GaussianModel activations (scene/gaussian_model.py:38-46,118-138)."""
    scales = np.exp(params["_scaling"].astype(np.float64))
    q = params["_rotation"].astype(np.float64)
    rot = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    opac = 1.0 / (1.0 + np.exp(-params["_opacity"].astype(np.float64)))
    return scales.astype(np.float32), rot.astype(np.float32), opac.astype(np.float32)


def random_small_scene(P, cam, seed=0, zmin=0.6, zmax=1.6, scale_px=(1.5, 6.0), channels=3):
    """Small random cloud in front of `cam` for oracle self-tests / finite differences."""
    rng = np.random.default_rng(seed)
    W, H = cam["image_width"], cam["image_height"]
    K = cam["K"]
    z = rng.uniform(zmin, zmax, P)
    u = rng.uniform(-0.1 * W, 1.1 * W, P)
    v = rng.uniform(-0.1 * H, 1.1 * H, P)
    xyz = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
    spx = rng.uniform(scale_px[0], scale_px[1], (P, 3))
    scales = spx * (z / K[0, 0])[:, None]
    rot = rng.standard_normal((P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, P)
    colors = rng.uniform(0.0, 1.0, (P, channels))
    return xyz, colors, opac, scales, rot
