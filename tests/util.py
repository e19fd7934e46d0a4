"""Shared helpers of the parity tests."""
import numpy as np

from fsgs_amd import synth


def sh0_colors(params):
    """colors_precomp at SH degree 0: clamp_min(C0 * dc + 0.5, 0) (scene/gaussian_model.py:316-320)."""
    return np.clip(params["_features_dc"][:, 0, :] * synth.SH_C0 + 0.5, 0.0, None).astype(np.float32)


def to_camera_frame(xyz, w2c):
    """transform_to_frame (scene/pose_optimizer.py:960-989) on the host."""
    x = xyz.astype(np.float64)
    return (x @ w2c[:3, :3].T + w2c[:3, 3]).astype(np.float32)


def rel_err(a, b, floor):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def norm_err(a, b):
    """max abs error relative to the inf-norm of the reference tensor (SURVEY.md s8d thresholds)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def c1_poses():
    """8 poses for config C1: identity, the SURVEY perturbed pose, and 6 seeded small motions."""
    poses = [synth.pose_matrix(), synth.pose_matrix(**synth.PERTURBED_POSE)]
    rng = np.random.default_rng(123)
    for _ in range(6):
        q = np.array([1.0, 0, 0, 0]) + 0.03 * rng.standard_normal(4)
        t = 0.04 * rng.standard_normal(3)
        poses.append(synth.pose_matrix(q, t))
    return poses
