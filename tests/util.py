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


def assert_close_attributed(got, want, amp, what, tol=1e-4, floor=0.0, factor=2.0, max_outlier=5e-2):
    """Parity with flip ATTRIBUTION (replaces the blanket flip budget wherever the oracle can be asked which elements
    are fragile).  The rasteriser is discontinuous (alpha < 1/255 skips, alpha clamps at 0.99, T < 1e-4 stops,
    radius = ceil(3 sigma), tile rects truncate); two correct fp32 implementations resolve a near-tie differently for
    a handful of (pixel, Gaussian) pairs.  `amp` = how far the ORACLE's own element moves when every decision
    threshold is shifted by a rounding-sized hair either way (Oracle.flip_amplitudes; zero almost everywhere).
    Asserted, element by element:   |got - want| <= tol * ||want||_inf + factor * amp.
    So an element no near-tie reaches gets no allowance at all, and a fragile one only as much as a flip can
    actually move it -- an outlier without such a witness is a bug, however few there are.
    -> (elements beyond tol, elements with a non-zero allowance)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    amp = np.asarray(amp, np.float64)
    assert got.shape == want.shape == amp.shape, (what, got.shape, want.shape, amp.shape)
    if got.size == 0:
        return 0, 0
    scale = float(np.max(np.abs(want))) + floor + 1e-30
    err = np.abs(got - want)
    assert np.isfinite(err).all(), "%s: non-finite values" % what
    rogue = err > tol * scale + factor * amp
    if rogue.any():
        idx = np.argwhere(rogue)
        worst = idx[np.argmax(err[rogue])]
        raise AssertionError("%s: %d element(s) beyond %g of the inf-norm WITHOUT a near-tie witness; worst at %s: "
                             "got %g want %g (err %g of the norm, oracle flip amplitude %g); %d witnessed outliers" % (
                                 what, len(idx), tol, worst.tolist(), got[tuple(worst)], want[tuple(worst)],
                                 err[tuple(worst)] / scale, amp[tuple(worst)], int(((err > tol * scale) & ~rogue).sum())))
    assert err.max() <= max_outlier * scale, "%s: worst error %g of the inf-norm exceeds the flip bound %g" % (
        what, err.max() / scale, max_outlier)
    return int((err > tol * scale).sum()), int((amp > 0).sum())
