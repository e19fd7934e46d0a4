"""Shared helpers of the parity tests."""
import collections

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


# outliers .. neg: see assert_close_attributed; max_err / p9999 / max_err_plain: the ACHIEVED error of the tensor relative to
# the scale the tolerance is taken of (max |want| + floor) -- over all elements, its 99.99th percentile, and over the elements
# that carry no allowance reaching the tolerance (what the plain 1e-4 criterion alone is up against); scale itself last
# max_err_zero_amp: over the elements NO near-tie reaches at all (allowance exactly zero; None when there are none) and
# zero_amp_fraction, their share of the tensor
Attribution = collections.namedtuple(
    "Attribution", "outliers fragile size pos neg max_err p9999 max_err_plain scale max_err_zero_amp zero_amp_fraction",
    defaults=(0.0, 0.0, 0.0, 0.0, None, 0.0))
# every call of assert_close_attributed leaves a record here (what, the counts above, the caller's tag); the sweep-level
# statistics tests read it (fraction of witnessed outliers, sign balance of got - want over them), and
# dump_attribution_log() writes it out next to the profiles
ATTRIBUTION_LOG = []
# ... and one with the achieved error of EVERY comparison, passed or not (max / 99.99th percentile / scale)
ACHIEVED_LOG = []
# Bounds on the WITNESSED outliers of one tensor.  Same inputs on both sides (the operator-boundary rasteriser against the
# oracle): 1e-4 of the elements -- measured 8e-7 (image) ... 1.2e-5 (gradients) at C2 / C4, 5e-5 at C1.  One flipped
# (pixel, Gaussian) pair moves every gradient component of that Gaussian and of the ones behind it at that pixel, so
# small tensors get a floor of 64 elements (a 650-Gaussian cloud crammed into a 24 x 54 image: 36 of 1944).  The end-to-end render comparisons (fused HIP glue against torch CPU glue)
# pass RENDER_OUTLIER_FRACTION instead: the two sides compute the view-space depth with differently rounded arithmetic,
# so list neighbours within a few ulp of depth swap places (Oracle.find_order_ties) -- ~0.6 such pairs per tile at C2,
# each visible wherever both Gaussians overlap: 2.1e-4 of the image at C2, 3.8e-4 at C4 (profiles/r03_full_size_parity.jsonl).
MAX_OUTLIER_FRACTION = 1e-4
RENDER_OUTLIER_FRACTION = 1e-3
MIN_OUTLIER_COUNT = 64


def assert_close_attributed(got, want, amp, what, tol=1e-4, floor=0.0, factor=2.0, max_outlier=5e-2, max_fraction=None,
                            tag=None):
    """Parity with flip ATTRIBUTION (replaces the blanket flip budget wherever the oracle can be asked which elements
    are fragile).  The rasteriser is discontinuous (alpha < 1/255 skips, alpha clamps at 0.99, T < 1e-4 stops,
    radius = ceil(3 sigma), tile rects truncate); two correct fp32 implementations resolve a near-tie differently for
    a handful of (pixel, Gaussian) pairs.  `amp` = how far the ORACLE's own element moves when every decision
    threshold is shifted by a rounding-sized hair either way (Oracle.flip_amplitudes; zero almost everywhere).
    Asserted, element by element:   |got - want| <= tol * ||want||_inf + factor * amp.
    So an element no near-tie reaches gets no allowance at all, and a fragile one only as much as a flip can
    actually move it -- an outlier without such a witness is a bug, however few there are.  The WITNESSED outliers are
    counted and bounded too (max_fraction of the tensor's elements, default MAX_OUTLIER_FRACTION, with a floor of
    MIN_OUTLIER_COUNT for small tensors): a path that lost every near-tie would still be "witnessed" element by
    element, but not in a handful of places.
    -> Attribution(elements beyond tol, elements whose allowance reaches the plain tolerance, size, outliers with got > want,
    < want)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    amp = np.asarray(amp, np.float64)
    assert got.shape == want.shape == amp.shape, (what, got.shape, want.shape, amp.shape)
    if got.size == 0:
        return Attribution(0, 0, 0, 0, 0)
    scale = float(np.max(np.abs(want))) + floor + 1e-30
    diff = got - want
    err = np.abs(diff)
    assert np.isfinite(err).all(), "%s: non-finite values" % what
    # the achieved error, published whatever the verdict below (how much slack does the tolerance leave?)
    flat = err.reshape(-1)
    k = max(0, min(flat.size - 1, int(np.ceil(0.9999 * flat.size)) - 1))
    achieved = dict(max_err=float(flat.max() / scale), p9999=float(np.partition(flat, k)[k] / scale),
                    max_err_plain=float((err[factor * amp < tol * scale].max() if (factor * amp < tol * scale).any() else 0.0) / scale),
                    scale=scale, max_err_zero_amp=(float(err[amp == 0].max() / scale) if (amp == 0).any() else None),
                    zero_amp_fraction=float((amp == 0).mean()))
    ACHIEVED_LOG.append(dict(what=str(what), tag=None if tag is None else str(tag), tol=tol, size=int(got.size), **achieved))
    rogue = err > tol * scale + factor * amp
    if rogue.any():
        idx = np.argwhere(rogue)
        worst = idx[np.argmax(err[rogue])]
        raise AssertionError("%s: %d element(s) beyond %g of the inf-norm WITHOUT a near-tie witness; worst at %s: "
                             "got %g want %g (err %g of the norm, oracle flip amplitude %g); %d witnessed outliers" % (
                                 what, len(idx), tol, worst.tolist(), got[tuple(worst)], want[tuple(worst)],
                                 err[tuple(worst)] / scale, amp[tuple(worst)], int(((err > tol * scale) & ~rogue).sum())))
    # (a sanity cap on what a witness may excuse; in a scene of a handful of Gaussians ONE flipped pair legitimately
    # moves a gradient by a large part of its norm -- soak seed 1459: two Gaussians, 15 % -- so it applies to real tensors)
    cap = max_outlier if got.size >= 4096 else 1.0
    assert err.max() <= cap * scale, "%s: worst error %g of the inf-norm exceeds the flip bound %g" % (
        what, err.max() / scale, cap)
    out = err > tol * scale
    # fragile = elements whose allowance is as large as the plain tolerance itself, i.e. where the witness can decide anything
    # (the round-off term makes the allowance non-zero almost everywhere; what matters is where it is not negligible)
    res = Attribution(int(out.sum()), int((factor * amp >= tol * scale).sum()), int(got.size), int((diff[out] > 0).sum()),
                      int((diff[out] < 0).sum()), **achieved)
    ATTRIBUTION_LOG.append(dict(what=str(what), tag=None if tag is None else str(tag), **res._asdict()))
    frac = MAX_OUTLIER_FRACTION if max_fraction is None else max_fraction
    assert res.outliers <= max(MIN_OUTLIER_COUNT, frac * res.size), \
        "%s: %d witnessed outliers among %d elements (more than %g of them)" % (what, res.outliers, res.size, frac)
    return res


def sign_balance(records):
    """(pos, neg, z) over a set of ATTRIBUTION_LOG records: z = (pos - neg) / sqrt(pos + neg), the deviation of the sign
    of got - want over the witnessed outliers from a fair coin in standard deviations -- IF the outliers were independent.
    They are not: one flipped decision moves a cluster of elements the same way (a swapped pair of depth neighbours
    every pixel where the two overlap, times three channels; one flipped pixel every gradient component of the
    Gaussians behind it), so z overstates the evidence by the square root of the cluster size (measured: z = 6.7 over
    14 889 elements of the C4 render comparison whose positive share is 0.53)."""
    pos = sum(r["pos"] for r in records)
    neg = sum(r["neg"] for r in records)
    return pos, neg, (pos - neg) / max(1.0, float(np.sqrt(pos + neg)))


def assert_sign_balanced(records, what, lo=0.25, min_count=50):
    """An implementation that resolved near-ties systematically ONE way (a biased exp, a one-sided threshold) puts
    (nearly) all of its witnessed outliers on one side of the oracle; two unbiased ones split them about evenly, up to
    the clustering described in sign_balance().  Asserted once a set holds `min_count` outliers: the positive share lies
    in [lo, 1 - lo].  -> (pos, neg, z, share)"""
    pos, neg, z = sign_balance(records)
    n = pos + neg
    share = pos / n if n else 0.5
    if n >= min_count:
        assert lo <= share <= 1.0 - lo, "%s: %d of %d witnessed outliers lie above the oracle (share %.2f, z = %.1f): " \
            "near-ties are resolved one way" % (what, pos, n, share, z)
    return pos, neg, z, share


def dump_attribution_log(name, extra=None):
    """Append the records collected so far (and `extra`) to gpurun_out/<name>.jsonl when that directory exists (the GPU
    box merges it back; copies judged live under profiles/)."""
    import json
    import os

    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(d):
        return None
    path = os.path.join(d, name + ".jsonl")
    if isinstance(extra, dict):  # which flavour of the blend kernels the record belongs to (tests/conftest.py forces both)
        try:
            from fsgs_amd import rasterizer

            extra = dict(extra, blend_variant=rasterizer.blend_variant(), deterministic=rasterizer.deterministic())
        except Exception:  # noqa: BLE001
            pass
    with open(path, "a") as f:
        if extra is not None:
            f.write(json.dumps(extra) + "\n")
    return path
