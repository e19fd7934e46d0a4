"""Known-answer tests of the KNN oracle: the reference ships no tests for
simple-knn, so the brute-force definition is pinned on closed-form cases."""
import numpy as np


def test_regular_grid_gives_h_squared(oracle32):
    h = 0.25
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3) * h
    d = oracle32.knn_meandist2(g)
    np.testing.assert_allclose(d, h * h, rtol=1e-6)  # every point has >= 3 axis neighbours at h


def test_duplicates_give_zero_and_self_is_excluded_by_position(oracle32):
    pts = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0], [1, 0, 0]], np.float32)
    d = oracle32.knn_meandist2(pts)
    np.testing.assert_array_equal(d[:4], 0.0)
    np.testing.assert_allclose(d[4], 1.0)


def test_fewer_than_three_neighbours_overflows_like_the_reference(oracle32):
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)
    d = oracle32.knn_meandist2(pts)
    # one missing neighbour: a FLT_MAX term stays in the mean (simple_knn.cu:154,182)
    np.testing.assert_allclose(d, np.float32(3.4028235e38) / 3, rtol=1e-6)
    assert np.isinf(oracle32.knn_meandist2(pts[:2])).all()  # two FLT_MAX terms overflow to +inf
    d4 = oracle32.knn_meandist2(np.vstack([pts, [[0, 0, 3]]]).astype(np.float32))
    np.testing.assert_allclose(d4[0], (1 + 4 + 9) / 3.0, rtol=1e-6)


def test_matches_scipy_kdtree(oracle32):
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(0)
    pts = rng.standard_normal((3000, 3)).astype(np.float32)
    d = oracle32.knn_meandist2(pts)
    dd, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    np.testing.assert_allclose(d, (dd[:, 1:] ** 2).mean(1), rtol=2e-5)
