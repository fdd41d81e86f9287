"""CPU tests of the Heuristic::filterPoints restatement (heuristic.cpp:55-176)."""
import numpy as np


def _cloud(rng, N, outliers=20):
    pts = np.concatenate([rng.uniform(0, 1, (N, 2)), 0.02 * rng.uniform(0, 1, (N, 1)), np.ones((N, 1))], 1)
    out = np.concatenate([rng.uniform(5, 9, (outliers, 3)), np.ones((outliers, 1))], 1)
    allp = np.concatenate([pts, out]).astype(np.float32)
    allp *= rng.uniform(0.5, 2.0, (allp.shape[0], 1)).astype(np.float32)   # homogeneous scale must not matter
    return allp


def test_outliers_removed_and_cluster_thinned(oracle):
    rng = np.random.default_rng(0)
    N = 3000
    pts = _cloud(rng, N)
    keep, dens = oracle.filter_points(pts, 0.01)
    assert np.all(np.diff(keep) > 0)                    # ascending original indices (heuristic.cpp:166)
    assert not np.any(keep >= N)                        # isolated points have score 0 < 0.7
    assert 0.3 * N < len(keep) < N                      # redundancy removed, cluster kept
    assert dens.max() <= 2.0 and np.all(dens[N:] == 0)  # clamp at 2 (heuristic.cpp:127-128); outliers have no neighbours


def test_radius_is_compared_with_squared_distance(oracle):
    """two points at distance d are neighbours iff d^2 <= alpha/4 (SURVEY A-13), not d <= alpha/4"""
    alpha = 0.04                       # radius 0.01 -> reach 0.1
    near = np.array([[0, 0, 0, 1], [0.09, 0, 0, 1]], np.float32)   # d = 0.09 > radius, d^2 = 0.0081 <= radius
    _, dens = oracle.filter_points(near, alpha)
    assert np.all(dens > 0)
    far = np.array([[0, 0, 0, 1], [0.11, 0, 0, 1]], np.float32)    # d^2 = 0.0121 > radius
    keep2, dens2 = oracle.filter_points(far, alpha)
    # no neighbour pairs at all: the normaliser is N/0 and the densities become NaN, exactly as in the reference; nothing is kept
    assert len(keep2) == 0 and np.all(np.isnan(dens2))


def test_deterministic_and_scale_invariant(oracle):
    rng = np.random.default_rng(1)
    pts = _cloud(rng, 800, outliers=5)
    k1, d1 = oracle.filter_points(pts, 0.01)
    k2, d2 = oracle.filter_points(pts * np.float32(4.0), 0.01)   # homogeneous rescaling by a power of two: same cloud
    np.testing.assert_array_equal(k1, k2)
    np.testing.assert_array_equal(d1, d2)
