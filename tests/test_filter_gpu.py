"""GPU parity of Heuristic::filterPoints: the neighbour table and weights are exact; the power iteration adds f32 terms in
the reference's order; the two global f64 sums use a different (fixed) association than the sequential oracle, which
can move them by ~1e-16 relative -- not enough to change a float in practice, so the kept index set is required to be
identical."""
import numpy as np
import pytest

import mvs_amd

pytestmark = pytest.mark.gpu


def _cloud(rng, N, outliers):
    pts = np.concatenate([rng.uniform(0, 1, (N, 2)), 0.05 * rng.normal(0, 1, (N, 1)), np.ones((N, 1))], 1)
    out = np.concatenate([rng.uniform(5, 9, (outliers, 3)), np.ones((outliers, 1))], 1)
    allp = np.concatenate([pts, out]).astype(np.float32)
    allp *= rng.uniform(0.5, 2.0, (allp.shape[0], 1)).astype(np.float32)
    return allp[rng.permutation(allp.shape[0])]


@pytest.mark.parametrize("N,alpha", [(500, 0.02), (6000, 0.004), (20000, 0.001), (9000, 0.05), (40000, 0.012)])  # the last two: long lists -> global-sort path
def test_filter_points_matches_oracle(oracle, N, alpha):
    rng = np.random.default_rng(N)
    pts = _cloud(rng, N, 30)
    ref, _ = oracle.filter_points(pts, alpha)
    with mvs_amd.Context(64, 48) as ctx:
        got = ctx.filter_points(pts, alpha)
        again = ctx.filter_points(pts, alpha)
    np.testing.assert_array_equal(got, again)   # atomics' arrival order must not leak into the result
    np.testing.assert_array_equal(got, ref)
    assert 0 < len(ref) < N


def test_filter_points_edge_cases(oracle):
    with mvs_amd.Context(64, 48) as ctx:
        assert len(ctx.filter_points(np.zeros((0, 4), np.float32), 0.1)) == 0
        one = np.array([[1, 2, 3, 1]], np.float32)
        assert len(ctx.filter_points(one, 0.1)) == 0          # a lone point has score 0 < 0.7
        with pytest.raises(mvs_amd.MvsError):
            ctx.filter_points(one, 0.0)
        same = np.tile(np.array([[0.5, 0.5, 0.5, 1.0]], np.float32), (50, 1))   # 50 coincident points
        np.testing.assert_array_equal(ctx.filter_points(same, 0.04), oracle.filter_points(same, 0.04)[0])


def test_round_limit_fallback_gives_the_same_selection(oracle, tmp_path):
    """the greedy pass runs in dependency rounds on the device; past a round limit the rest is decided by the reference's
    sequential walk on the host.  With the limit forced down to 8 rounds (MVS_FILTER_MAX_ROUNDS, read once per process,
    hence the child process) the selection must equal the unlimited one and the oracle's"""
    import os
    import subprocess
    import sys
    pts = _cloud(np.random.default_rng(11), 9000, 30)
    alpha = 0.05           # ~350 neighbours per point: dependency chains far longer than the forced limit
    np.save(tmp_path / "pts.npy", pts)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import mvs_amd\n"
            "pts = np.load(%r)\n"
            "with mvs_amd.Context(64, 64) as ctx: keep = ctx.filter_points(pts, %r)\n"
            "np.save(%r, keep)\n") % (os.path.join(root, "mesh-reconstruction_amd", "python"), str(tmp_path / "pts.npy"), alpha,
                                      str(tmp_path / "keep.npy"))
    env = dict(os.environ, MVS_FILTER_MAX_ROUNDS="8", MVS_FILTER_TIMING="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "finished on the host after 8 rounds" in out.stderr, out.stderr[-1500:]   # the fallback really ran
    limited = np.load(tmp_path / "keep.npy")
    with mvs_amd.Context(64, 64) as ctx:
        full = ctx.filter_points(pts, alpha)
    ref = oracle.filter_points(pts, alpha)[0]
    np.testing.assert_array_equal(limited, full)
    np.testing.assert_array_equal(full, ref)


def test_a_radius_far_beyond_the_spacing_is_refused_not_wrapped():
    """heuristic.cpp:74-89 asks for every neighbour within the radius: a radius that spans the whole cloud makes N^2 / 2 of them per table.
    At 70 000 points that is 2.45e9 entries -- more than the 32-bit offsets hold; the prefix sum wraps (to a POSITIVE total here, so the sign
    alone does not show it).  The call must say so, not build a table from wrapped offsets; the context stays usable."""
    rng = np.random.default_rng(3)
    N = 70000
    pts = np.concatenate([rng.uniform(0, 1e-3, (N, 3)), np.ones((N, 1))], 1).astype(np.float32)
    with mvs_amd.Context(64, 48) as ctx:
        with pytest.raises(mvs_amd.MvsError, match="2\\^31"):
            ctx.filter_points(pts, 10.0)
        small = _cloud(rng, 3000, 10)
        assert len(ctx.filter_points(small, 0.01)) > 0
