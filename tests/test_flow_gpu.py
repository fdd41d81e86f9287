"""GPU parity of calculateFlow() vs the oracle.  Both flow algorithms are f32/f64 arithmetic in a fixed operation
order; the kernels are required to match the oracle bit for bit, with 1e-4 px as the stated float tolerance."""
import numpy as np
import pytest

import mvs_amd

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _pair(W, H, dx, dy, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)

    def tex(x, y):
        return 127 + 50 * np.sin(x / 7.0) * np.cos(y / 9.0) + 40 * np.sin((x + y) / 13.0) + 30 * np.cos((x - 2 * y) / 17.0)
    a = (tex(xx, yy) + rng.normal(0, 2, (H, W))).clip(0, 255).astype(np.uint8)
    b = (tex(xx - dx, yy - dy) + rng.normal(0, 2, (H, W))).clip(0, 255).astype(np.uint8)
    return a, b


@pytest.mark.parametrize("W,H", [(70, 50), (320, 240), (333, 201), (640, 480)])
@pytest.mark.parametrize("farneback", [True, False])
def test_calculate_flow_matches_oracle(oracle, W, H, farneback):
    a, b = _pair(W, H, 2.5, -1.5, seed=W)
    ref = oracle.calculate_flow(a, b, farneback)
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.flow(a, b, farneback)
    assert np.all(np.isfinite(got))
    assert np.abs(got[..., :2] - ref[..., :2]).max() < TOL
    np.testing.assert_array_equal(got[..., :2], ref[..., :2])
    np.testing.assert_array_equal(got[..., 2], ref[..., 2])
    assert not got[..., 3].any()


def test_farneback_tall_tiles_match_oracle(oracle):
    """1280 x 720 (window 20): the finest levels run the tiled iteration kernel with 64 x 16 tiles (8 rows / 4 pixels per thread), the
    coarse ones with 64 x 8 -- same sums in the same order as the oracle's, bit for bit"""
    W, H = 1280, 720
    a, b = _pair(W, H, 3.5, -2.0, seed=5)
    ref = oracle.calculate_flow(a, b, True)
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.flow(a, b, True)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("W,H,farneback", [(1920, 1080, True), (1920, 1080, False), (3840, 2160, True), (3840, 2160, False)])
def test_calculate_flow_matches_oracle_at_the_sizes_baseline_names(oracle, W, H, farneback):
    """flow.cpp:24-26 at BASELINE's frame sizes: window (H + W) / 100 = 30 and poly 7 / 3.0 at c3, window 60 and poly 7 / 6.0 at c4 -- parameter
    sets the smaller cases above never reach -- and the variational default (flow.cpp:29) at both; mvs_flow against oracle.calculate_flow
    (OpenMP over rows, the same sums in the same order), all four channels, bit for bit.  The pair carries sigma = 2 noise: Farneback
    must also RETURN the shift (most of it: the 1e-3 in the 2x2 solve biases it low), so the warp path runs on real sub-pixel flow."""
    dx, dy = 4.0, 1.5
    a, b = _pair(W, H, dx, dy, seed=W + farneback)
    ref = oracle.calculate_flow(a, b, farneback)
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.flow(a, b, farneback)
    assert np.all(np.isfinite(got))
    assert np.abs(got[..., :2] - ref[..., :2]).max() < TOL
    np.testing.assert_array_equal(got, ref)
    if farneback:
        c = got[100:-100, 100:-100]
        assert np.median(c[..., 0]) > 0.7 * dx and np.median(c[..., 1]) > 0.7 * dy


@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 2160), (1000, 1099)])
def test_farneback_tiled_iteration_equals_the_direct_one(monkeypatch, W, H):
    """the tiled iteration kernel (every value read once per thread, running sums in registers) against the round-2 kernel that sums each
    window term by term (test hook MVS_FB_DIRECT_BOX, itself equal to the oracle at the sizes the oracle is run at): windows 30, 60 and 20
    with ragged tile edges; single flows and the batched pass of mvs_process_frame's shape"""
    a, b = _pair(W, H, 4.0, 1.5, seed=W)
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.flow(a, b, True)
    monkeypatch.setenv("MVS_FB_DIRECT_BOX", "1")
    with mvs_amd.Context(W, H) as ctx:
        ref = ctx.flow(a, b, True)
    np.testing.assert_array_equal(got, ref)
    assert np.abs(got[..., :2]).max() > 1.0


def test_flow_of_mixed_background_stage(oracle):
    """flow.cpp is called on (originalImage, mixBackground(projected)) (recon.cpp:86-89): outside the mask the two
    images are identical, so the flow vanishes there"""
    W, H = 320, 240
    a, b = _pair(W, H, 1.0, 0.5)
    mixed = a.copy()
    mixed[60:180, 80:240] = b[60:180, 80:240]
    with mvs_amd.Context(W, H) as ctx:
        out = ctx.flow(a, mixed, False)
    np.testing.assert_array_equal(out, oracle.calculate_flow(a, mixed, False))
    assert np.abs(out[:30, :40, :2]).max() < 0.05


def test_graph_replay_after_the_first_poisson_call_is_right_without_memset_nodes():
    """tools/graph_repro.py, the reproducer of round 4's hipGraph finding: a captured calculateFlow sequence replayed after the process's
    first hipFFT / Poisson call.  Round 5 found the culprit -- the captured hipMemsetAsync NODE (the flow's zero initialisation) stops
    doing its job in such replays; none of this library's kernels, none of which uses scratch.  With a zero-fill kernel in its place
    (test hook MVS_FLOW_GRAPH=2) the replay is bit-identical to the eager launches, which is what this asserts; the library launches
    eagerly either way."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for alg in ("farneback", "variational"):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_repro.py"), alg, "--kernel-memset"], capture_output=True, text=True, timeout=600)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
        r = json.loads(lines[0])
        assert r["replay_before_poisson_equal"] and r["replay_after_poisson_equal"] and r["eager_after_poisson_equal"], r
        assert r["first_buffer_that_differs"] is None
