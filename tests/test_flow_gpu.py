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
