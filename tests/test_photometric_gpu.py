"""GPU parity of compare() and flowRemap() vs the oracle: f32 pyramid arithmetic in a fixed operation order and
Q15 integer remap -> both bit-exact."""
import numpy as np
import pytest

import mvs_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H", [(640, 480), (333, 211), (64, 48), (1280, 720)])
def test_compare_matches_oracle(oracle, W, H):
    rng = np.random.default_rng(W)
    yy, xx = np.mgrid[0:H, 0:W]
    a = (127 + 90 * np.sin(xx / 9.0) * np.cos(yy / 11.0) + rng.normal(0, 8, (H, W))).clip(0, 255).astype(np.uint8)
    b = np.roll(a, (1, 2), (0, 1))
    b[H // 3:H // 2, W // 4:W // 2] = 30
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.compare(a, b)
        same = ctx.compare(a, a)
    np.testing.assert_array_equal(got, oracle.compare(a, b))
    assert not same.any()


@pytest.mark.parametrize("W,H,stride", [(640, 480, 4), (301, 177, 2)])
def test_flow_remap_matches_oracle(oracle, W, H, stride):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    flow = np.zeros((H, W, stride), np.float32)
    flow[..., :2] = rng.normal(0, 3, (H, W, 2))
    flow[:10, :10, :2] = 1e6      # far outside: constant border
    flow[10:20, :10, 0] = -1e6
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.flow_remap(flow, img)
    np.testing.assert_array_equal(got, oracle.flow_remap(flow, img))


def test_resize_u8_equals_oracle(oracle):
    """mvs_resize_u8 (Configuration's -s / clip-size resize, configuration.cpp:233) == the oracle's restatement of cv::resize, byte for byte"""
    rng = np.random.default_rng(4)
    with mvs_amd.Context(64, 48) as ctx:
        for shape in ((480, 640), (211, 333, 3), (1080, 1920, 3)):
            img = rng.integers(0, 256, shape, dtype=np.uint8)
            sh, sw = shape[:2]
            for dw, dh in ((sw // 2, sh // 2), (int(sw / 1.5), int(sh / 1.5)), (sw, sh), (sw + 37, sh + 11)):
                np.testing.assert_array_equal(ctx.resize(img, dw, dh), oracle.resize_linear(img, dw, dh))
