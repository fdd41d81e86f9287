"""CPU tests of the compare()/flowRemap() restatement (util.cpp:332-361, 390-403)."""
import numpy as np
import pytest


def _levels(W, H):
    size, n = min(W, H), 0
    while True:
        n += 1
        if size <= 2:
            return n
        size //= 2


def test_level_counts_match_survey():
    # 9 levels at 640x480 and 1280x720, 10 at 1080p, 11 at 4K
    assert [_levels(640, 480), _levels(1280, 720), _levels(1920, 1080), _levels(3840, 2160)] == [9, 9, 10, 11]


@pytest.mark.parametrize("W,H", [(640, 480), (97, 61), (33, 20)])
def test_compare_identity_and_constant_offset(oracle, W, H):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 200, (H, W), dtype=np.uint8)
    assert not oracle.compare(a, a).any()
    out = oracle.compare(a, (a + 7).astype(np.uint8))
    # a constant difference survives pyrDown/pyrUp exactly: every level contributes 7
    np.testing.assert_allclose(out, 7.0 * _levels(W, H), rtol=0, atol=1e-3)


def test_compare_is_symmetric_and_localised(oracle):
    rng = np.random.default_rng(1)
    W, H = 128, 96
    a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    b = a.copy()
    b[40:48, 60:70] = 255 - b[40:48, 60:70]
    o1, o2 = oracle.compare(a, b), oracle.compare(b, a)
    np.testing.assert_array_equal(o1, o2)
    assert o1[44, 65] > 10 * o1[5, 5] and o1[5, 5] >= 0
    # f32 entry point agrees with the u8 one
    np.testing.assert_array_equal(oracle.compare(a.astype(np.float32), b.astype(np.float32)), o1)


def test_cubic_table_properties(oracle):
    t = oracle.cubic_table().astype(np.int64)
    assert np.all(t.sum(1) == 32768)                    # forced normalisation
    # zero fraction: 1.0 saturates to 32767 in a short; the residue lands on tap (2,2) (OpenCV's initInterTab2D quirk)
    assert t[0].reshape(4, 4)[1, 1] == 32767 and t[0].reshape(4, 4)[2, 2] == 1 and np.count_nonzero(t[0]) == 2
    w = t[16 * 32 + 16].reshape(4, 4) / 32768.0         # half-pixel: separable (-1/9.. ) symmetric
    np.testing.assert_allclose(w, w.T, atol=2e-4)
    np.testing.assert_allclose(w.sum(0), [-0.09375, 0.59375, 0.59375, -0.09375], atol=3e-4)


def test_remap_identity_shift_and_border(oracle):
    rng = np.random.default_rng(2)
    W, H = 50, 30
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    flow = np.zeros((H, W, 4), np.float32)
    np.testing.assert_array_equal(oracle.flow_remap(flow, img), img)
    flow[..., 0], flow[..., 1] = 3.0, -2.0  # dst(x,y) = src(x+3, y-2)
    out = oracle.flow_remap(flow, img)
    np.testing.assert_array_equal(out[2:, :W - 3], img[:H - 2, 3:])
    assert not out[:2].any() and not out[:, W - 3:].any()   # BORDER_CONSTANT 0
    # 2-channel flow gives the same answer (util.cpp:393-394 keeps channels 0,1)
    np.testing.assert_array_equal(oracle.flow_remap(np.ascontiguousarray(flow[..., :2]), img), out)


def test_remap_matches_float_keys_cubic(oracle):
    """fixed-point result == float a=-0.75 cubic convolution at the 1/32-quantised position, within 1 grey level"""
    rng = np.random.default_rng(3)
    W, H = 64, 48
    yy, xx = np.mgrid[0:H, 0:W]
    img = (127 + 100 * np.sin(xx / 5.0) * np.cos(yy / 7.0)).astype(np.uint8)
    flow = rng.uniform(-2, 2, (H, W, 2)).astype(np.float32)
    out = oracle.flow_remap(flow, img).astype(np.float64)

    def k(t):
        t = np.abs(t)
        A = -0.75
        return np.where(t <= 1, (A + 2) * t ** 3 - (A + 3) * t ** 2 + 1, np.where(t < 2, A * t ** 3 - 5 * A * t ** 2 + 8 * A * t - 4 * A, 0))
    qx = np.rint((flow[..., 0] + xx.astype(np.float32)) * 32) / 32
    qy = np.rint((flow[..., 1] + yy.astype(np.float32)) * 32) / 32
    x0, y0 = np.floor(qx).astype(int), np.floor(qy).astype(int)
    ref = np.zeros((H, W))
    for dy in range(-1, 3):
        for dx in range(-1, 3):
            xs, ys = x0 + dx, y0 + dy
            ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
            v = np.where(ok, img[np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)], 0)
            ref += v * k(qx - xs) * k(qy - ys)
    assert np.abs(out - np.clip(np.rint(ref), 0, 255)).max() <= 1
