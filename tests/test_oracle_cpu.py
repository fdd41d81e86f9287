"""CPU tests of the oracle itself (no GPU): algebra cross-checks, properties, golden fixtures."""
import os

import numpy as np
import pytest

import np_mirror
from mvs_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rot_cam(W, H, center, yaw, pitch):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return synth.camera_at(center, W, H, rot=Rx @ Ry)


def test_view_matrix_matches_float64_algebra(oracle):
    W, H = 96, 64
    main = synth.camera_at([0, 0, 0], W, H)
    side = _rot_cam(W, H, [0.2, -0.1, 0.05], 0.04, -0.03)
    q = oracle.view_matrix(main, side, W, H).astype(np.float64)
    T = side.astype(np.float64) @ np.linalg.inv(main.astype(np.float64))
    ref = np.stack([W / 2 * T[0] + (W / 2 + 0.5) * T[3], -H / 2 * T[1] + (H / 2 + 0.5) * T[3], T[3]])
    np.testing.assert_allclose(q, ref, rtol=2e-6, atol=1e-6 * np.abs(ref).max())


def test_pad_image_wraps(oracle):
    img = np.arange(5 * 7, dtype=np.uint8).reshape(5, 7)
    pad = oracle.pad_image(img)
    assert pad.shape == (7, 9)
    np.testing.assert_array_equal(pad[1:-1, 1:-1], img)
    np.testing.assert_array_equal(pad[0, 1:-1], img[-1])
    np.testing.assert_array_equal(pad[-1, 1:-1], img[0])
    np.testing.assert_array_equal(pad[1:-1, 0], img[:, -1])
    assert pad[0, 0] == img[-1, -1] and pad[-1, -1] == img[0, 0]


def test_plane_table(oracle):
    z = oracle.plane_table(4, -1.0, 1.0)
    np.testing.assert_array_equal(z, np.array([-0.75, -0.25, 0.25, 0.75], np.float32))
    assert not np.any(oracle.plane_table(128, -1.0, 1.0) == 1.0)  # never backgroundDepth


@pytest.mark.parametrize("rot", [False, True])
def test_sample_plane_matches_numpy_mirror(oracle, rot):
    """the folded f32 sample path agrees with the float64 two-camera warp except at u8 rounding edges"""
    W, H, D = 80, 56, 8
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, 2, freq_scale=0.3)
    if rot:
        side_cams = np.stack([_rot_cam(W, H, [0.12, 0.05, 0.02], 0.03, 0.02), _rot_cam(W, H, [-0.5, 0.3, -0.1], -0.3, 0.2)])
    z = oracle.plane_table(D, -1.0, 1.0)
    for v in range(2):
        # single-view sweep against a black main image: cell = cnt<<16 | Iq
        _, _, _, vol = oracle.sweep(main_cam, np.zeros((H, W), np.uint8), side_cams[v:v + 1], sides[v:v + 1], D,
                                    want_volume=True)
        for d in range(D):
            res, valid = np_mirror.warp_plane(main_cam, side_cams[v], sides[v], float(z[d]))
            cnt = (vol[d] >> 16).astype(bool)
            iq = (vol[d] & 0xffff).astype(np.float64)
            # validity may differ only where the position is within float noise of the frame edge
            assert np.mean(cnt != valid) < 2e-3
            both = cnt & valid
            diff = np.abs(iq - np.floor(res + 0.5))[both]
            assert diff.max() <= 1.0
            assert np.mean(diff > 0) < 5e-3  # only at x.5 rounding boundaries


def test_sweep_recovers_ground_truth_depth(oracle):
    W, H, D, V = 192, 144, 32, 4
    main_cam, main_img, side_cams, sides, gt = synth.make_views(W, H, V, radius=0.4)
    depth, cost, idx, _ = oracle.sweep(main_cam, main_img, side_cams, sides, D, nthreads=4)
    step = 2.0 / D
    err = np.abs(depth - gt)[8:-8, 8:-8]
    assert np.median(err) <= step
    assert np.mean(err <= 1.5 * step) > 0.95
    assert np.all(idx >= 0) and np.all(np.isfinite(cost))


def test_no_views_gives_background(oracle):
    W, H = 16, 12
    main_cam, main_img, _, _, _ = synth.make_views(W, H, 0)
    depth, cost, idx, vol = oracle.sweep(main_cam, main_img, np.zeros((0, 4, 4)), [], 4, want_volume=True)
    assert np.all(depth == np.float32(1.0)) and np.all(idx == -1) and np.all(np.isinf(cost)) and np.all(vol == 0)


def test_argmin_ties_and_empty(oracle):
    z = oracle.plane_table(4, -1.0, 1.0)
    vol = np.zeros((4, 1, 4), np.uint32)
    # pixel 0: equal normalised cost 10/2 == 5/1 at d=1 and d=2 -> lowest d wins
    vol[:, 0, 0] = [(1 << 16) | 9, (2 << 16) | 10, (1 << 16) | 5, (1 << 16) | 7]
    # pixel 1: no valid plane
    # pixel 2: only d=3 valid
    vol[3, 0, 2] = (3 << 16) | 30
    # pixel 3: 7/3 < 5/2
    vol[:, 0, 3] = [(2 << 16) | 5, (3 << 16) | 7, 0, (1 << 16) | 3]
    depth, cost, idx = oracle.argmin(vol, z)
    assert idx[0].tolist() == [1, -1, 3, 1]
    assert depth[0, 1] == np.float32(1.0) and np.isinf(cost[0, 1])
    assert cost[0, 0] == 5.0 and cost[0, 2] == 10.0 and cost[0, 3] == np.float32(7) / np.float32(3)


def test_view_shards_add_exactly(oracle):
    """packed cells of view shards sum to the full-view volume (the all-reduce invariant)"""
    W, H, D, V = 64, 40, 16, 4
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, freq_scale=0.3)
    _, _, _, full = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True)
    acc = np.zeros_like(full)
    for r in range(2):
        _, _, _, part = oracle.sweep(main_cam, main_img, side_cams[2 * r:2 * r + 2], sides[2 * r:2 * r + 2], D,
                                     want_volume=True)
        acc += part
    np.testing.assert_array_equal(acc, full)
    z = oracle.plane_table(D, -1.0, 1.0)
    d_full = oracle.sweep(main_cam, main_img, side_cams, sides, D)[0]
    np.testing.assert_array_equal(oracle.argmin(acc, z)[0], d_full)


def test_mix_background(oracle):
    rng = np.random.default_rng(1)
    H, W = 6, 9
    img3 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img3[..., 1] = np.where(rng.random((H, W)) < 0.4, 0, 255)
    bg = rng.integers(0, 256, (H, W), dtype=np.uint8)
    depth = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.3] = 1.0
    out, d2 = oracle.mix_background(img3, bg, depth)
    masked = (depth == 1.0) | (img3[..., 1] == 0)
    np.testing.assert_array_equal(out, np.where(masked, bg, img3[..., 0]))
    np.testing.assert_array_equal(d2, np.where(masked, np.float32(1.0), depth))


def test_golden_sweep_fixture(oracle):
    """pins the oracle against the committed vectors (tests/golden/make_golden.py regenerates them)"""
    g = np.load(os.path.join(GOLDEN, "sweep_small.npz"))
    depth, cost, idx, vol = oracle.sweep(g["main_cam"], g["main_img"], g["side_cams"], list(g["side_imgs"]),
                                         int(g["D"]), want_volume=True)
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(depth, g["depth"])
    np.testing.assert_array_equal(vol, g["vol"])


def test_golden_stage_fixture(oracle):
    """the reference's per-frame stage at 96x64 (raster, mixBackground, compare, flowRemap, both flows, triangulatePixels,
    filterPoints) against the committed vectors: pins every oracle module against silent drift.  Integer / byte / f32 outputs
    of the fixed-order C code must reproduce exactly; the triangulation's pdf-scaled normals go through exp / pow of libm and
    get a last-ulp tolerance."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    g = np.load(os.path.join(GOLDEN, "stage_small.npz"))
    out = make_golden.stage_outputs(oracle)
    assert set(out) == set(g.files)
    for k in g.files:
        if k == "points_probe":      # positions exact; pdf-scaled normals within a last-ulp tolerance
            assert out[k].shape == g[k].shape
            np.testing.assert_array_equal(out[k][:, :4], g[k][:, :4])
            ok = np.isfinite(g[k][:, 4:]).all(1)
            np.testing.assert_allclose(out[k][ok, 4:], g[k][ok, 4:], rtol=1e-5, atol=1e-9)
        else:
            np.testing.assert_array_equal(out[k], g[k], err_msg=k)
    assert int(g["points_n"]) > 1000 and 0 < int(g["filter_keep_n"]) < int(g["filter_n"])


# ---- the fixed sampler (contract v2) ---------------------------------------------------------------------------------

def test_fixed_sampler_weight_table():
    """32 x 32 entries of four 8-bit weights: every entry sums to 255, the texel-centre entry is (255, 0, 0, 0), symmetric under
    x <-> y, within rounding of the exact bilinear weights"""
    from orc import load
    lut = load().fx_weight_table().astype(np.int64)
    assert lut.shape == (32, 32, 4) and (lut.sum(-1) == 255).all()
    np.testing.assert_array_equal(lut[0, 0], [255, 0, 0, 0])
    np.testing.assert_array_equal(lut[0, 16], [127, 128, 0, 0])          # the fix-up goes to the first of the largest weights
    np.testing.assert_array_equal(lut[:, :, 1], lut[:, :, 2].T)            # w01(kx, ky) = w10(ky, kx)
    a = np.arange(32) / 32.0
    exact = np.stack([np.outer(1 - a, 1 - a), np.outer(1 - a, a), np.outer(a, 1 - a), np.outer(a, a)], -1) * 255.0   # [ky][kx]
    assert np.abs(lut - exact).max() <= 1.5 and np.abs(lut - exact).mean() < 0.35   # 1.41 where the sum fix-up lands


def _mirror_fixed(oracle, main_cam, main_img, side_cam, side, D):
    """numpy restatement of one view of orc_sweep_fx (float64 products are exact for f32 factors; rint = round half to even)"""
    H, W = main_img.shape
    f32 = np.float32

    def fma32(a, b, c):
        return (np.float64(a) * np.float64(b) + np.float64(c)).astype(f32)
    Q = oracle.view_matrix(main_cam, side_cam, W, H).astype(f32)
    z = oracle.plane_table(D, -1.0, 1.0)
    pad = oracle.pad_image(side).astype(np.int64)
    lut = oracle.fx_weight_table().astype(np.int64)
    xn = fma32(f32(2 * np.arange(W) + 1), f32(1.0) / f32(W), f32(-1.0))
    yn = fma32(-f32(2 * np.arange(H) + 1), f32(1.0) / f32(H), f32(1.0))
    XN, YN = np.meshgrid(xn, yn)
    A = [fma32(Q[r, 0], XN, fma32(Q[r, 1], YN, Q[r, 3])) for r in range(3)]
    out = np.zeros((D, H, W), np.uint32)
    for d in range(D):
        sx, sy, sw = fma32(z[d], Q[0, 2], A[0]), fma32(z[d], Q[1, 2], A[1]), fma32(z[d], Q[2, 2], A[2])
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            r256 = ((f32(1.0) / sw).astype(f32) * f32(256)).astype(f32)
            ux = np.nan_to_num(np.rint(sx.astype(np.float64) * r256.astype(np.float64) + 4.0), nan=-1, posinf=-1, neginf=-1).astype(np.int64)
            uy = np.nan_to_num(np.rint(sy.astype(np.float64) * r256.astype(np.float64) + 4.0), nan=-1, posinf=-1, neginf=-1).astype(np.int64)
        ok = (sw > 0) & (ux > 132) & (ux < 256 * W + 132) & (uy > 132) & (uy < 256 * H + 132)
        ix, iy = np.clip(ux >> 8, 0, W), np.clip(uy >> 8, 0, H)
        w = lut[(uy >> 3) & 31, (ux >> 3) & 31]
        dot = w[..., 0] * pad[iy, ix] + w[..., 1] * pad[iy, ix + 1] + w[..., 2] * pad[iy + 1, ix] + w[..., 3] * pad[iy + 1, ix + 1]
        out[d] = np.where(ok, (1 << 24) + np.abs(dot - 255 * main_img.astype(np.int64)), 0)
    return out


def test_fixed_sampler_matches_an_independent_numpy_mirror(oracle):
    """orc_sweep_fx cell for cell against a vectorised numpy restatement of the contract (1/256-texel coordinates by one rounding of
    the exact product, 1/32-texel table index, in-frame test on the rounded coordinate), ring and rotated cameras, ragged size"""
    W, H, D = 90, 40, 9
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, 3, radius=0.4, freq_scale=0.3)
    side_cams = side_cams.copy()
    cy, sy = np.cos(0.3), np.sin(0.3)
    side_cams[2] = synth.camera_at([-0.5, 0.3, -0.2], W, H, rot=np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]))
    for v in range(3):
        vol = oracle.sweep(main_cam, main_img, side_cams[v:v + 1], sides[v:v + 1], D, want_volume=True, sampler="fixed")[3]
        np.testing.assert_array_equal(vol, _mirror_fixed(oracle, main_cam, main_img, side_cams[v], sides[v], D))
        assert (vol >> 24).max() == 1 and (vol == 0).any() or v < 2


def test_fixed_sampler_cells_add_over_views_and_select_like_the_exact_one(oracle):
    """(1) shard volumes add to the full volume as integers (the all-reduce invariant); (2) orc_argmin_fx on the volume equals the
    sweep's own selection; (3) against the exact sampler: identical in-frame counts, mean cost within a fraction of a grey level"""
    W, H, D, V = 160, 96, 16, 4
    main_cam, main_img, side_cams, sides, gt = synth.make_views(W, H, V, radius=0.3, freq_scale=0.3)
    d, c, i, full = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, sampler="fixed")
    acc = np.zeros_like(full)
    for r in range(2):
        acc += oracle.sweep(main_cam, main_img, side_cams[2 * r:2 * r + 2], sides[2 * r:2 * r + 2], D, want_volume=True, sampler="fixed")[3]
    np.testing.assert_array_equal(acc, full)
    d2, c2, i2 = oracle.argmin(full, oracle.plane_table(D, -1.0, 1.0), sampler="fixed")
    np.testing.assert_array_equal(i2, i)
    np.testing.assert_array_equal(d2, d)
    np.testing.assert_array_equal(c2, c)
    v1 = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True)[3]
    np.testing.assert_array_equal(v1 >> 16, full >> 24)
    cnt = np.maximum(v1 >> 16, 1)
    diff = np.abs((v1 & 0xffff) / cnt - (full & 0xffffff) / cnt / 255.0)
    assert diff.mean() < 0.3 and diff.max() < 2.0, (diff.mean(), diff.max())


def test_golden_sweep_fixture_fixed_sampler(oracle):
    """pins orc_sweep_fx and its weight table against the committed vectors (tests/golden/make_golden.py regenerates them)"""
    g = np.load(os.path.join(GOLDEN, "sweep_small.npz"))
    f = np.load(os.path.join(GOLDEN, "sweep_small_fx.npz"))
    depth, cost, idx, vol = oracle.sweep(g["main_cam"], g["main_img"], g["side_cams"], list(g["side_imgs"]), int(f["D"]), want_volume=True,
                                         sampler="fixed")
    np.testing.assert_array_equal(oracle.fx_weight_table(), f["lut"])
    np.testing.assert_array_equal(vol, f["vol"])
    np.testing.assert_array_equal(idx, f["idx"])
    np.testing.assert_array_equal(depth, f["depth"])
    np.testing.assert_array_equal(cost, f["cost"])


def test_golden_rect_fixture(oracle):
    """pins both samplers' oracles on the geometry the rectified-view kernels serve (tests/golden/sweep_rect_small.npz; the GPU suite holds
    sweep_fx_rect / sweep_exact_rect against the same file without the oracle in the loop)"""
    import zlib
    g = np.load(os.path.join(GOLDEN, "sweep_rect_small.npz"))
    for sampler in ("exact", "fixed"):
        depth, cost, idx, vol = oracle.sweep(g["main_cam"], g["main_img"], g["side_cams"], list(g["side_imgs"]), int(g["D"]), want_volume=True, sampler=sampler)
        np.testing.assert_array_equal(depth, g["depth_" + sampler])
        np.testing.assert_array_equal(cost, g["cost_" + sampler])
        np.testing.assert_array_equal(idx, g["idx_" + sampler])
        np.testing.assert_array_equal(vol[::2, ::2, ::2], g["vol_probe_" + sampler])
        assert np.uint32(zlib.crc32(np.ascontiguousarray(vol).tobytes())) == g["vol_crc_" + sampler]
