"""The rectified-view kernel of the EXACT sampler (csrc/sweep_xrect.hip: the reference-closest contract -- f32 bilinear of shader.frag:22,
rounded to u8 like the RGB8 read-back of render_glx.cpp:359) against sweep_tiled and the oracle: same cells, same depth, bit for bit -- on
the SURVEY 8d ring, with ragged sizes, views partly out of frame, view subsets, row bands, plane groups and plane-split launches."""
import numpy as np
import pytest

import mvs_amd
from mvs_amd import synth

pytestmark = pytest.mark.gpu

BOTH = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
NR = mvs_amd.MVS_SWEEP_NO_RECT


def _run(ctx, V, flags, want_volume=True):
    ctx.sweep_run(0, V, flags)
    return ctx.sweep_fetch(want_volume=want_volume)


def _same(a, b):
    for x, y, name in zip(a, b, ("depth", "cost", "index", "volume")):
        if x is None and y is None:
            continue
        bad = np.count_nonzero(x != y)
        assert bad == 0, "%s: %d of %d differ" % (name, bad, x.size)


@pytest.mark.parametrize("W,H,D,V,radius", [
    (64, 8, 16, 1, 0.05),       # one tile, one chunk
    (200, 90, 20, 3, 0.3),      # ragged tiles, ragged last chunk, wide baseline
    (320, 240, 32, 4, 0.3),
    (640, 480, 128, 4, 0.15),   # c5's shape on the ring
    (1280, 720, 64, 8, 0.15),   # c2
    (333, 77, 37, 5, 0.6),      # odd sizes, views mostly out of frame at the near planes
    (66, 10, 16, 2, 0.2),
])
def test_xrect_equals_tiled_and_oracle(oracle, W, H, D, V, radius):
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=radius, freq_scale=max(W / 1920.0, 0.25))
    with mvs_amd.Context(W, H, sampler="exact") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        rect = _run(ctx, V, BOTH)
        shape = ctx.plan_shape()
        gen = _run(ctx, V, BOTH | NR)
        assert ctx.plan_shape() in (1, 2)
        fused_only = _run(ctx, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN, want_volume=False)
        vol_only = _run(ctx, V, mvs_amd.MVS_SWEEP_VOLUME)[3]
    _same(rect, gen)
    _same(fused_only[:3], gen[:3])
    np.testing.assert_array_equal(vol_only, gen[3])
    if W * H * D * V <= 320 * 240 * 32 * 4 * 4:
        ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler="exact")
        np.testing.assert_array_equal(rect[3], ref[3])
        np.testing.assert_array_equal(rect[2], ref[2])
        np.testing.assert_array_equal(rect[0], ref[0])
    assert shape in (1, 2, 5)
    if W >= 640 or (W, H) == (64, 8):
        assert shape == 5, "BASELINE's shapes on the ring should take the rectified kernel (small images with wide baselines have boxes beyond the LDS slots)"


def test_xrect_noise_frames_c3(oracle):
    """adversarial input (i.i.d. noise): every cell matters; c3 in full against sweep_tiled (which tests/test_sweep_gpu.py holds against the oracle)"""
    W, H, D, V = 1920, 1080, 128, 16
    main_cam, main_img, side_cams, sides = synth.noise_views(W, H, V)
    with mvs_amd.Context(W, H, sampler="exact") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        gen = _run(ctx, V, BOTH | NR)
        rect = _run(ctx, V, BOTH)
        assert ctx.plan_shape() == 5
    _same(rect, gen)


def test_xrect_view_subsets_plane_groups_row_bands_and_splits():
    """view-sharded use: subsets of the views into the volume, plane groups, then the separate depth selection; row bands; forced plane splits"""
    W, H, D, V = 640, 360, 64, 6
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.2, freq_scale=0.5)
    with mvs_amd.Context(W, H, sampler="exact") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        full = _run(ctx, V, BOTH | NR)
        acc = np.zeros_like(full[3])
        for v0, vn in ((0, 2), (2, 3), (5, 1)):
            for p0, pn in ((0, 32), (32, 32)):
                ctx.sweep_run_planes(v0, vn, p0, pn, mvs_amd.MVS_SWEEP_VOLUME)
                assert ctx.plan_shape() == 5
                vol = ctx.sweep_fetch(want_volume=True)[3]
                acc[p0:p0 + pn] += vol[p0:p0 + pn]
        np.testing.assert_array_equal(acc, full[3])
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME)
        ctx.sweep_argmin()
        _same(ctx.sweep_fetch(want_volume=False)[:3], full[:3])
        g = ctx.row_granularity()
        ctx.sweep_run(0, 0, BOTH | NR)   # empty the results (a run over no views)
        for r0 in range(0, H, 3 * g):
            ctx.sweep_run_rows(r0, min(3 * g, H - r0), 0, V, BOTH)
        _same(ctx.sweep_fetch(want_volume=False)[:3], full[:3])
        for nsplit in (1, 2, 3, 4):
            _same(_run(ctx, V, BOTH | (nsplit << 16), want_volume=False)[:3], full[:3])


def test_xrect_is_left_when_a_view_is_not_rectified_or_the_boxes_do_not_fit(oracle):
    W, H, D, V = 320, 160, 16, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.9)   # wide baseline, few planes: boxes wider than the slots
    with mvs_amd.Context(W, H, sampler="exact") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        got = _run(ctx, V, BOTH)
        assert ctx.plan_shape() in (1, 2)
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler="exact")
    np.testing.assert_array_equal(got[3], ref[3])
