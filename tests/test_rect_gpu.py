"""The rectified-view kernel of the fixed sampler (csrc/sweep_rect.hip) against the general tiled kernel and the oracle:
same cells, same depth, bit for bit -- on the SURVEY 8d ring, with ragged sizes, with views partly out of frame, with view
subsets, row bands, plane groups and plane-split launches."""
import numpy as np
import pytest

import mvs_amd
from mvs_amd import synth

pytestmark = pytest.mark.gpu

BOTH = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN


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
])
def test_rect_equals_general_and_oracle(oracle, W, H, D, V, radius):
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=radius, freq_scale=max(W / 1920.0, 0.25))
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        rect = _run(ctx, V, BOTH)
        shape = ctx.plan_shape()
        gen = _run(ctx, V, BOTH | mvs_amd.MVS_SWEEP_NO_RECT)
        fused_only = _run(ctx, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN, want_volume=False)
    _same(rect, gen)
    _same(fused_only[:3], gen[:3])
    if W * H * D * V <= 320 * 240 * 32 * 4 * 4:
        ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler="fixed")
        np.testing.assert_array_equal(rect[3], ref[3])
        np.testing.assert_array_equal(rect[2], ref[2])
    assert shape in (3, 4)
    if radius <= 0.3 and W >= 200:
        assert shape == 4, "the ring geometry should take the rectified kernel"


def test_rect_noise_frames_c3_shape_rows(oracle):
    """adversarial input (i.i.d. noise): every cell matters; a 64-row band of c3 against the general kernel"""
    W, H, D, V = 1920, 1080, 128, 16
    main_cam, main_img, side_cams, sides = synth.noise_views(W, H, V)
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, BOTH | mvs_amd.MVS_SWEEP_NO_RECT)
        gen = ctx.sweep_fetch(want_volume=False)
        ctx.sweep_run(0, V, BOTH)
        assert ctx.plan_shape() == 4
        rect = ctx.sweep_fetch(want_volume=False)
    _same(rect[:3], gen[:3])


def test_rect_view_subsets_and_plane_groups():
    """view-sharded use: subsets of the views into the volume, plane groups, then the separate depth selection"""
    W, H, D, V = 640, 360, 64, 6
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.2, freq_scale=0.5)
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, BOTH | mvs_amd.MVS_SWEEP_NO_RECT)
        full = ctx.sweep_fetch(want_volume=True)
        assert ctx.plan_shape() == 4
        acc = np.zeros_like(full[3])
        for v0, vn in ((0, 2), (2, 3), (5, 1)):
            for p0, pn in ((0, 32), (32, 32)):
                ctx.sweep_run_planes(v0, vn, p0, pn, mvs_amd.MVS_SWEEP_VOLUME)
                vol = ctx.sweep_fetch(want_volume=True)[3]
                acc[p0:p0 + pn] += vol[p0:p0 + pn]
        np.testing.assert_array_equal(acc, full[3])
        # row bands with fused selection
        ctx.sweep_run(0, V, BOTH | mvs_amd.MVS_SWEEP_NO_RECT)
        ref = ctx.sweep_fetch(want_volume=False)
        g = ctx.row_granularity()
        for r0 in range(0, H, 5 * g):
            ctx.sweep_run_rows(r0, min(5 * g, H - r0), 0, V, BOTH)
        band = ctx.sweep_fetch(want_volume=False)
    _same(band[:3], ref[:3])


def test_rect_forced_plane_splits():
    W, H, D, V = 512, 128, 96, 4
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.15, freq_scale=0.5)
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ref = _run(ctx, V, BOTH | mvs_amd.MVS_SWEEP_NO_RECT, want_volume=False)
        for nsplit in (1, 2, 3, 6):
            got = _run(ctx, V, BOTH | (nsplit << 16), want_volume=False)
            _same(got[:3], ref[:3])
