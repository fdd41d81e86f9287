"""GPU parity of triangulatePixels(): positions are f32/f64 arithmetic in a fixed order -> expected bit-exact; pdf and
normals go through exp()/pow() of the device math library -> relative tolerance 1e-5 (stated)."""
import numpy as np
import pytest

import mvs_amd
import scenes
from mvs_amd import synth

pytestmark = pytest.mark.gpu


def test_triangulate_matches_oracle_on_the_recon_stage(oracle):
    """depth -> projected -> mixBackground -> calculateFlow per side view -> triangulatePixels (recon.cpp:70-114)"""
    W, H = 320, 240
    verts, faces = scenes.heightfield_mesh(64, extent=1.4)
    soup = oracle.load_mesh(verts, faces)
    sc = synth.Scene(freq_scale=0.2)
    main_c = [0.0, 0.0, 0.0]
    side_cs = [[0.15, 0.0, 0.0], [-0.1, 0.12, 0.03], [0.02, -0.14, -0.02]]
    main = synth.camera_at(main_c, W, H)
    sides = np.stack([synth.camera_at(c, W, H) for c in side_cs])
    main_img = sc.render(main_c, W, H)
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        depth = ctx.depth(main)
        flows = []
        for c, cam in zip(side_cs, sides):
            proj = ctx.projected(main, sc.render(c, W, H), cam)
            mixed, depth = ctx.mix_background(proj, main_img, depth)
            flows.append(ctx.flow(main_img, mixed, False))
        got = ctx.triangulate(flows, main, sides, depth)
    ref = oracle.triangulate_pixels(flows, main, sides, depth)
    assert got.shape == ref.shape and ref.shape[0] > 0.2 * W * H
    np.testing.assert_array_equal(got[:, :4], ref[:, :4])
    np.testing.assert_allclose(got[:, 4:], ref[:, 4:], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("W,H", [(200, 96), (201, 97)])   # the second: W*H not a multiple of 4 (16-byte-aligned records in the arena)
def test_triangulate_edge_cases(oracle, W, H):
    verts, faces = scenes.heightfield_mesh(48, extent=2.2)
    soup = oracle.load_mesh(verts, faces)
    main = synth.camera_at([0, 0, 0], W, H)
    sides = np.stack([synth.camera_at([0.2, 0.05, 0], W, H), synth.camera_at([-0.2, 0.0, 0.05], W, H)])
    depth = oracle.depth(soup, main, W, H)
    depth[:, :7] = 1.0                      # background stripe: fewer than 3 neighbours never happens here, but holes do
    depth[40:44, 100:104] = 1.0
    rng = np.random.default_rng(4)
    flows = []
    for _ in range(2):
        f = np.zeros((H, W, 4), np.float32)
        f[..., :2] = rng.normal(0, 1.5, (H, W, 2))
        f[..., 2] = rng.uniform(0.5, 4.0, (H, W))
        f[:3, :, 0] = -50.0                 # flows pointing far outside the image: goodSample fails
        flows.append(f)
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.triangulate(flows, main, sides, depth)
        none = ctx.triangulate([], main, np.zeros((0, 4, 4), np.float32), depth)
    ref = oracle.triangulate_pixels(flows, main, sides, depth)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got[:, :4], ref[:, :4])
    finite = np.isfinite(ref[:, 4:]).all(1)
    np.testing.assert_allclose(got[finite, 4:], ref[finite, 4:], rtol=1e-5, atol=1e-9)
    assert np.array_equal(np.isfinite(got[:, 4:]), np.isfinite(ref[:, 4:]))
    ref0 = oracle.triangulate_pixels([], main, np.zeros((0, 4, 4), np.float32), depth)
    assert none.shape == ref0.shape


@pytest.mark.parametrize("nviews", [5, 8, 11])
def test_triangulate_view_counts(oracle, nviews):
    """tri_points_kernel is instantiated for up to 4, up to 8 and up to 32 side views (per-view arrays in registers vs
    scratch); 2-4 views are covered above, these counts exercise the other two instantiations against the oracle"""
    W, H = 160, 96
    verts, faces = scenes.heightfield_mesh(40, extent=2.2)
    soup = oracle.load_mesh(verts, faces)
    main = synth.camera_at([0, 0, 0], W, H)
    rng = np.random.default_rng(nviews)
    centres = [[0.25 * np.cos(2 * np.pi * k / nviews), 0.25 * np.sin(2 * np.pi * k / nviews), 0.03 * (k % 3 - 1)] for k in range(nviews)]
    sides = np.stack([synth.camera_at(c, W, H) for c in centres])
    depth = oracle.depth(soup, main, W, H)
    flows = []
    for _ in range(nviews):
        f = np.zeros((H, W, 4), np.float32)
        f[..., :2] = rng.normal(0, 1.0, (H, W, 2))
        f[..., 2] = rng.uniform(0.5, 4.0, (H, W))
        flows.append(f)
    with mvs_amd.Context(W, H) as ctx:
        got = ctx.triangulate(flows, main, sides, depth)
    ref = oracle.triangulate_pixels(flows, main, sides, depth)
    assert got.shape == ref.shape and got.shape[0] > 0.5 * W * H
    np.testing.assert_array_equal(got[:, :4], ref[:, :4])
    finite = np.isfinite(ref[:, 4:]).all(1)
    np.testing.assert_allclose(got[finite, 4:], ref[finite, 4:], rtol=1e-5, atol=1e-9)
