"""GPU parity tests of the renderer path (Render::depth / Render::projected / mixBackground) vs the CPU oracle.
Index / byte work (coverage, visible face, RGB8 output, masks) must be bit-exact; depth maps are f32 and
are required bit-exact too (same arithmetic contract), with RMSE < 1e-4 as the stated north-star bound."""
import numpy as np
import pytest

import mvs_amd
import scenes
from data import glx_scene
from mvs_amd import synth

pytestmark = pytest.mark.gpu


def _both(oracle, W, H, verts, faces):
    soup = oracle.load_mesh(verts, faces)
    ctx = mvs_amd.Context(W, H)
    ctx.load_mesh(verts, faces)
    return soup, ctx


def _assert_depth(got, ref):
    rmse = np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))
    assert rmse < 1e-4
    np.testing.assert_array_equal(got, ref)


def test_glx_scene_depth_and_projected(oracle):
    W, H = glx_scene.W, glx_scene.H
    soup, ctx = _both(oracle, W, H, glx_scene.POINTS, glx_scene.FACES)
    with ctx:
        _assert_depth(ctx.depth(glx_scene.MVP), oracle.depth(soup, glx_scene.MVP, W, H))
        _assert_depth(ctx.depth(glx_scene.SIDE_MVP), oracle.depth(soup, glx_scene.SIDE_MVP, W, H))
        rng = np.random.default_rng(0)
        frame = rng.integers(0, 256, (H, W), dtype=np.uint8)
        got = ctx.projected(glx_scene.MVP, frame, glx_scene.SIDE_MVP)
        ref = oracle.projected(soup, glx_scene.MVP, frame, glx_scene.SIDE_MVP)
        np.testing.assert_array_equal(got, ref)
        assert (ref[..., 1] == 255).sum() > 1000


@pytest.mark.parametrize("W,H,n", [(160, 120, 32), (333, 211, 64), (640, 480, 128)])
def test_heightfield_mesh(oracle, W, H, n):
    verts, faces = scenes.heightfield_mesh(n)
    soup, ctx = _both(oracle, W, H, verts, faces)
    sc = synth.Scene(freq_scale=0.2)
    main_c, side_c = [0.02, -0.01, 0.0], [0.25, 0.1, 0.05]
    cam, prj = synth.camera_at(main_c, W, H), synth.camera_at(side_c, W, H)
    side_img = sc.render(side_c, W, H)
    with ctx:
        _assert_depth(ctx.depth(cam), oracle.depth(soup, cam, W, H))
        got = ctx.projected(cam, side_img, prj)
        ref = oracle.projected(soup, cam, side_img, prj)
        np.testing.assert_array_equal(got, ref)


def test_random_triangle_soup_and_face_camera(oracle):
    """overlapping random triangles (visibility order, ties) and a camera sitting on the mesh with near = 0.001
    (heuristic.cpp:193-247): triangles cross the camera plane"""
    rng = np.random.default_rng(11)
    W, H = 320, 240
    verts, faces = scenes.random_triangles(rng, 300)
    soup, ctx = _both(oracle, W, H, verts, faces)
    with ctx:
        cam = synth.camera_at([0, 0, 0], W, H)
        _assert_depth(ctx.depth(cam), oracle.depth(soup, cam, W, H))
        # camera placed at a vertex of the soup, looking along -z with a tiny near plane
        v = verts[17, :3] / verts[17, 3]
        K = np.array([[0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, (0.001 + 10) / (10 - 0.001), 2 * 0.001 * 10 / (0.001 - 10)],
                      [0, 0, 1, 0]], np.float64)
        RT = np.eye(4)
        RT[:3, 3] = -v
        face_cam = (K @ RT).astype(np.float32)
        d_ref = oracle.depth(soup, face_cam, W, H)
        _assert_depth(ctx.depth(face_cam), d_ref)
        assert (d_ref != 1.0).sum() > 100
        frame = rng.integers(0, 256, (H, W), dtype=np.uint8)
        prj = synth.camera_at([0.4, 0.2, 0.3], W, H)
        np.testing.assert_array_equal(ctx.projected(cam, frame, prj), oracle.projected(soup, cam, frame, prj))


def test_binned_rasteriser_on_a_large_mesh(oracle):
    """meshes of >= 16384 faces are rasterised through 64x64-pixel face bins (per-bin lists + a shared list of large faces):
    depth, projected and a camera ON the mesh (many faces across w = 0 -> the shared list) against the oracle, ragged size"""
    W, H = 333, 211
    verts, faces = scenes.heightfield_mesh(150)           # 44 402 faces
    assert faces.shape[0] >= 16384
    soup, ctx = _both(oracle, W, H, verts, faces)
    sc = synth.Scene(freq_scale=0.2)
    main_c, side_c = [0.02, -0.01, 0.0], [0.25, 0.1, 0.05]
    cam, prj = synth.camera_at(main_c, W, H), synth.camera_at(side_c, W, H)
    side_img = sc.render(side_c, W, H)
    with ctx:
        _assert_depth(ctx.depth(cam), oracle.depth(soup, cam, W, H))
        np.testing.assert_array_equal(ctx.projected(cam, side_img, prj), oracle.projected(soup, cam, side_img, prj))
        v = verts[faces[20000, 0], :3] / verts[faces[20000, 0], 3]
        K = np.array([[0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, (0.001 + 10) / (10 - 0.001), 2 * 0.001 * 10 / (0.001 - 10)],
                      [0, 0, 1, 0]], np.float64)
        RT = np.eye(4)
        RT[:3, 3] = -v
        RT = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float64) @ RT   # look along the surface
        face_cam = (K @ RT).astype(np.float32)
        d_ref = oracle.depth(soup, face_cam, W, H)
        _assert_depth(ctx.depth(face_cam), d_ref)
        assert (d_ref != 1.0).sum() > 100
        rows = np.array([0, H // 2, H - 1], np.int32)
        cols = np.array([0, W // 2, W - 1], np.int32)
        np.testing.assert_array_equal(ctx.depth_probe(face_cam, rows, cols), d_ref[rows, cols])


def test_empty_mesh_and_errors(oracle):
    W, H = 64, 48
    with mvs_amd.Context(W, H) as ctx:
        with pytest.raises(mvs_amd.MvsError):
            ctx.depth(np.eye(4, dtype=np.float32))  # no mesh loaded
        ctx.load_mesh(np.zeros((0, 4), np.float32), np.zeros((0, 3), np.int32))
        assert np.all(ctx.depth(synth.camera_at([0, 0, 0], W, H)) == np.float32(1.0))
        out = ctx.projected(synth.camera_at([0, 0, 0], W, H), np.full((H, W), 9, np.uint8), synth.camera_at([1, 0, 0], W, H))
        assert not out.any()
        with pytest.raises(mvs_amd.MvsError):
            ctx.load_mesh(np.ones((3, 4), np.float32), np.array([[0, 1, 7]], np.int32))  # bad index


def test_depth_probe_equals_depth_map(oracle):
    """mvs_depth_probe: the pixels Heuristic::filterCameras reads (heuristic.cpp:307-312) without moving the map"""
    W, H = 320, 240
    verts, faces = scenes.heightfield_mesh(48)
    soup, ctx = _both(oracle, W, H, verts, faces)
    cam = synth.camera_at([0.1, -0.05, 0.0], W, H)
    rng = np.random.default_rng(4)
    rows = np.concatenate([rng.integers(0, H, 500), [0, H - 1, 0, H - 1]]).astype(np.int32)
    cols = np.concatenate([rng.integers(0, W, 500), [0, 0, W - 1, W - 1]]).astype(np.int32)
    with ctx:
        depth = ctx.depth(cam)
        got = ctx.depth_probe(cam, rows, cols)
        np.testing.assert_array_equal(got, depth[rows, cols])
        np.testing.assert_array_equal(depth, oracle.depth(soup, cam, W, H))
        assert ctx.depth_probe(cam, [], []).shape == (0,)
        for bad in ((H, 0), (-1, 0), (0, W)):
            with pytest.raises(mvs_amd.MvsError):
                ctx.depth_probe(cam, [bad[0]], [bad[1]])


def test_mix_background(oracle):
    rng = np.random.default_rng(2)
    W, H = 200, 100
    img3 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img3[..., 1] = np.where(rng.random((H, W)) < 0.4, 0, 255)
    bg = rng.integers(0, 256, (H, W), dtype=np.uint8)
    depth = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.3] = 1.0
    ref_out, ref_depth = oracle.mix_background(img3, bg, depth)
    with mvs_amd.Context(W, H) as ctx:
        out, d2 = ctx.mix_background(img3, bg, depth)
    np.testing.assert_array_equal(out, ref_out)
    np.testing.assert_array_equal(d2, ref_depth)


def test_reference_stage_per_pair(oracle):
    """the stage recon.cpp:70,85-86 runs per (main, side) pair: depth -> projected -> mixBackground"""
    W, H = 320, 240
    verts, faces = scenes.heightfield_mesh(64, extent=1.0)  # mesh covers only the centre: background pixels exist
    soup, ctx = _both(oracle, W, H, verts, faces)
    sc = synth.Scene(freq_scale=0.2)
    main_c, side_c = [0.0, 0.0, 0.0], [0.15, 0.0, 0.0]
    cam, prj = synth.camera_at(main_c, W, H), synth.camera_at(side_c, W, H)
    main_img, side_img = sc.render(main_c, W, H), sc.render(side_c, W, H)
    with ctx:
        depth = ctx.depth(cam)
        proj = ctx.projected(cam, side_img, prj)
        mixed, depth2 = ctx.mix_background(proj, main_img, depth)
    d_ref = oracle.depth(soup, cam, W, H)
    p_ref = oracle.projected(soup, cam, side_img, prj)
    m_ref, d2_ref = oracle.mix_background(p_ref, main_img, d_ref)
    np.testing.assert_array_equal(mixed, m_ref)
    np.testing.assert_array_equal(depth2, d2_ref)
    assert (depth2 == 1.0).mean() > 0.2 and (depth2 != 1.0).mean() > 0.1


@pytest.mark.parametrize("sampler", ["fixed", "exact"])
def test_parity_hook_warp_by_depth(oracle, sampler):
    """mvs_warp_by_depth == oracle (bit-exact) and ~= Render::projected on unoccluded geometry; with the
    fixed sampler (1/32-texel positions, 8-bit weights) the warped grey level may differ from projected()'s f32 bilinear by one"""
    W, H = 320, 240
    verts, faces = scenes.heightfield_mesh(96)
    soup, ctx = _both(oracle, W, H, verts, faces)
    sc = synth.Scene(freq_scale=0.2)
    main_c, side_c = [0.0, 0.0, 0.0], [0.2, 0.1, 0.0]
    cam, prj = synth.camera_at(main_c, W, H), synth.camera_at(side_c, W, H)
    side_img = sc.render(side_c, W, H)
    with ctx:
        ctx.set_sampler(sampler)
        depth = ctx.depth(cam)
        warp = ctx.warp_by_depth(cam, depth, prj, side_img)
        ctx.set_texture_filter("level0")   # the sweep's samplers fetch level 0: the hook compares like with like (the mip-mapped
        proj = ctx.projected(cam, side_img, prj)  # projected() departs from it wherever the footprint exceeds one texel)
    np.testing.assert_array_equal(warp, oracle.warp_by_depth(cam, depth, prj, side_img, sampler=sampler))
    both = (proj[..., 1] == 255) & (warp[..., 1] == 255)
    diff = np.abs(proj[..., 0].astype(int) - warp[..., 0].astype(int))[both]
    print("sampler %s: |warp - projected| <= %d grey levels, equal in %.4f of the pixels" % (sampler, diff.max(), np.mean(diff == 0)))
    assert both.mean() > 0.8 and diff.max() <= (1 if sampler == "exact" else 2) and np.mean(diff == 0) > (0.97 if sampler == "exact" else 0.80)


@pytest.mark.parametrize("zoom", [1.0, 1.5, 2.0, 3.3])
def test_projected_with_mipmaps_under_minification(oracle, zoom):
    """the frame texture's mip chain (render_glx.cpp:83-85): a projector whose field of view is `zoom` times narrower than the main
    camera's sees the surface `zoom` times denser, so a main-view pixel covers ~zoom texels of the side frame -- the footprint picks
    levels 0 / 0-1 / 1 / 1-2 of the chain.  HIP == oracle byte for byte with the mip chain and with level 0 only; the two filters agree
    at 1 : 1 and part ways under minification (on noise: the worst case)"""
    W, H = 320, 208
    verts, faces = scenes.heightfield_mesh(48)
    soup, ctx = _both(oracle, W, H, verts, faces)
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, (H, W), dtype=np.uint8)
    cam = synth.camera_at([0.02, -0.01, 0.0], W, H)
    prj = synth.camera_at([0.05, 0.02, 0.0], W, H, fovx=synth.FOVX / zoom)
    with ctx:
        got = ctx.projected(cam, frame, prj)
        ref = oracle.projected(soup, cam, frame, prj, mipmap=True)
        np.testing.assert_array_equal(got, ref)
        ctx.set_texture_filter("level0")
        got0 = ctx.projected(cam, frame, prj)
        ref0 = oracle.projected(soup, cam, frame, prj, mipmap=False)
        np.testing.assert_array_equal(got0, ref0)
        ctx.set_texture_filter("mipmap")
        np.testing.assert_array_equal(ctx.projected(cam, frame, prj), ref)
    valid = (ref[..., 1] == 255) & (ref0[..., 1] == 255)
    assert valid.sum() > 2000 and np.array_equal(ref[..., 1], ref0[..., 1])
    diff = np.abs(ref[..., 0].astype(int) - ref0[..., 0].astype(int))[valid]
    print("zoom %.1f: mean |mipmap - level0| = %.2f grey levels on noise, %.1f %% of the pixels differ" % (zoom, diff.mean(), 100.0 * (diff > 0).mean()))
    if zoom == 1.0:
        assert diff.mean() < 2.0          # footprints within a few per cent of one texel: level 0 almost everywhere
    if zoom >= 2.0:
        assert diff.mean() > 10.0         # this is where the reference's mip-mapping matters
