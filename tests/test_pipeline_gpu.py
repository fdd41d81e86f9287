"""mvs_process_frame (the body of recon.cpp:65-117 on device-resident data) == the stage-by-stage ABI calls == the oracle."""
import time

import numpy as np
import pytest

import mvs_amd
import scenes
from mvs_amd import synth

pytestmark = pytest.mark.gpu


def _setup(W, H, nside):
    verts, faces = scenes.heightfield_mesh(64, extent=1.4)
    sc = synth.Scene(freq_scale=max(W / 1920.0, 0.2))
    main_c = [0.0, 0.0, 0.0]
    side_cs = [[0.15, 0.0, 0.0], [-0.1, 0.12, 0.03], [0.02, -0.14, -0.02], [0.1, 0.1, 0.0]][:nside]
    main = synth.camera_at(main_c, W, H)
    sides = np.stack([synth.camera_at(c, W, H) for c in side_cs])
    return verts, faces, main, sides, sc.render(main_c, W, H), [sc.render(c, W, H) for c in side_cs]


@pytest.mark.parametrize("farneback", [False, True])
def test_process_frame_equals_stagewise_and_oracle(oracle, farneback):
    W, H, nside = 320, 240, 3
    verts, faces, main, sides, main_img, side_imgs = _setup(W, H, nside)
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        pts, depth_after = ctx.process_frame(main, main_img, sides, side_imgs, farneback, want_depth=True)
        # the same through the one-stage entry points, as recon.cpp calls them
        depth = ctx.depth(main)
        flows = []
        for cam, img in zip(sides, side_imgs):
            mixed, depth = ctx.mix_background(ctx.projected(main, img, cam), main_img, depth)
            flows.append(ctx.flow(main_img, mixed, farneback))
        ref_pts = ctx.triangulate(flows, main, sides, depth)
    np.testing.assert_array_equal(depth_after, depth)
    np.testing.assert_array_equal(pts, ref_pts)
    assert pts.shape[0] > 0.2 * W * H
    # and against the CPU oracle, stage by stage
    soup = oracle.load_mesh(verts, faces)
    d = oracle.depth(soup, main, W, H)
    oflows = []
    for cam, img in zip(sides, side_imgs):
        mixed, d = oracle.mix_background(oracle.projected(soup, main, img, cam), main_img, d)
        oflows.append(oracle.calculate_flow(main_img, mixed, farneback))
    o_pts = oracle.triangulate_pixels(oflows, main, sides, d)
    np.testing.assert_array_equal(depth_after, d)
    assert pts.shape == o_pts.shape
    np.testing.assert_array_equal(pts[:, :4], o_pts[:, :4])
    ok = np.isfinite(o_pts[:, 4:]).all(1)
    np.testing.assert_allclose(pts[ok, 4:], o_pts[ok, 4:], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("farneback", [False, True])
def test_process_frame_equals_the_oracle_at_1080p(oracle, farneback):
    """recon.cpp:65-117 at c3's frame size (1920 x 1080: Farneback window 30, poly 7 / 3.0; the batched pass over both side views), against the
    oracle stage by stage: depth after mixBackground, every point bit for bit, normals to 1e-5.  The proxy mesh sits 0.08 in front of the surface
    the frames were rendered from and the frames carry sigma = 2 sensor noise, so the flows are real (Farneback: 2-3 px of parallax error)"""
    W, H, nside = 1920, 1080, 2
    verts, faces, main, sides, main_img, side_imgs = _setup(W, H, nside)
    verts = verts.copy()
    verts[:, 2] += 0.08 * verts[:, 3]
    rng = np.random.default_rng(7)

    def noisy(im):
        return (im.astype(np.float64) + rng.normal(0, 2, im.shape)).clip(0, 255).astype(np.uint8)
    main_img, side_imgs = noisy(main_img), [noisy(im) for im in side_imgs]
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        pts, depth_after = ctx.process_frame(main, main_img, sides, side_imgs, farneback, want_depth=True)
    soup = oracle.load_mesh(verts, faces)
    d = oracle.depth(soup, main, W, H)
    oflows = []
    for cam, img in zip(sides, side_imgs):
        mixed, d = oracle.mix_background(oracle.projected(soup, main, img, cam), main_img, d)
        oflows.append(oracle.calculate_flow(main_img, mixed, farneback))
    if farneback:
        assert min(np.median(np.abs(f[100:-100, 100:-100, :2])) for f in oflows) > 0.15
    o_pts = oracle.triangulate_pixels(oflows, main, sides, d)
    np.testing.assert_array_equal(depth_after, d)
    assert pts.shape == o_pts.shape and pts.shape[0] > 0.2 * W * H
    np.testing.assert_array_equal(pts[:, :4], o_pts[:, :4])
    ok = np.isfinite(o_pts[:, 4:]).all(1)
    np.testing.assert_allclose(pts[ok, 4:], o_pts[ok, 4:], rtol=1e-5, atol=1e-9)


def test_process_frame_farneback_batch_at_a_window_the_tiled_iteration_serves():
    """1280 x 720, window 20: the batched Farneback pass of mvs_process_frame (all side views per launch, blockIdx.z = flow) runs the tiled
    iteration kernel -- its per-flow strides included -- and must equal the one-flow-at-a-time entry points, which equal the oracle
    (tests/test_flow_gpu.py)"""
    W, H, nside = 1280, 720, 3
    verts, faces, main, sides, main_img, side_imgs = _setup(W, H, nside)
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        pts = ctx.process_frame(main, main_img, sides, side_imgs, True)
        depth = ctx.depth(main)
        flows = []
        for cam, img in zip(sides, side_imgs):
            mixed, depth = ctx.mix_background(ctx.projected(main, img, cam), main_img, depth)
            flows.append(ctx.flow(main_img, mixed, True))
        ref_pts = ctx.triangulate(flows, main, sides, depth)
    np.testing.assert_array_equal(pts, ref_pts)
    assert pts.shape[0] > 0.2 * W * H


def test_process_frame_is_faster_than_stagewise():
    W, H, nside = 640, 480, 4
    verts, faces, main, sides, main_img, side_imgs = _setup(W, H, nside)
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        ctx.process_frame(main, main_img, sides, side_imgs, False)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.process_frame(main, main_img, sides, side_imgs, False)
        fused = (time.perf_counter() - t0) / 5

        def stagewise():
            depth = ctx.depth(main)
            flows = []
            for cam, img in zip(sides, side_imgs):
                mixed, depth = ctx.mix_background(ctx.projected(main, img, cam), main_img, depth)
                flows.append(ctx.flow(main_img, mixed, False))
            return ctx.triangulate(flows, main, sides, depth)
        stagewise()
        t0 = time.perf_counter()
        for _ in range(5):
            stagewise()
        staged = (time.perf_counter() - t0) / 5
    print("process_frame %.2f ms vs stage-by-stage %.2f ms (640x480, 4 side views)" % (fused * 1e3, staged * 1e3))
    assert fused < staged


@pytest.mark.parametrize("nside", [0, 1, 6, 9])
def test_process_frame_side_view_counts(nside):
    """the side views' flows run concurrently in up to four lanes (stream + arena + compare pyramid each): fewer views than lanes,
    more views than lanes (lanes are reused in order) and repeated calls must all equal the stage-by-stage result"""
    W, H = 256, 160
    verts, faces = scenes.heightfield_mesh(48, extent=1.4)
    sc = synth.Scene(freq_scale=0.2)
    main_c = [0.0, 0.0, 0.0]
    side_cs = [[0.16 * np.cos(0.7 * k), 0.14 * np.sin(0.7 * k), 0.02 * (k % 3 - 1)] for k in range(1, nside + 1)]
    main = synth.camera_at(main_c, W, H)
    sides = np.stack([synth.camera_at(c, W, H) for c in side_cs]) if nside else np.zeros((0, 4, 4), np.float32)
    main_img = sc.render(main_c, W, H)
    side_imgs = [sc.render(c, W, H) for c in side_cs]
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        depth = ctx.depth(main)
        flows = []
        for cam, img in zip(sides, side_imgs):
            mixed, depth = ctx.mix_background(ctx.projected(main, img, cam), main_img, depth)
            flows.append(ctx.flow(main_img, mixed, False))
        ref = ctx.triangulate(flows, main, sides, depth)
        for _ in range(3):      # lanes and events are reused from the second call on
            pts, depth_after = ctx.process_frame(main, main_img, sides, side_imgs, False, want_depth=True)
            np.testing.assert_array_equal(depth_after, depth)
            np.testing.assert_array_equal(pts, ref)


@pytest.mark.parametrize("farneback", [False, True])
def test_process_frame_from_the_frame_store(farneback):
    """mvs_process_frame_slots: the frames of a sequence uploaded once (mvs_frame_upload), every main frame's call naming slots -- the points and
    the depth map of mvs_process_frame on the same frames, bit for bit; slots refilled between calls are seen; bad and empty slots are refused"""
    W, H = 320, 240
    verts, faces = scenes.heightfield_mesh(48, extent=1.4)
    sc = synth.Scene(freq_scale=0.2)
    centres = [[0.05 * k - 0.15, 0.03 * np.sin(k), 0.01 * (k % 3 - 1)] for k in range(7)]
    cams = [synth.camera_at(c, W, H) for c in centres]
    frames = [sc.render(c, W, H) for c in centres]
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        ctx.frame_store(8)
        with pytest.raises(mvs_amd.MvsError):
            ctx.process_frame_slots(cams[2], 2, np.stack([cams[0]]), [0], farneback)       # nothing uploaded yet
        for k, f in enumerate(frames):
            ctx.frame_upload(k, f)
        for main, side in ((2, [0, 1, 3, 4]), (3, [1, 2, 4, 5]), (4, [2, 6]), (5, [])):
            side_cams = np.stack([cams[k] for k in side]) if side else np.zeros((0, 4, 4), np.float32)
            ref, ref_depth = ctx.process_frame(cams[main], frames[main], side_cams, [frames[k] for k in side], farneback, want_depth=True)
            pts, depth = ctx.process_frame_slots(cams[main], main, side_cams, side, farneback, want_depth=True)
            np.testing.assert_array_equal(depth, ref_depth)
            np.testing.assert_array_equal(pts, ref)
        # a slot refilled with another frame is what the next call reads
        ctx.frame_upload(7, frames[0])
        a = ctx.process_frame_slots(cams[2], 2, np.stack([cams[0], cams[1]]), [7, 1], farneback)
        b = ctx.process_frame(cams[2], frames[2], np.stack([cams[0], cams[1]]), [frames[0], frames[1]], farneback)
        np.testing.assert_array_equal(a, b)
        with pytest.raises(mvs_amd.MvsError):
            ctx.process_frame_slots(cams[2], 2, np.stack([cams[0]]), [8], farneback)       # outside the store
        with pytest.raises(mvs_amd.MvsError):
            ctx.process_frame_slots(cams[2], -1, np.stack([cams[0]]), [0], farneback)


def test_stage_matches_the_committed_golden_vectors():
    """every stage of the reference's per-frame path, HIP through the C ABI, against tests/golden/stage_small.npz (written by the
    oracle via tests/golden/make_golden.py and committed): no oracle runs here, so HIP and oracle cannot drift together unnoticed"""
    import os
    import sys
    import zlib
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, golden)
    import make_golden
    g = np.load(os.path.join(golden, "stage_small.npz"))
    W, H, verts, faces, main, sides, main_img, side_imgs = make_golden.stage_inputs()

    def crc(a):
        return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        d = ctx.depth(main)
        np.testing.assert_array_equal(d, g["depth"])
        flows = []
        for k, (cam, img) in enumerate(zip(sides, side_imgs)):
            proj = ctx.projected(main, img, cam)
            assert crc(proj) == g["projected%d_crc" % k]
            mixed, d = ctx.mix_background(proj, main_img, d)
            np.testing.assert_array_equal(mixed, g["mixed%d" % k])
            fv, ff = ctx.flow(main_img, mixed, False), ctx.flow(main_img, mixed, True)
            for name, arr in (("compare", ctx.compare(main_img, mixed)), ("flow_var", fv), ("flow_fb", ff)):
                np.testing.assert_array_equal(arr[::4, ::4], g["%s%d_probe" % (name, k)], err_msg=name)
                assert crc(arr) == g["%s%d_crc" % (name, k)], name
            np.testing.assert_array_equal(ctx.flow_remap(fv, mixed), g["remap%d" % k])
            flows.append(fv)
        np.testing.assert_array_equal(d, g["depth_after"])
        pts = ctx.triangulate(flows, main, sides, d)
        assert pts.shape[0] == int(g["points_n"]) and crc(pts[:, :4]) == g["points_xyzw_crc"]
        probe = pts[::16]
        ok = np.isfinite(g["points_probe"][:, 4:]).all(1)
        np.testing.assert_allclose(probe[ok, 4:], g["points_probe"][ok, 4:], rtol=1e-5, atol=1e-9)
        cloud = pts[:, :4][np.isfinite(pts[:, :4]).all(1)]
        keep = ctx.filter_points(cloud, 0.02)
        assert cloud.shape[0] == int(g["filter_n"]) and len(keep) == int(g["filter_keep_n"]) and crc(np.asarray(keep, np.int32)) == g["filter_keep_crc"]
        # and the one-call form of the same stage
        one = ctx.process_frame(main, main_img, sides, side_imgs, False)
        assert one.shape[0] == int(g["points_n"]) and crc(one[:, :4]) == g["points_xyzw_crc"]


def test_no_scratch_of_one_stage_shows_in_another_stages_result():
    """tests/perf/fuzz_pipeline.py in short: random sequences of loadMesh, depth, projected, calculateFlow, mvs_process_frame, filterPoints and a small
    sweep on one context (the stages share arenas and lanes), every output equal to a fresh context's"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import fuzz_pipeline
    assert sum(fuzz_pipeline.run(seed, 25) for seed in (21, 22)) == 0


def test_contexts_on_host_threads_are_independent():
    """a sequence's main frames are independent (recon.cpp:65): N host threads with a context each on ONE GPU (the library serialises nothing
    between contexts; calls on one context are the caller's to serialise) must return what a single context returns, for both flow algorithms --
    the lanes, arenas and streams of one context must not leak into another's (bench.py's reference_stage block times exactly this)"""
    import threading
    import zlib
    W, H, nside = 320, 240, 3
    verts, faces, main, sides, main_img, side_imgs = _setup(W, H, nside)
    variants = [(main_img, side_imgs), (side_imgs[0], [main_img] + side_imgs[1:]), (side_imgs[1], [side_imgs[0], main_img] + side_imgs[2:])]
    ref = {}
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        for fb in (False, True):
            for v, (m, s) in enumerate(variants):
                ref[fb, v] = zlib.crc32(ctx.process_frame(main, m, sides, s, fb).tobytes())
    bad = []

    def work(k):
        with mvs_amd.Context(W, H) as c:
            c.load_mesh(verts, faces)
            for rep in range(4):
                for fb in ((False, True) if k % 2 else (True, False)):
                    v = (k + rep) % len(variants)
                    m, s = variants[v]
                    if zlib.crc32(c.process_frame(main, m, sides, s, fb).tobytes()) != ref[fb, v]:
                        bad.append((k, rep, fb, v))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad
