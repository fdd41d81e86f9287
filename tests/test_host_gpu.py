"""The reference's per-pair stage driven through the C++ mirror exactly as recon.cpp:21,42,46,65-89 does it
(spawnRender -> loadMesh -> chooseCameras -> depth -> projected -> mixBackground -> compare / flowRemap), outputs
checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "mesh-reconstruction_amd", os.environ.get("MVS_BUILD_VARIANT", ""), "bin", "host_selftest")
TRACKS = os.path.join(ROOT, "tests", "data", "tracks")


def test_recon_stage_through_cpp_mirror(oracle, tmp_path):
    r = subprocess.run([SELFTEST, "gpu", TRACKS, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu selftest: 0 failures" in r.stdout
    fa, fb, W, H, F = [int(x) for x in open(tmp_path / "meta.txt").read().split()]

    def raw(name, dtype, shape):
        return np.fromfile(tmp_path / name, dtype).reshape(shape)

    verts = raw("mesh_verts.f32", np.float32, (-1, 4))
    faces = raw("mesh_faces.i32", np.int32, (F, 3))
    cam_a, cam_b = raw("cam_a.f32", np.float32, (4, 4)), raw("cam_b.f32", np.float32, (4, 4))
    frame_a, frame_b = raw("frame_a.u8", np.uint8, (H, W)), raw("frame_b.u8", np.uint8, (H, W))
    soup = oracle.load_mesh(verts, faces)
    d_ref = oracle.depth(soup, cam_a, W, H)
    np.testing.assert_array_equal(raw("depth.f32", np.float32, (H, W)), d_ref)
    p_ref = oracle.projected(soup, cam_a, frame_b, cam_b)
    np.testing.assert_array_equal(raw("projected.u8", np.uint8, (H, W, 3)), p_ref)
    m_ref, d2_ref = oracle.mix_background(p_ref, frame_a, d_ref)
    np.testing.assert_array_equal(raw("mixed.u8", np.uint8, (H, W)), m_ref)
    np.testing.assert_array_equal(raw("depth_after_mix.f32", np.float32, (H, W)), d2_ref)
    np.testing.assert_array_equal(raw("compare.f32", np.float32, (H, W)), oracle.compare(frame_a, m_ref))
    flow = raw("flow.f32", np.float32, (H, W, 4))
    np.testing.assert_array_equal(raw("remap.u8", np.uint8, (H, W)), oracle.flow_remap(flow, m_ref))
    # recon.cpp:89-116 through the mirror: calculateFlow per side view + triangulatePixels
    used, nrows = [int(x) for x in open(tmp_path / "tri_meta.txt").read().split()]
    tri = raw("tri.f32", np.float32, (nrows, 7))
    tflows = [raw("tri_flow%d.f32" % i, np.float32, (H, W, 4)) for i in range(used)]
    tcams = np.stack([raw("tri_cam%d.f32" % i, np.float32, (4, 4)) for i in range(used)])
    t_ref = oracle.triangulate_pixels(tflows, cam_a, tcams, raw("tri_depth.f32", np.float32, (H, W)))
    assert tri.shape == t_ref.shape
    np.testing.assert_array_equal(tri[:, :4], t_ref[:, :4])
    ok = np.isfinite(t_ref[:, 4:]).all(1)
    np.testing.assert_allclose(tri[ok, 4:], t_ref[ok, 4:], rtol=1e-5, atol=1e-9)
    # camera selection produced a usable, sorted schedule (heuristic.cpp:484)
    mains = [int(l.split(":")[0]) for l in open(tmp_path / "chosen.txt")]
    assert mains == sorted(mains) and len(mains) >= 1
    assert (d_ref != 1.0).mean() > 0.05


@pytest.mark.parametrize("name,threshold", [("koberec.yaml", 10.0), ("koberec-.yaml", 10.0), ("zatisi.yaml", 4.0), ("koule-tr.yaml", 10.0)])
def test_camera_selection_with_occlusion_matches_the_numpy_restatement(oracle, tmp_path, name, threshold):
    """Heuristic::chooseCameras through RenderHIP (depth look-ups served by mvs_depth_probe) against tests/policy_mirror.py with the
    ORACLE's depth maps, pair for pair, on a bumpy proxy mesh that hides some cameras from some faces (heuristic.cpp:307-314)."""
    import policy_mirror
    import scenes
    import tracks_yaml
    from test_host_cpu import _choose_cpp
    t = tracks_yaml.load(name)
    W, H = t["width"], t["height"]
    cams = [np.asarray(c, np.float32) for c in t["cameras"]]
    verts, faces = scenes.proxy_plane(t["bundles"], cams[len(cams) // 2], n=20, scale=0.6)
    # bumps along the plane normal: ridges tall enough to occlude grazing cameras
    xyz = verts[:, :3].astype(np.float64)
    g = xyz.mean(0)
    a, b = xyz[1] - xyz[0], xyz[20] - xyz[0]
    nrm = np.cross(a, b)
    nrm /= np.linalg.norm(nrm)
    ext = np.abs(xyz - g).max()
    u, v = (xyz - g) @ (a / np.linalg.norm(a)), (xyz - g) @ (b / np.linalg.norm(b))
    verts = verts.copy()
    verts[:, :3] = (xyz + nrm[None, :] * (0.6 * ext * np.sin(7.0 * u / ext) * np.cos(5.0 * v / ext))[:, None]).astype(np.float32)
    soup = oracle.load_mesh(verts, faces)
    blocked = [0]

    def depth_at(viewer, px):
        d = oracle.depth(soup, viewer, W, H)
        vals = [d[r, c] for r, c in px]
        blocked[0] += sum(1 for x in vals if x != policy_mirror.BACKGROUND)
        return vals
    m_count, m_chosen, rng = policy_mirror.choose_cameras(verts, faces, cams, W, H, threshold, depth_at)
    count, chosen, state = _choose_cpp(tmp_path, name, verts, faces, threshold, nodepth=False)
    print("%s: %d depth look-ups hit geometry" % (name, blocked[0]))
    if name == "koberec.yaml":
        assert blocked[0] > 0, "this scene is meant to produce look-ups that hit geometry"
    assert state == rng.state
    assert (count, chosen) == (m_count, m_chosen)


def test_a_clip_of_another_size_is_resized_on_the_way_in(oracle, tmp_path):
    """configuration.cpp:232-233 (cv::resize of every decoded frame that is not the clip's size) behind the YUV4MPEG2 reader"""
    import y4m_common
    y4m_common.run(tmp_path, oracle, scale=2)


def test_the_sweep_behind_the_cpp_seam(tmp_path):
    """RenderHIP::sweepDepth / projectedByDepth and the driver's loop body trackMainFrame (host/recon.hpp, host/driver.cpp) on the scene of
    tests/test_e2e_gpu.py: frames of the zatisi cameras rendered from a bumped surface T, handed to the C++ mirror as a `<clip>.frames`
    directory; the proxy mesh is the plane without the bump.  (a) the swept depth map that comes back through the C++ seam equals mvs_sweep
    through the Python binding, bit for bit (frame store + mvs_sweep_handles against the one-call entry); (b) with --sweep-planes the points
    trackMainFrame returns lie closer to T than the reference path's (proxy z-buffer) after the same single pass -- the outer iteration
    starts where the reference's would be after several rounds."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import c5_common
    import mvs_amd
    import scipy.ndimage as ndi
    from test_e2e_gpu import _sheets
    seq = c5_common.Sequence()
    W, H = seq.W, seq.H
    (Tv, Tf), (Pv, Pf) = _sheets(seq, 0.03)
    rng = np.random.default_rng(10)
    tex = ndi.gaussian_filter(rng.normal(size=(H, W)), 2.5)
    tex = (127.5 + 110.0 * tex / np.abs(tex).max()).clip(0, 255).astype(np.uint8)
    f = seq.mains[12]
    sides = seq.sides(f)
    shutil.copy(os.path.join(TRACKS, "zatisi.yaml"), tmp_path / "zatisi.yaml")
    os.makedirs(tmp_path / "zatisi.avi.frames")
    D = 48
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(Tv, Tf)
        frames = {j: ctx.projected(seq.cams[j], tex, seq.cams[seq.n // 2])[:, :, 0].copy() for j in [f] + sides}
        z_true = ctx.depth(seq.cams[f]).astype(np.float64)
        for j, img in frames.items():
            with open(tmp_path / "zatisi.avi.frames" / ("%06d.pgm" % (j + 1)), "wb") as fh:
                fh.write(b"P5\n%d %d\n255\n" % (W, H) + img.tobytes())
        d_py, c_py = ctx.sweep(seq.cams[f], frames[f], np.stack([seq.cams[j] for j in sides]), [frames[j] for j in sides], D, want_cost=True)
        ctx.load_mesh(Pv, Pf)
        z_proxy = ctx.depth(seq.cams[f]).astype(np.float64)
    Pv.astype(np.float32).tofile(tmp_path / "verts.f32")
    Pf.astype(np.int32).tofile(tmp_path / "faces.i32")
    out = tmp_path / "out"
    os.makedirs(out)
    r = subprocess.run([SELFTEST, "sweep", str(tmp_path / "zatisi.yaml"), str(tmp_path / "verts.f32"), str(tmp_path / "faces.i32"), str(out), str(D), "1", str(f)] +
                       [str(j) for j in sides], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sweep selftest OK" in r.stdout, r.stdout + r.stderr
    # (a) the same maps through both doors
    np.testing.assert_array_equal(np.fromfile(out / "sweep_depth.f32", np.float32).reshape(H, W), d_py)
    np.testing.assert_array_equal(np.fromfile(out / "sweep_cost.f32", np.float32).reshape(H, W), c_py)
    # (b) one pass of the loop body, the reference's way and with the swept depth
    cam = np.asarray(seq.cams[f], np.float64)
    med = {}
    for tag in ("proxy", "swept"):
        n = int(open(out / ("meta_%s.txt" % tag)).read())
        pts = np.fromfile(out / ("points_%s.f32" % tag), np.float32).reshape(n, 7)
        assert n > 50000
        clip = pts[:, :4].astype(np.float64) @ cam.T
        clip = clip[np.isfinite(clip).all(1) & (clip[:, 3] != 0.0)]
        ndc = clip[:, :3] / clip[:, 3:4]
        col = np.floor((ndc[:, 0] + 1.0) * 0.5 * W).astype(int)
        row = np.floor((1.0 - ndc[:, 1]) * 0.5 * H).astype(int)
        ok = (col >= 0) & (col < W) & (row >= 0) & (row < H)
        col, row, z = col[ok], row[ok], ndc[ok, 2]
        zt, zp = z_true[row, col], z_proxy[row, col]
        have = (zt < 1.0) & (zp < 1.0)
        med[tag] = float(np.median(np.abs(z - zt)[have]))
        med[tag + "_proxy_error"] = float(np.median(np.abs(zp - zt)[have]))
        used = np.fromfile(out / ("depth_%s.f32" % tag), np.float32).reshape(H, W)
        assert ((used < 1.0) <= (z_proxy < 1.0)).all()          # the proxy still says WHERE there is a surface
    print("median |z - z_true| after one pass: reference path %.5f, swept depth %.5f (proxy mesh itself %.5f)" % (med["proxy"], med["swept"], med["proxy_proxy_error"]))
    assert med["proxy"] < med["proxy_proxy_error"]       # the reference's path moves towards the surface (tests/test_e2e_gpu.py) ...
    assert med["swept"] < 0.6 * med["proxy"], med        # ... and the swept depth is already there


def test_the_driver_tracks_a_sequence_on_several_threads(tmp_path):
    """trackMainFrames (host/driver.cpp; `--threads N`): the main frames of one outer iteration on N host threads, each with a renderer and the free
    functions' contexts of its own on the one GPU -- the point blocks are the same bytes in the same order as on one thread, for both flow algorithms
    (host_selftest compares them and prints the two times)"""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import c5_common
    import mvs_amd
    import scipy.ndimage as ndi
    from test_e2e_gpu import _sheets
    seq = c5_common.Sequence()
    W, H = seq.W, seq.H
    (Tv, Tf), (Pv, Pf) = _sheets(seq, 0.03)
    rng = np.random.default_rng(10)
    tex = ndi.gaussian_filter(rng.normal(size=(H, W)), 2.5)
    tex = (127.5 + 110.0 * tex / np.abs(tex).max()).clip(0, 255).astype(np.uint8)
    mains = list(range(20, 92, 3))   # 24 main frames with overlapping neighbourhoods: a renderer's frame store serves several main frames per upload
    need = sorted(set(j for f in mains for j in [f] + seq.sides(f)))
    shutil.copy(os.path.join(TRACKS, "zatisi.yaml"), tmp_path / "zatisi.yaml")
    os.makedirs(tmp_path / "zatisi.avi.frames")
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(Tv, Tf)
        for j in need:
            img = ctx.projected(seq.cams[j], tex, seq.cams[seq.n // 2])[:, :, 0].copy()
            with open(tmp_path / "zatisi.avi.frames" / ("%06d.pgm" % (j + 1)), "wb") as fh:
                fh.write(b"P5\n%d %d\n255\n" % (W, H) + img.tobytes())
    Pv.astype(np.float32).tofile(tmp_path / "verts.f32")
    Pf.astype(np.int32).tofile(tmp_path / "faces.i32")
    for fb in ("0", "1"):
        r = subprocess.run([SELFTEST, "sequence", str(tmp_path / "zatisi.yaml"), str(tmp_path / "verts.f32"), str(tmp_path / "faces.i32"), "4", fb] + [str(f) for f in mains],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "sequence selftest OK" in r.stdout, r.stdout + r.stderr
        print(r.stdout.strip().splitlines()[-2])
