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
