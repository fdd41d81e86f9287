"""CPU tests of the triangulatePixels() restatement (util.cpp:44-329)."""
import numpy as np

import scenes
from mvs_amd import synth


def _setup(oracle, W=160, H=120, extent=2.2):
    verts, faces = scenes.heightfield_mesh(96, extent=extent)
    soup = oracle.load_mesh(verts, faces)
    main = synth.camera_at([0, 0, 0], W, H)
    sides = np.stack([synth.camera_at([0.2, 0.05, 0], W, H), synth.camera_at([-0.15, 0.1, 0.02], W, H)])
    return oracle.depth(soup, main, W, H), main, sides


def test_zero_flow_keeps_points_on_the_proxy_surface(oracle):
    """with zero flows the measured points are the proxy's own projections: Newton stays on the proxy (up to the
    reference's sampling quirks, SURVEY A-2/A-14), normals follow the surface and face the cameras"""
    W, H = 160, 120
    depth, main, sides = _setup(oracle, W, H)
    flows = [np.zeros((H, W, 4), np.float32) for _ in sides]
    for f in flows:
        f[..., 2] = 1.0
    pts = oracle.triangulate_pixels(flows, main, sides, depth)
    assert pts.shape[1] == 7 and 0.95 * W * H < pts.shape[0] <= W * H
    xyz = pts[:, :3] / pts[:, 3:4]
    err = np.abs(xyz[:, 2] - synth.Scene.height(xyz[:, 0], xyz[:, 1]))
    assert err.max() < 0.03                       # about one pixel footprint (2.76/160) times the slope, plus sag
    n = pts[:, 4:7]
    nn = np.linalg.norm(n, axis=1)
    assert np.all(np.isfinite(n)) and nn.min() > 0
    assert np.mean(n[:, 2] / nn > 0.8) > 0.95     # cameras sit at z = 0 above the surface (z ~ -3): normals point +z
    assert np.allclose(nn, nn[0], rtol=0.05)      # |n| = pdf^(1/V): uniform variance -> uniform scale


def test_background_and_behind_camera_pixels_are_dropped(oracle):
    W, H = 96, 64
    depth, main, sides = _setup(oracle, W, H, extent=0.8)   # mesh covers only the centre
    nbg = int((depth == 1.0).sum())
    assert nbg > 500
    flows = [np.zeros((H, W, 4), np.float32) for _ in sides]
    for f in flows:
        f[..., 2] = 2.0
    pts = oracle.triangulate_pixels(flows, main, sides, depth)
    assert pts.shape[0] <= W * H - nbg
    # a side camera that sees the scene at NDC z < -1 (its near plane is beyond the surface) rejects every pixel
    far_side = synth.camera_at([0.2, 0, 0], W, H, near=5.0, far=9.0)
    pts2 = oracle.triangulate_pixels(flows[:1], main, far_side[None], depth)
    assert pts2.shape[0] == 0


def test_flow_moves_points_in_depth(oracle):
    """a horizontal flow towards a side camera displaced in +x changes the triangulated depth in a consistent direction"""
    W, H = 128, 96
    depth, main, sides = _setup(oracle, W, H)
    base = [np.zeros((H, W, 4), np.float32) for _ in sides]
    for f in base:
        f[..., 2] = 1.0
    p0 = oracle.triangulate_pixels(base, main, sides, depth)
    shifted = [f.copy() for f in base]
    shifted[0][..., 0] = 1.0
    p1 = oracle.triangulate_pixels(shifted, main, sides, depth)
    assert p0.shape == p1.shape
    dz = (p1[:, 2] / p1[:, 3]) - (p0[:, 2] / p0[:, 3])
    assert np.abs(np.median(dz)) > 1e-3 and np.mean(np.sign(dz) == np.sign(np.median(dz))) > 0.9
