"""End to end, against GEOMETRIC truth (not against the oracle): frames of the zatisi cameras rendered from a known textured surface T,
the pipeline of recon.cpp:65-117 (mvs_process_frame: depth -> projected -> mixBackground -> calculateFlow -> triangulatePixels) given a
proxy mesh P that is NOT T -- the plane T is a bumped copy of -- and the reconstructed points must lie closer to T than P does: the one
property the reference's outer iteration lives on, and a check of every sign and coordinate convention between the stages at once
(each stage alone is bit-identical to its oracle elsewhere in the suite).  Depths compared in the main camera's NDC z, where mvs_depth
renders both surfaces."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _sheets(seq, amplitude):
    """(true surface, proxy plane): the plane through the bundle points' centroid facing the middle camera (scenes.proxy_plane's frame),
    with and without a smooth bump of the given amplitude along its normal"""
    cam_mid = np.asarray(seq.cams[seq.n // 2], np.float64)
    xyz = seq.bundles[:, :3] / seq.bundles[:, 3:4]
    g = xyz.mean(0).astype(np.float64)
    c = np.linalg.svd(cam_mid[[0, 1, 3], :])[2][-1]
    c = c[:3] / c[3]
    nrm = (c - g) / np.linalg.norm(c - g)
    a = np.cross(nrm, [0.0, 0.0, 1.0] if abs(nrm[2]) < 0.9 else [1.0, 0.0, 0.0])
    a /= np.linalg.norm(a)
    b = np.cross(nrm, a)
    ext = 0.6 * np.abs(xyz - g).max()

    def sheet(n, amp):
        t = np.linspace(-ext, ext, n)
        U, V = np.meshgrid(t, t)
        bump = amp * np.sin(2 * np.pi * U / ext * 0.9 + 0.4) * np.cos(2 * np.pi * V / ext * 0.7 - 0.3)
        P = g[None, :] + U.reshape(-1, 1) * a[None, :] + V.reshape(-1, 1) * b[None, :] + bump.reshape(-1, 1) * nrm[None, :]
        verts = np.concatenate([P, np.ones((n * n, 1))], 1).astype(np.float32)
        idx = np.arange(n * n).reshape(n, n)
        q0, q1, q2, q3 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
        return verts, np.concatenate([np.stack([q0, q1, q2], 1), np.stack([q1, q3, q2], 1)]).astype(np.int32)
    amp = amplitude * float(np.linalg.norm(c - g))
    return sheet(160, amp), sheet(48, 0.0)


@pytest.mark.parametrize("amplitude", [0.01, 0.03])
def test_reconstructed_points_move_from_the_proxy_towards_the_surface_the_frames_show(amplitude):
    import c5_common
    import mvs_amd
    seq = c5_common.Sequence()
    W, H = seq.W, seq.H
    (Tv, Tf), (Pv, Pf) = _sheets(seq, amplitude)
    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (H // 4 + 2, W // 4 + 2), dtype=np.uint8).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    tex = (0.7 * np.kron(base, np.ones((4, 4), np.float32))[:H, :W] + 40 + 30 * np.sin(xx / 9.0) * np.cos(yy / 11.0)).clip(0, 255).astype(np.uint8)
    with mvs_amd.Context(W, H) as ctx:
        f = seq.mains[12]
        ids = [f] + seq.sides(f)
        # the sequence's frames: the texture projected onto T from the middle camera, seen from each camera (Render::projected itself)
        ctx.load_mesh(Tv, Tf)
        frames = {}
        for j in ids:
            out = ctx.projected(seq.cams[j], tex, seq.cams[seq.n // 2])
            frames[j] = out[:, :, 0].copy()
            assert (out[:, :, 1] > 0).mean() > 0.25
        z_true = ctx.depth(seq.cams[f]).astype(np.float64)
        ctx.load_mesh(Pv, Pf)
        z_proxy = ctx.depth(seq.cams[f]).astype(np.float64)
        cams = np.stack([seq.cams[j] for j in ids[1:]])
        side = [frames[j] for j in ids[1:]]
        for farneback in (False, True):
            pts = ctx.process_frame(seq.cams[f], frames[f], cams, side, farneback)
            assert len(pts) > 50000
            clip = pts[:, :4].astype(np.float64) @ np.asarray(seq.cams[f], np.float64).T
            clip = clip[np.isfinite(clip).all(1) & (clip[:, 3] != 0.0)]
            ndc = clip[:, :3] / clip[:, 3:4]
            col = np.floor((ndc[:, 0] + 1.0) * 0.5 * W).astype(int)        # SURVEY A-2: xn = (2 col + 1) / W - 1, yn = 1 - (2 row + 1) / H
            row = np.floor((1.0 - ndc[:, 1]) * 0.5 * H).astype(int)
            ok = (col >= 0) & (col < W) & (row >= 0) & (row < H)
            col, row, z = col[ok], row[ok], ndc[ok, 2]
            zt, zp = z_true[row, col], z_proxy[row, col]
            have = (zt < 1.0) & (zp < 1.0)
            off = have & (np.abs(zt - zp) > 0.25 * np.abs(zt - zp)[have].max())   # where the proxy is noticeably wrong
            assert off.sum() > 20000
            e_rec, e_proxy = np.abs(z - zt)[off], np.abs(zp - zt)[off]
            closer = float((e_rec < e_proxy).mean())
            if farneback:
                # Farneback's pyramid follows the parallax of the bump (about a pixel): most of the error goes in ONE pass
                assert closer > 0.95 and np.median(e_rec) < 0.3 * np.median(e_proxy), (amplitude, closer, float(np.median(e_rec)), float(np.median(e_proxy)))
            else:
                # the reference's default (flow.cpp:29: variational refinement from a zero flow) takes a small step in the right direction
                # per outer iteration: right direction nearly everywhere, a few per cent of the error
                assert closer > 0.9 and np.median(e_rec) < 0.99 * np.median(e_proxy), (amplitude, closer, float(np.median(e_rec)), float(np.median(e_proxy)))


@pytest.mark.parametrize("sampler", ["fixed", "exact"])
def test_the_plane_sweep_finds_the_surface_the_frames_show_on_the_tracks_cameras(sampler):
    """the D-plane path on real camera geometry (tracks/zatisi.yaml: rotated, general cameras -- the general kernels) against geometric
    truth: frames rendered from the bumped surface, 32 planes across its depth range, and the selected plane must be the one nearest to
    the rendered depth of that surface (within a plane or two) on the bulk of the pixels all four side views see"""
    import c5_common
    import mvs_amd
    seq = c5_common.Sequence()
    W, H = seq.W, seq.H
    (Tv, Tf), _ = _sheets(seq, 0.03)
    # a texture with a gradient everywhere: the cost is per pixel (shader.frag:22, no window), so flat patches tie over many planes
    import scipy.ndimage as ndi
    rng = np.random.default_rng(10)
    tex = ndi.gaussian_filter(rng.normal(size=(H, W)), 2.5)
    tex = (127.5 + 110.0 * tex / np.abs(tex).max()).clip(0, 255).astype(np.uint8)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        f = seq.mains[12]
        ids = [f] + seq.sides(f)
        ctx.load_mesh(Tv, Tf)
        frames, seen = {}, np.ones((H, W), bool)
        for j in ids:
            out = ctx.projected(seq.cams[j], tex, seq.cams[seq.n // 2])
            frames[j] = out[:, :, 0].copy()
        z_true = ctx.depth(seq.cams[f]).astype(np.float64)
        on = z_true < 1.0
        # pixels of the main view whose surface point every side view sees inside its frame
        Xc = None
        for j in ids[1:]:
            seen &= ctx.projected(seq.cams[f], frames[j], seq.cams[j])[:, :, 1] > 0
        seen &= on & (ctx.projected(seq.cams[f], tex, seq.cams[seq.n // 2])[:, :, 1] > 0)
        assert seen.mean() > 0.15
        lo, hi = float(z_true[on].min()), float(z_true[on].max())
        margin = 0.25 * (hi - lo)
        D = 32
        depth = ctx.sweep(seq.cams[f], frames[f], np.stack([seq.cams[j] for j in ids[1:]]), [frames[j] for j in ids[1:]], D, lo - margin, hi + margin)
        step = (hi - lo + 2 * margin) / D
        err = np.abs(depth.astype(np.float64) - z_true)[seen] / step
        assert np.median(err) <= 1.0 and (err <= 2.0).mean() > 0.8, (sampler, float(np.median(err)), float((err <= 2.0).mean()))


def test_the_outer_iteration_converges_on_the_surface_the_frames_show():
    """recon.cpp:42-136 as a whole, twice round: proxy mesh -> per main frame (depth, projected, mixBackground, Farneback flow, triangulatePixels)
    -> the point blocks concatenated -> filterPoints -> poissonSurface (with the reference's facet criteria) -> loadMesh -> again.  The frames
    are rendered from the bumped surface T, the first proxy is the plane without the bump; after each round the mesh the NEXT round renders
    must lie closer to T (median |z - z_true| in a main camera's NDC z), and still cover what that camera sees of T."""
    import c5_common
    import mvs_amd
    import scipy.ndimage as ndi
    seq = c5_common.Sequence()
    W, H = seq.W, seq.H
    (Tv, Tf), mesh = _sheets(seq, 0.03)
    rng = np.random.default_rng(10)
    tex = ndi.gaussian_filter(rng.normal(size=(H, W)), 2.5)
    tex = (127.5 + 110.0 * tex / np.abs(tex).max()).clip(0, 255).astype(np.uint8)
    with mvs_amd.Context(W, H) as ctx:
        mains = seq.mains[8:16:2]
        need = sorted(set(j for f in mains for j in [f] + seq.sides(f)))
        ctx.load_mesh(Tv, Tf)
        frames = {j: ctx.projected(seq.cams[j], tex, seq.cams[seq.n // 2])[:, :, 0].copy() for j in need}
        check = seq.mains[11]
        z_true = ctx.depth(seq.cams[check]).astype(np.float64)
        errors, cover = [], []
        for it in range(3):
            ctx.load_mesh(*mesh)
            z_mesh = ctx.depth(seq.cams[check]).astype(np.float64)
            both = (z_true < 1.0) & (z_mesh < 1.0)
            errors.append(float(np.median(np.abs(z_mesh - z_true)[both])))
            cover.append(both.sum() / (z_true < 1.0).sum())
            if it == 2:
                break
            blocks = [ctx.process_frame(seq.cams[f], frames[f], np.stack([seq.cams[j] for j in seq.sides(f)]), [frames[j] for j in seq.sides(f)], True) for f in mains]
            cloud = np.concatenate(blocks)
            cloud = cloud[np.isfinite(cloud).all(1)]
            cloud = cloud[::max(1, len(cloud) // 60000)]
            xyz = cloud[:, :3] / cloud[:, 3:4]
            keep = ctx.filter_points(cloud[:, :4], 0.01 * float(np.ptp(xyz, axis=0).max()))
            assert len(keep) > 0.5 * len(cloud)
            rep = {}
            v, f = mvs_amd.poisson_surface(cloud[keep, :4], cloud[keep, 4:7], report=rep)
            assert 1000 < len(v) < 10 * len(keep) and rep["facets_below_angle"] < 0.005 * rep["simplify"]["facets_before"]   # (counted before the simplification pass)
            mesh = (v, f)
        # measured: 0.0224 -> 0.0027 -> 0.0008, coverage 99 % -> 98 % -> 92 %
        assert errors[1] < 0.25 * errors[0] and errors[2] < 0.6 * errors[1], errors
        assert min(cover) > 0.85, cover
