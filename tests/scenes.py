"""Test scenes for the renderer path (numpy only)."""
import numpy as np

from mvs_amd import synth


def heightfield_mesh(n=48, extent=2.2):
    """triangulated grid of synth.Scene.height over [-extent, extent]^2 (world coordinates, w = 1)"""
    xs = np.linspace(-extent, extent, n)
    X, Y = np.meshgrid(xs, xs)
    Z = synth.Scene.height(X, Y)
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel(), np.ones(n * n)], 1).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int32)
    return verts, faces


def random_triangles(rng, n, zrange=(-6.0, -2.0), spread=2.0):
    c = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(*zrange, n)], 1)
    verts = (c[:, None, :] + rng.normal(0, 0.5, (n, 3, 3))).reshape(-1, 3)
    verts = np.concatenate([verts, np.ones((3 * n, 1))], 1).astype(np.float32)
    # homogeneous scale must not matter (loadMesh dehomogenises, render_glx.cpp:242-244)
    verts *= rng.uniform(0.5, 2.0, (3 * n, 1)).astype(np.float32)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    return verts, faces


def proxy_plane(bundles4, camera, n=48, scale=1.5):
    """a flat n x n proxy mesh through the centroid of the reconstructed bundle points, facing the given camera --
    the kind of first-iteration mesh the reference hands to the renderer (Heuristic::tessellate on sparse points)"""
    xyz = bundles4[:, :3] / bundles4[:, 3:4]
    g = xyz.mean(0).astype(np.float64)
    M = np.asarray(camera, np.float64)[[0, 1, 3], :]          # centre = null vector of rows 0, 1, 3 (util.cpp:33-41)
    c = np.linalg.svd(M)[2][-1]
    c = c[:3] / c[3]
    nrm = (c - g) / np.linalg.norm(c - g)
    a = np.cross(nrm, [0.0, 0.0, 1.0] if abs(nrm[2]) < 0.9 else [1.0, 0.0, 0.0])
    a /= np.linalg.norm(a)
    b = np.cross(nrm, a)
    ext = scale * np.abs(xyz - g).max()
    t = np.linspace(-ext, ext, n)
    U, V = np.meshgrid(t, t)
    P = g[None, :] + U.reshape(-1, 1) * a[None, :] + V.reshape(-1, 1) * b[None, :]
    verts = np.concatenate([P, np.ones((n * n, 1))], 1).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    q0, q1, q2, q3 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([q0, q1, q2], 1), np.stack([q1, q3, q2], 1)]).astype(np.int32)
    return verts, faces
