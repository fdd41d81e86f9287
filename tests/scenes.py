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
