"""Regenerates tests/golden/meshing_small.npz from oracle/meshing_oracle.py (run in the dev container):
    python tests/golden/make_meshing_golden.py
Like the other golden files these pin the restatement against silent drift, not against the reference (CGAL / PCL hold no vectors
for these functions in the reference tree)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import meshing_oracle as mo  # noqa: E402


def canonical(f):
    f = np.asarray(f).reshape(-1, 3)
    k = np.argmin(f, 1)
    r = np.arange(len(f))
    g = np.stack([f[r, (k + i) % 3] for i in range(3)], 1)
    return g[np.lexsort((g[:, 2], g[:, 1], g[:, 0]))].astype(np.int32)


def sorted_rows(c):
    c = np.sort(np.asarray(c), 1)
    return c[np.lexsort(c.T[::-1])].astype(np.int32)


def main():
    rng = np.random.default_rng(0x5EED0F4)
    # alpha shapes: a noisy shell with a few interior points and duplicates
    u = rng.normal(size=(260, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    pts = np.vstack([u * (1.0 + 0.03 * rng.normal(size=(260, 1))), rng.normal(size=(30, 3)) * 0.2]).astype(np.float32)
    pts = np.vstack([pts, pts[:10]])
    faces, alpha, comps, cells = mo.alpha_shape(pts)
    # Poisson: an ellipsoid, 1500 samples, 32^3 nodes
    v = rng.normal(size=(1500, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    radii = np.array([1.0, 0.7, 0.5])
    xyz = v * radii + np.array([0.2, -0.1, 0.4])
    nrm = v / radii
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    ppts = np.hstack([xyz, np.ones((1500, 1))]).astype(np.float32)
    pnrm = nrm.astype(np.float32)
    G, origin, h = mo.poisson_grid(ppts, 5)
    splat = mo.poisson_splat(ppts, pnrm, G, origin, h)
    chi = mo.poisson_chi(splat, 1.0)
    iso = mo.poisson_level(chi, G, origin, h, ppts[:, :3] / ppts[:, 3:4])
    verts, pfaces = mo.surface_nets(chi.astype(np.float32), np.float32(iso), origin, h)
    # the facet criteria of cgal_poisson.cpp:50-52 on that mesh (average spacing of the 1500 samples; 20 degrees, 300 and 0.375 spacings)
    spacing = np.float32(mo.average_spacing(ppts))
    cverts, cfaces, crep = mo.enforce_facet_criteria(verts, pfaces, 20.0, 300.0 * float(spacing), 0.375 * float(spacing))
    np.savez_compressed(os.path.join(HERE, "meshing_small.npz"), criteria_spacing=spacing, criteria_vertices=cverts, criteria_faces=cfaces,
                        criteria_ops=np.array([crep["collapses"], crep["flips"], crep["facets_below_angle"]], np.int32), alpha_points=pts, alpha_faces=canonical(faces), alpha=np.float32(alpha), alpha_components=comps,
                        alpha_cells=sorted_rows(cells), poisson_points=ppts, poisson_normals=pnrm, poisson_G=G, poisson_origin=origin,
                        poisson_h=np.float32(h), poisson_splat=splat, poisson_chi=chi.astype(np.float32), poisson_level=np.float32(iso), poisson_vertices=verts, poisson_faces=pfaces)
    print("wrote meshing_small.npz: %d alpha faces (alpha %g), %d Poisson vertices, %d faces; criteria pass: %s" % (len(faces), alpha, len(verts), len(pfaces), crep))


if __name__ == "__main__":
    main()
