"""alphaShapeFaces without CGAL (host/alpha_shapes.cpp; alpha_shapes.cpp:36-99 of the reference): its own exact-predicate Delaunay
triangulator against scipy's Qhull, the alpha complex / optimal alpha / face orientation against the numpy restatement of CGAL's
definitions in oracle/meshing_oracle.py.  Host code: no GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import meshing_common as mc  # noqa: E402
import meshing_oracle as mo  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    return mc.host_lib()


def _cells(c):
    return set(map(tuple, np.sort(np.asarray(c), 1).tolist()))


def _shell(rng, n, noise):
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    return (u * (1.0 + noise * rng.normal(size=(n, 1)))).astype(np.float32)


@pytest.mark.parametrize("n,seed", [(5, 0), (40, 1), (400, 2), (3000, 3)])
def test_delaunay_cells_equal_qhulls_and_alpha_shape_equals_the_oracle(lib, n, seed):
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    faces, alpha, comps = mc.alpha_shape(lib, pts)
    of, oa, oc, ocells = mo.alpha_shape(pts)
    assert _cells(mc.delaunay_cells(lib, pts)) == _cells(ocells)
    assert alpha == oa and comps == oc == 1
    assert mc.canonical_faces(faces) == mc.canonical_faces(of)


def test_surface_samples_make_a_closed_outward_oriented_shell(lib):
    rng = np.random.default_rng(7)
    pts = _shell(rng, 4000, 0.01)
    faces, alpha, comps = mc.alpha_shape(lib, pts)
    of, oa, _, _ = mo.alpha_shape(pts)
    assert alpha == oa and comps == 1 and mc.canonical_faces(faces) == mc.canonical_faces(of)
    use = mc.edge_use(faces)
    assert all(use[(b, a)] == c for (a, b), c in use.items())            # closed: every directed edge has its reverse
    assert mc.signed_volume(pts, faces) > 0                               # normals out of the solid (alpha_shapes.cpp:91-95)
    # a larger alpha fills the ball: one outer boundary, volume close to the unit ball's
    faces2, _, comps2 = mc.alpha_shape(lib, pts, forced=4.0)
    assert comps2 == 1 and abs(mc.signed_volume(pts, faces2) - 4.0 / 3.0 * np.pi) < 0.15
    assert mc.canonical_faces(faces2) == mc.canonical_faces(mo.alpha_shape(pts, forced_alpha=4.0)[0])


def test_duplicates_homogeneous_rows_and_wide_exponent_ranges(lib):
    rng = np.random.default_rng(11)
    pts = rng.normal(size=(300, 3)).astype(np.float32)
    dup = np.vstack([pts, pts[:80]])                                      # the later row takes the index over (alpha_shapes.cpp:49)
    f, a, _ = mc.alpha_shape(lib, dup)
    of, oa, _, _ = mo.alpha_shape(dup)
    assert a == oa and mc.canonical_faces(f) == mc.canonical_faces(of) and f.max() >= 300
    w = rng.uniform(0.5, 2.0, size=(300, 1)).astype(np.float32)
    hom = np.hstack([pts * w, w]).astype(np.float32)                      # x/w in float, alpha_shapes.cpp:55
    f, a, _ = mc.alpha_shape(lib, hom)
    of, oa, _, _ = mo.alpha_shape(hom)
    assert a == oa and mc.canonical_faces(f) == mc.canonical_faces(of)
    # coordinates over 12 decades (Qhull's floating point is no judge here): the Delaunay property itself, in exact integer arithmetic
    wide = (pts[:60] * np.float32(1e-6)).astype(np.float32)
    wide[:5] *= np.float32(1e6)
    _assert_delaunay_exact(wide, mc.delaunay_cells(lib, wide))


def test_degenerate_inputs(lib):
    rng = np.random.default_rng(5)
    assert len(mc.alpha_shape(lib, rng.normal(size=(3, 3)))[0]) == 0      # fewer than four points
    flat = rng.normal(size=(50, 3)).astype(np.float32)
    flat[:, 2] = 0.25
    assert len(mc.alpha_shape(lib, flat)[0]) == 0                         # one plane: no cells
    line = np.outer(np.arange(10, dtype=np.float32), np.float32([1, 2, 3]))
    assert len(mc.alpha_shape(lib, line)[0]) == 0
    cube = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)   # eight cospherical points
    f, a, c = mc.alpha_shape(lib, cube)
    assert len(f) == 12 and a == 0.75 and c == 1 and abs(mc.signed_volume(cube, f) - 1.0) < 1e-6
    # a lattice (cospherical everywhere): any Delaunay triangulation fills the hull exactly once, with empty circumspheres
    g = np.stack(np.meshgrid(*[np.arange(5.0)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    cells = mc.delaunay_cells(lib, g)
    vol6 = sum(abs(int(round(np.linalg.det((g[c4][1:] - g[c4][0]).astype(np.float64))))) for c4 in cells)
    assert vol6 == 6 * 64
    _assert_delaunay_exact(g[::2], mc.delaunay_cells(lib, g[::2]))
    f, a, c = mc.alpha_shape(lib, g)
    assert c == 1 and abs(mc.signed_volume(g, f) - 64.0) < 1e-4


def _exact_ints(points):
    """float32 rows -> rows of Python ints on one common power-of-two grid (exact)"""
    from fractions import Fraction
    fr = [[Fraction(float(v)) for v in row] for row in np.asarray(points, np.float32)]
    den = max(v.denominator for row in fr for v in row)
    return [[int(v * den) for v in row] for row in fr]


def _assert_delaunay_exact(points, cells):
    """no point strictly inside the circumsphere of any cell, cells non-degenerate: exact integer determinants"""
    pi = _exact_ints(points)
    lift = [row + [sum(v * v for v in row), 1] for row in pi]
    big = 4 * max(abs(v) for row in pi for v in row) + 1
    far = [big, big, big, 3 * big * big, 1]
    assert len(cells) > 0
    for c4 in np.asarray(cells).tolist():
        rows = [lift[i] for i in c4]
        assert _det_int([pi[i] + [1] for i in c4]) != 0
        outside = _det_int(rows + [far])            # the sign "outside the circumsphere" has for this cell's vertex order
        assert outside != 0
        for e in range(len(pi)):
            if e not in c4:
                assert _det_int(rows + [lift[e]]) * outside >= 0


def _det_int(m):
    """exact determinant of a small integer matrix (fraction-free Bareiss)"""
    m = [row[:] for row in m]
    n = len(m)
    sign, prev = 1, 1
    for k in range(n - 1):
        if m[k][k] == 0:
            for r in range(k + 1, n):
                if m[r][k] != 0:
                    m[k], m[r] = m[r], m[k]
                    sign = -sign
                    break
            else:
                return 0
        for i in range(k + 1, n):
            for j in range(k + 1, n):
                m[i][j] = (m[i][j] * m[k][k] - m[i][k] * m[k][j]) // prev
        prev = m[k][k]
    return sign * m[n - 1][n - 1]


def test_the_cpp_mirror_uses_it_for_the_first_iteration():
    import subprocess
    exe = os.path.join(ROOT, "mesh-reconstruction_amd", os.environ.get("MVS_BUILD_VARIANT", ""), "bin", "host_selftest")
    r = subprocess.run([exe, "cpu", os.path.join(ROOT, "tests", "data", "tracks")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cpu selftest: 0 failures" in r.stdout and "alpha shape of the zatisi bundle" in r.stdout


def test_python_binding():
    sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
    import mvs_amd
    cube = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    f, a, c = mvs_amd.alpha_shape_faces(cube)
    assert f.shape == (12, 3) and a == 0.75 and c == 1


def test_golden_vectors_alpha_and_poisson_oracle(lib):
    """tests/golden/meshing_small.npz (tests/golden/make_meshing_golden.py): the C++ alpha shapes and the oracle both reproduce the stored
    faces / alpha / cells; the Poisson oracle reproduces its stored fields (drift guard; the HIP side: tests/test_meshing_gpu.py)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "meshing_small.npz"))
    want = set(map(tuple, g["alpha_faces"].tolist()))
    faces, alpha, comps = mc.alpha_shape(lib, g["alpha_points"])
    assert mc.canonical_faces(faces) == want and np.float32(alpha) == g["alpha"] and comps == int(g["alpha_components"])
    assert _cells(mc.delaunay_cells(lib, g["alpha_points"])) == _cells(g["alpha_cells"])
    of, oa, oc, _ = mo.alpha_shape(g["alpha_points"])
    assert mc.canonical_faces(of) == want and np.float32(oa) == g["alpha"]
    G, origin, h = mo.poisson_grid(g["poisson_points"], 5)
    assert G == int(g["poisson_G"]) and np.array_equal(origin, g["poisson_origin"]) and np.float32(h) == g["poisson_h"]
    splat = mo.poisson_splat(g["poisson_points"], g["poisson_normals"], G, origin, h)
    assert np.array_equal(splat, g["poisson_splat"])
    chi = mo.poisson_chi(splat, 1.0)
    np.testing.assert_allclose(chi, g["poisson_chi"], rtol=0, atol=2e-6 * float(np.ptp(g["poisson_chi"])))
    v, f = mo.surface_nets(g["poisson_chi"], g["poisson_level"], origin, h)
    assert np.array_equal(f, g["poisson_faces"]) and np.array_equal(v, g["poisson_vertices"])


def test_surface_nets_patch_table_and_what_it_buys():
    """csrc/poisson.hip: cell_patch_table (host code, the kernels' table) == oracle/meshing_oracle.py: cell_components; every patch is a closed
    loop of at least 3 crossings; complementary patterns cross the same edges.  And what one vertex per PATCH instead of per cell buys: a
    plate thinner than a cell, two sheets through the same cells, comes out as two sheets -- no edge with more than two facets (one vertex
    per cell welds them: the oracle's previous rule is restated inline for the comparison)."""
    import ctypes
    import mvs_amd
    hip = mvs_amd.load_library()
    table = (ctypes.c_uint32 * 256)()
    assert hip.mvs_test_cell_patch_table(table) == 0 and hip.mvs_test_cell_patch_table(None) != 0
    comp, count = mo.cell_components()
    assert np.bincount(count).tolist() == [2, 162, 82, 8, 2]
    for m in range(256):
        word = table[m]
        assert (word >> 24) == count[m]
        cross = comp[m] >= 0
        assert np.array_equal(cross, comp[255 - m] >= 0)
        for e in range(12):
            assert ((word >> (2 * e)) & 3) == (comp[m, e] if cross[e] else 0)
        for c in range(count[m]):
            assert (comp[m] == c).sum() >= 3
    G = 32
    z, y, x = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
    chi = (np.abs(z - (6.3 + 0.37 * x + 0.23 * y)) - 0.32).astype(np.float32)
    v, f = mo.surface_nets(chi, 0.0, np.zeros(3, np.float32), 1.0)
    assert mc.facets_per_edge(f) == 2 and max(mc.edge_use(f).values()) == 1 and len(f) == 4502
    # one vertex per cell on the same field: the same facets over fewer vertices, and an edge that four of them share
    inside = chi < 0
    C = G - 1
    case = sum(inside[dz:dz + C, dy:dy + C, dx:dx + C].astype(np.int32) << c for c, (dx, dy, dz) in enumerate([(c & 1, (c >> 1) & 1, c >> 2) for c in range(8)]))
    mixed = (case != 0) & (case != 255)
    first = (np.cumsum(np.where(mixed, count[case], 0).ravel()) - np.where(mixed, count[case], 0).ravel())
    owner = np.repeat(np.arange(mixed.size), np.where(mixed, count[case], 0).ravel())      # vertex -> cell
    assert len(owner) == len(v) and len(np.unique(owner)) == int(mixed.sum()) < len(v)
    welded = owner[f]
    assert mc.facets_per_edge(welded) == 4
