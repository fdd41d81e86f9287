"""The reference's facet criteria (cgal_poisson.cpp:50-52, 95-97: min angle 20 degrees, facet radius <= 300 average spacings, facet
distance <= 0.375 average spacings) as a pass over the surface-nets mesh: csrc/surface_criteria.cpp through mvs_surface_from_mesh +
mvs_surface_enforce_criteria.  Host code, no GPU: the input meshes are the ORACLE's surface nets (oracle/meshing_oracle.py, the same
mesh csrc/poisson.hip makes, tests/test_meshing_gpu.py), and what is checked is what the criteria state, recomputed here in numpy:
every angle, every circumradius, distances to the analytic surface, plus what the pass promises on top -- the surface stays closed,
oriented and manifold, encloses the same volume, and every vertex is one of the mesher's."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import meshing_common as mc  # noqa: E402
import meshing_oracle as mo  # noqa: E402
import mvs_amd  # noqa: E402


def facet_angles(v, f):
    """all three angles of every facet, degrees, float64"""
    p = np.asarray(v, np.float64)[:, :3]
    a, b, c = p[f[:, 0]], p[f[:, 1]], p[f[:, 2]]

    def ang(x, y, z):
        u, w = y - x, z - x
        return np.degrees(np.arccos(np.clip((u * w).sum(1) / np.sqrt((u * u).sum(1) * (w * w).sum(1)), -1.0, 1.0)))
    return np.stack([ang(a, b, c), ang(b, c, a), ang(c, a, b)], 1)


def circumradii(v, f):
    p = np.asarray(v, np.float64)[:, :3]
    a, b, c = p[f[:, 0]], p[f[:, 1]], p[f[:, 2]]
    la, lb, lc = np.linalg.norm(b - c, axis=1), np.linalg.norm(c - a, axis=1), np.linalg.norm(a - b, axis=1)
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    return la * lb * lc / (4.0 * area)


def closed_oriented_manifold(f):
    use = mc.edge_use(f)
    return all(n == 1 and use[(b, a)] == 1 for (a, b), n in use.items())


def oracle_surface(pts, nrm):
    G, origin, h = mo.poisson_grid(pts, 0)
    chi = mo.poisson_chi(mo.poisson_splat(pts, nrm, G, origin, h), 1.0)
    iso = mo.poisson_level(chi, G, origin, h, pts[:, :3] / pts[:, 3:4])
    v, f = mo.surface_nets(chi, iso, origin, h)
    return v, f, mo.average_spacing(pts)


def sphere(rng, n, noise=0.0):
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = d * (1.0 + noise * rng.normal(size=(n, 1)))
    return np.concatenate([p, np.ones((n, 1))], 1).astype(np.float32), d.astype(np.float32)


def torus(rng, n, R=1.0, r=0.4):
    u, w = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    p = np.stack([(R + r * np.cos(w)) * np.cos(u), (R + r * np.cos(w)) * np.sin(u), r * np.sin(w)], 1)
    nn = np.stack([np.cos(w) * np.cos(u), np.cos(w) * np.sin(u), np.sin(w)], 1)
    return np.concatenate([p, np.ones((n, 1))], 1).astype(np.float32), nn.astype(np.float32)


SURFACES = {
    "sphere": (lambda rng: sphere(rng, 2500), lambda x: np.linalg.norm(x, axis=1) - 1.0),
    "noisy sphere": (lambda rng: sphere(rng, 3000, 0.01), None),
    "torus": (lambda rng: torus(rng, 1800), lambda x: np.sqrt((np.sqrt(x[:, 0] ** 2 + x[:, 1] ** 2) - 1.0) ** 2 + x[:, 2] ** 2) - 0.4),
}


@pytest.mark.parametrize("name", sorted(SURFACES))
def test_the_three_criteria_hold_on_surface_nets_and_the_surface_stays_what_it_was(name):
    make, distance = SURFACES[name]
    pts, nrm = make(np.random.default_rng(11))
    v, f, sp = oracle_surface(pts, nrm)
    before = facet_angles(v, f).min(1)
    assert (before < 20.0).mean() > 0.005 and before.min() < 12.0         # surface nets do make needles and caps: the pass has work to do
    assert closed_oriented_manifold(f)
    v2, f2, rep = mvs_amd.enforce_facet_criteria(v, f, sp)
    # (1) the angle bound, recomputed here
    ang = facet_angles(v2, f2)
    assert ang.min() >= 20.0 - 1e-6 and rep["facets_below_angle"] == 0
    assert abs(rep["min_angle_deg"] - ang.min()) < 1e-3
    # (2) the radius bound: never close (facets span at most a few cells of <= 0.75 spacings each)
    R = circumradii(v2, f2)
    assert R.max() <= 300.0 * sp and rep["facets_above_radius"] == 0 and R.max() < 1.5 * sp
    assert abs(rep["max_circumradius"] - R.max()) <= 1e-5 * R.max()
    # (3) the distance bound: vertices AND facet centres within 0.375 spacings of the analytic surface
    if distance is not None:
        centres = np.asarray(v2, np.float64)[:, :3][f2].mean(1)
        assert np.abs(distance(np.asarray(v2, np.float64)[:, :3])).max() <= 0.375 * sp
        assert np.abs(distance(centres)).max() <= 0.375 * sp
    # what the pass promises besides: same topology class, same solid, no new vertex, order kept
    assert closed_oriented_manifold(f2)
    assert len(v2) - len(f2) // 2 == len(v) - len(f) // 2                   # Euler characteristic (V - E + F with E = 3F/2)
    assert abs(mc.signed_volume(v2, f2) / mc.signed_volume(v, f) - 1.0) < 1e-4
    assert rep["collapses"] == len(v) - len(v2) and len(f) - len(f2) == 2 * rep["collapses"]
    rows = {r.tobytes(): i for i, r in enumerate(v)}
    where = np.array([rows[r.tobytes()] for r in v2])
    assert np.all(np.diff(where) > 0)                                        # a subsequence of the mesher's vertices
    assert rep["flips"] > 0 and rep["collapses"] > 0
    # the oracle's restatement of the pass (pure Python, operation for operation): the same mesh index for index, the same counts
    ov, of, orep = mo.enforce_facet_criteria(v, f, 20.0, 300.0 * sp, 0.375 * sp)
    assert np.array_equal(ov, v2) and np.array_equal(of, f2)
    assert all(orep[k] == rep[k] for k in ("collapses", "flips", "facets_below_angle", "facets_above_radius"))
    assert abs(orep["min_angle_deg"] - rep["min_angle_deg"]) < 1e-4 and abs(orep["max_circumradius"] - rep["max_circumradius"]) <= 1e-6 * orep["max_circumradius"]
    # idempotent and deterministic
    v3, f3, rep3 = mvs_amd.enforce_facet_criteria(v2, f2, sp)
    assert rep3["collapses"] == rep3["flips"] == 0 and np.array_equal(v3, v2) and np.array_equal(f3, f2)
    v4, f4, _ = mvs_amd.enforce_facet_criteria(v, f, sp)
    assert v4.tobytes() == v2.tobytes() and f4.tobytes() == f2.tobytes()


def _flat_grid(n, jitter=None):
    """n x n vertices on the plane z = 0, unit spacing, two triangles per square, closed into a pillow by a mirrored back side so that
    every edge has two facets"""
    ys, xs = np.mgrid[0:n, 0:n]
    top = np.stack([xs.ravel(), ys.ravel(), np.zeros(n * n)], 1).astype(np.float64)
    if jitter is not None:
        top[:, :2] += jitter
    idx = lambda x, y: y * n + x  # noqa: E731
    f = []
    for y in range(n - 1):
        for x in range(n - 1):
            f += [[idx(x, y), idx(x + 1, y), idx(x + 1, y + 1)], [idx(x, y), idx(x + 1, y + 1), idx(x, y + 1)]]
    f = np.array(f, np.int32)
    # back side: interior vertices duplicated slightly below, border shared (the two corner facets that have no interior vertex would
    # coincide with their mirror images: left out, their corner vertices stay unused)
    border = (xs.ravel() == 0) | (ys.ravel() == 0) | (xs.ravel() == n - 1) | (ys.ravel() == n - 1)
    f = f[~border[f].all(1)]
    back_id = np.where(border, np.arange(n * n), n * n + np.cumsum(~border) - 1)
    back = top[~border] - np.array([0.0, 0.0, 0.5])
    v = np.concatenate([top, back])
    fb = back_id[f][:, ::-1]
    v4 = np.concatenate([v, np.ones((len(v), 1))], 1).astype(np.float32)
    return v4, np.concatenate([f, fb]).astype(np.int32)


def test_a_needle_is_collapsed_and_a_cap_is_flipped():
    # needle: two neighbouring vertices of the top sheet almost coincide
    n = 7
    jitter = np.zeros((n * n, 2))
    jitter[3 * n + 3] = [0.97, 0.0]                # vertex (3, 3) moved next to (4, 3)
    v, f = _flat_grid(n, jitter)
    assert closed_oriented_manifold(f) and facet_angles(v, f).min() < 3.0
    v2, f2, rep = mvs_amd.enforce_facet_criteria(v, f, 1.0)
    assert rep["collapses"] >= 1 and rep["facets_below_angle"] == 0 and facet_angles(v2, f2).min() >= 20.0
    assert closed_oriented_manifold(f2) and len(v2) == len(np.unique(f)) - rep["collapses"]   # (the two unused corner vertices go too)
    # cap: a vertex pushed onto the diagonal of its square makes a facet with an angle near 180 degrees and no short edge
    jitter = np.zeros((n * n, 2))
    jitter[3 * n + 4] = [-0.48, 0.48]              # vertex (4, 3) moved towards the middle of the square (3..4, 3..4): on its diagonal
    v, f = _flat_grid(n, jitter)
    a = facet_angles(v, f)
    assert a.max() > 170.0
    v2, f2, rep = mvs_amd.enforce_facet_criteria(v, f, 1.0)
    assert rep["flips"] >= 1 and rep["facets_below_angle"] == 0 and facet_angles(v2, f2).min() >= 20.0
    assert closed_oriented_manifold(f2)
    assert abs(mc.signed_volume(v2, f2) - mc.signed_volume(v, f)) < 1e-6 * abs(mc.signed_volume(v, f)) + 1e-6
    ov, of, _ = mo.enforce_facet_criteria(v, f, 20.0, 300.0, 0.375)
    assert np.array_equal(ov, v2) and np.array_equal(of, f2)


def test_an_open_patch_takes_the_second_round_and_still_equals_the_oracle():
    """samples of a spherical cap only: the Poisson surface closes it with a sheet of its own, curvature at the rim is high and surface nets
    have non-manifold edges there -- the facets the first round cannot help get the second one (half of the distance bound), and what is
    left is reported; oracle and library agree on every index either way"""
    pts, nrm = sphere(np.random.default_rng(5), 8000)
    keep = pts[:, 2] > -0.2
    v, f, sp = oracle_surface(pts[keep], nrm[keep])
    v2, f2, rep = mvs_amd.enforce_facet_criteria(v, f, sp)
    ov, of, orep = mo.enforce_facet_criteria(v, f, 20.0, 300.0 * sp, 0.375 * sp)
    assert np.array_equal(ov, v2) and np.array_equal(of, f2)
    assert all(orep[k] == rep[k] for k in ("collapses", "flips", "facets_below_angle", "facets_above_radius"))
    ang = facet_angles(v2, f2).min(1)
    assert rep["facets_below_angle"] == int((ang < 20.0).sum()) <= 0.001 * len(f2) and ang.min() > 10.0
    assert (facet_angles(v, f).min(1) < 20.0).sum() > 100 * max(1, rep["facets_below_angle"])
    use = mc.edge_use(f2)
    assert all(use[(b, a)] == n for (a, b), n in use.items())                  # still closed and oriented (not manifold everywhere: it was not before)


def test_golden_vectors():
    """tests/golden/meshing_small.npz holds the pass's result on the golden Poisson mesh (made by the oracle, tests/golden/make_meshing_golden.py)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "meshing_small.npz"))
    sp = float(g["criteria_spacing"])
    ov, of, orep = mo.enforce_facet_criteria(g["poisson_vertices"], g["poisson_faces"], 20.0, 300.0 * sp, 0.375 * sp)
    assert np.array_equal(ov, g["criteria_vertices"]) and np.array_equal(of, g["criteria_faces"])
    assert [orep["collapses"], orep["flips"], orep["facets_below_angle"]] == g["criteria_ops"].tolist()
    v2, f2, rep = mvs_amd.enforce_facet_criteria(g["poisson_vertices"], g["poisson_faces"], sp)
    assert np.array_equal(v2, g["criteria_vertices"]) and np.array_equal(f2, g["criteria_faces"])
    assert [rep["collapses"], rep["flips"], rep["facets_below_angle"]] == g["criteria_ops"].tolist()
    assert facet_angles(f=f2, v=v2).min() >= 20.0


def test_what_cannot_be_helped_is_left_alone_and_reported():
    # a sliver on an edge shared by FOUR facets (two sheets meeting in a line): neither a collapse nor a flip of that edge is defined
    v = np.array([[0, 0, 0, 1], [0.05, 0, 0, 1], [0, 1, 0, 1], [0, -1, 0, 1], [0, 0, 1, 1], [0, 0, -1, 1]], np.float32)
    f = np.array([[0, 1, 2], [1, 0, 3], [0, 1, 4], [1, 0, 5]], np.int32)
    v2, f2, rep = mvs_amd.enforce_facet_criteria(v, f, 1.0)
    assert np.array_equal(v2, v) and np.array_equal(f2, f)
    assert rep["collapses"] == rep["flips"] == 0 and rep["facets_below_angle"] == 4 and rep["min_angle_deg"] < 3.0
    # a radius bound that does bind is reported, not acted upon
    big = np.array([[0, 0, 0, 1], [10, 0, 0, 1], [0, 10, 0, 1], [0, 0, 10, 1]], np.float32)
    tet = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]], np.int32)
    _, _, rep = mvs_amd.enforce_facet_criteria(big, tet, 1.0, criteria=(20.0, 5.0, 0.375))
    assert rep["facets_above_radius"] == 4 and rep["facets_below_angle"] == 0
    # facets that are not triangles are dropped, unused vertices with them; the empty mesh is fine
    v2, f2, rep = mvs_amd.enforce_facet_criteria(big, np.array([[0, 0, 1], [0, 2, 1]], np.int32), 1.0)
    assert len(f2) == 1 and len(v2) == 3
    v2, f2, rep = mvs_amd.enforce_facet_criteria(np.zeros((0, 4), np.float32), np.zeros((0, 3), np.int32), 1.0)
    assert len(v2) == 0 and len(f2) == 0 and rep["min_angle_deg"] == 180.0


def test_bad_arguments():
    import ctypes as C
    lib = mvs_amd.load_library()
    v = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1]], np.float32)
    s = C.c_void_p()
    vp = C.c_void_p
    bad = np.array([[0, 1, 3]], np.int32)
    assert lib.mvs_surface_from_mesh(v.ctypes.data_as(vp), 3, bad.ctypes.data_as(vp), 1, 1.0, C.byref(s)) != 0 and not s.value   # index out of range
    ok = np.array([[0, 1, 2]], np.int32)
    assert lib.mvs_surface_from_mesh(None, 3, ok.ctypes.data_as(vp), 1, 1.0, C.byref(s)) != 0
    assert lib.mvs_surface_from_mesh(v.ctypes.data_as(vp), 3, ok.ctypes.data_as(vp), 1, 1.0, C.byref(s)) == 0
    try:
        for args in ((-1.0, 1.0, 1.0), (60.0, 1.0, 1.0), (float("nan"), 1.0, 1.0), (20.0, 0.0, 1.0), (20.0, 1.0, -1.0)):
            assert lib.mvs_surface_enforce_criteria(s, *args, None) != 0
        assert lib.mvs_surface_enforce_criteria(None, 20.0, 1.0, 1.0, None) != 0
        assert lib.mvs_surface_enforce_criteria(s, 20.0, 1.0, 1.0, None) == 0     # report is optional
    finally:
        lib.mvs_surface_free(s)


@pytest.mark.parametrize("name", sorted(SURFACES))
def test_simplification_returns_the_facet_count_the_criteria_ask_for(name):
    """cgal_poisson.cpp:95-97's criteria bound a facet's angles from below and its distance from the surface from above -- not its size --
    so the reference's mesher returns as few facets as the curvature allows.  mvs_surface_simplify after the criteria pass: an order of
    magnitude fewer facets, every angle still >= 20 degrees, vertices AND facet centres still within 0.375 spacings of the analytic
    surface, the surface still closed, oriented and a manifold of the same topology and (nearly) the same volume, no vertex moved or added."""
    make, distance = SURFACES[name]
    pts, nrm = make(np.random.default_rng(11))
    v, f, sp = oracle_surface(pts, nrm)
    v1, f1, _ = mvs_amd.enforce_facet_criteria(v, f, sp)
    v2, f2, rep = mvs_amd.simplify_surface(v1, f1, sp)
    assert rep["facets_before"] == len(f1) and rep["facets_after"] == len(f2) and rep["vertices_after"] == len(v2)
    assert len(f2) * 10 < len(f1), (len(f1), len(f2))
    assert facet_angles(v2, f2).min() >= 20.0 - 1e-6
    assert rep["max_accumulated_distance"] <= 0.375 * sp
    if distance is not None:
        P = np.asarray(v2, np.float64)[:, :3]
        assert np.abs(distance(P)).max() <= 0.375 * sp
        assert np.abs(distance(P[f2].mean(1))).max() <= 0.375 * sp            # the chords' sagitta: bounded through the kept normals
    assert closed_oriented_manifold(f2)
    assert len(v2) - len(f2) // 2 == len(v1) - len(f1) // 2                   # Euler characteristic
    assert abs(mc.signed_volume(v2, f2) / mc.signed_volume(v1, f1) - 1.0) < 0.03   # (chords lie inside a convex surface: a tube of radius 0.4 meshed to 0.2 spacings loses 2 %)
    assert rep["collapses"] == len(v1) - len(v2) and len(f1) - len(f2) == 2 * rep["collapses"]
    rows = {r.tobytes(): i for i, r in enumerate(v1)}
    where = np.array([rows[r.tobytes()] for r in v2])
    assert np.all(np.diff(where) > 0)                                        # a subsequence of the input's vertices
    # deterministic; a second application finds little left
    v3, f3, _ = mvs_amd.simplify_surface(v1, f1, sp)
    assert v3.tobytes() == v2.tobytes() and f3.tobytes() == f2.tobytes()


def test_simplification_equals_the_oracle_and_keeps_borders():
    """the oracle's restatement (oracle/meshing_oracle.py: simplify_surface, pure Python) gives the same mesh index for index; on an open
    patch the border vertices all stay"""
    pts, nrm = sphere(np.random.default_rng(3), 600)
    v, f, sp = oracle_surface(pts, nrm)
    v1, f1, _ = mvs_amd.enforce_facet_criteria(v, f, sp)
    v2, f2, rep = mvs_amd.simplify_surface(v1, f1, sp)
    ov, of, _ = mo.simplify_surface(v1, f1, 20.0, 0.375 * sp)
    assert np.array_equal(ov, v2) and np.array_equal(of, f2) and len(f2) * 10 < len(f1)
    # an open patch: the facets of the upper half only
    keep = np.asarray(v1, np.float64)[:, 2][f1].min(1) > 0.0
    fo = f1[keep]
    used = np.unique(fo)
    renum = -np.ones(len(v1), np.int64)
    renum[used] = np.arange(len(used))
    vo, fo = v1[used], renum[fo].astype(np.int32)
    use = mc.edge_use(fo)
    border = set(a for (a, b), n in use.items() if use.get((b, a), 0) == 0) | set(b for (a, b), n in use.items() if use.get((b, a), 0) == 0)
    v3, f3, rep3 = mvs_amd.simplify_surface(vo, fo, sp)
    assert len(f3) * 5 < len(fo)
    kept = set(r.tobytes() for r in v3)
    assert all(vo[i].tobytes() in kept for i in border)
    assert facet_angles(v3, f3).min() >= 20.0 - 1e-6


def test_simplification_argument_errors():
    v, f = _flat_grid(4)
    lib = mvs_amd.load_library()
    import ctypes as C
    s = C.c_void_p()
    assert lib.mvs_surface_from_mesh(v.ctypes.data_as(C.c_void_p), len(v), f.ctypes.data_as(C.c_void_p), len(f), 1.0, C.byref(s)) == 0
    assert lib.mvs_surface_simplify(None, 20.0, 0.1, None) == -1
    assert lib.mvs_surface_simplify(s, 60.0, 0.1, None) == -1
    assert lib.mvs_surface_simplify(s, 20.0, -1.0, None) == -1
    assert lib.mvs_surface_simplify(s, 20.0, 0.0, None) == 0          # no budget: nothing may move
    lib.mvs_surface_free(s)


def test_border_facets_that_cannot_be_repaired_are_trimmed():
    """An OPEN surface (mvs_poisson_surface cuts the level set where the samples' support ends): a facet below the angle bound on the outline
    that no collapse or flip repairs is removed when that only moves the outline -- two or three border edges, or one and an interior opposite
    vertex; the surface stays a manifold with ONE outline, the report says how many went; a sliver on an edge with four facets stays
    (test_what_cannot_be_helped_is_left_alone_and_reported).  C++ == oracle, index for index."""
    n = 8
    rng = np.random.default_rng(3)
    P = [[i + 0.15 * rng.uniform(-1, 1), j + 0.15 * rng.uniform(-1, 1), 0.05 * np.sin(i) * np.cos(j), 1.0] for j in range(n) for i in range(n)]
    F = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            F += [[a, a + 1, a + n + 1], [a, a + n + 1, a + n]]
    for i, t, out in ((1, 0.03, 0.01), (4, 0.97, 0.02), (5, 0.5, 0.004)):      # slivers hung on the bottom border, next to a corner of their edge / flat on it
        pa, pb = np.array(P[i][:3]), np.array(P[i + 1][:3])
        e = pa + t * (pb - pa) + np.array([0.0, -out, 0.0])
        P.append([e[0], e[1], e[2], 1.0])
        F.append([i + 1, i, len(P) - 1])
    v, f = np.array(P, np.float32), np.array(F, np.int32)
    assert (facet_angles(v, f).min(1) < 20.0).sum() == 3
    v2, f2, rep = mvs_amd.enforce_facet_criteria(v, f, 1.0)
    ov, of, orep = mo.enforce_facet_criteria(v, f, 20.0, 300.0, 0.375)
    assert np.array_equal(ov, v2) and np.array_equal(of, f2)
    assert {k: orep[k] for k in ("collapses", "flips", "facets_trimmed", "facets_below_angle")} == {k: rep[k] for k in ("collapses", "flips", "facets_trimmed", "facets_below_angle")}
    assert rep["facets_trimmed"] >= 1 and rep["facets_below_angle"] == 0 and facet_angles(v2, f2).min() >= 20.0
    assert len(f2) == len(f) - rep["facets_trimmed"] - 2 * rep["collapses"]
    use = mc.edge_use(f2)
    border = [(a, b) for (a, b), k in use.items() if (b, a) not in use]
    assert mc.facets_per_edge(f2) == 2 and max(use.values()) == 1
    starts = np.bincount([a for a, _ in border], minlength=len(v2))
    ends = np.bincount([b for _, b in border], minlength=len(v2))
    assert starts.max() == 1 and np.array_equal(starts, ends)                   # the outline passes through each of its vertices once
    nxt = dict(border)
    at, steps = border[0][0], 0
    while steps == 0 or at != border[0][0]:
        at, steps = nxt[at], steps + 1
    assert steps == len(border)                                                  # ... and is one loop
