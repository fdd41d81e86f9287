"""poissonSurface without CGAL / PCL (csrc/poisson.hip behind mvs_poisson_surface; cgal_poisson.cpp:47-136 of the reference) against
oracle/meshing_oracle.py: the box, the splatted integer fields (bit-equal), chi (float32 hipFFT vs float64 numpy: 1e-5 of its range),
the level, and the surface-nets mesh (faces equal, vertices to float32 rounding when both mesh the same field); then what a caller
needs from it: a closed, outward-oriented surface within a cell of the sampled shape."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import meshing_common as mc  # noqa: E402
import meshing_oracle as mo  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    import torch  # noqa: F401  (the HIP runtime, first)
    return ctypes.CDLL(os.path.join(mc.LIBDIR, "libmvs_hip.so"))


def _sphere(rng, n, centre, radius, w=None):
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    xyz = u * radius + np.asarray(centre)
    w = np.ones((n, 1)) if w is None else w
    return np.hstack([xyz * w, w]).astype(np.float32), u.astype(np.float32)


def _torus(rng, n, R, r):
    a, b = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    xyz = np.stack([(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)], 1)
    nrm = np.stack([np.cos(b) * np.cos(a), np.cos(b) * np.sin(a), np.sin(b)], 1)
    return np.hstack([xyz, np.ones((n, 1))]).astype(np.float32), nrm.astype(np.float32)


@pytest.mark.parametrize("lg,n", [(5, 3000), (6, 6000), (7, 20000)])
def test_every_stage_against_the_oracle(hip, lg, n):
    rng = np.random.default_rng(lg)
    pts, nrm = _sphere(rng, n, (0.3, -0.2, 1.0), 0.8, w=rng.uniform(0.5, 2.0, size=(n, 1)))
    nrm = (nrm * rng.uniform(0.6, 1.4, size=(n, 1))).astype(np.float32)          # lengths act as confidences
    nrm[::97] = np.nan                                                            # samples whose normal estimate failed vote for nothing
    r = mc.poisson(hip, pts, nrm, lg, 1.0)
    G, origin, h = mo.poisson_grid(pts, lg)
    assert r["G"] == G and np.array_equal(r["origin"], origin) and np.float32(r["h"]) == h
    splat = mo.poisson_splat(pts, nrm, G, origin, h)
    assert np.array_equal(splat, r["splat"])
    chi = mo.poisson_chi(splat, 1.0)
    assert np.abs(chi - r["chi"]).max() <= 1e-5 * (chi.max() - chi.min())
    xyz = pts[:, :3] / pts[:, 3:4]
    assert abs(mo.poisson_level(chi, G, origin, h, xyz) - r["iso"]) <= 1e-5 * (chi.max() - chi.min())
    assert r["support_nodes"] == min(G, int(np.ceil(8.0 * r["spacing"] / r["h"])))          # MVS_POISSON_SUPPORT_DEFAULT spacings, in nodes
    v, f = mo.surface_nets(r["chi"], r["iso"], origin, h, mo.poisson_support(splat, r["support_nodes"]))
    assert np.array_equal(f, r["faces"]) and len(v) == len(r["vertices"])
    assert np.abs(v - r["vertices"]).max() <= 2e-6 * float(np.abs(origin).max() + G * h)
    v0, f0 = mo.surface_nets(r["chi"], r["iso"], origin, h)                                 # a closed, evenly sampled surface: nothing is trimmed
    assert np.array_equal(f0, f) and np.array_equal(v0, v)
    # meshed from the oracle's own field the surface is the same up to the cells the float32 field decides differently
    v2, f2 = mo.surface_nets(chi.astype(np.float32), np.float32(mo.poisson_level(chi, G, origin, h, xyz)), origin, h)
    assert abs(len(v2) - len(v)) <= 0.002 * len(v) + 2


def test_sphere_and_torus_are_closed_outward_and_within_a_cell(hip):
    rng = np.random.default_rng(42)
    pts, nrm = _sphere(rng, 30000, (1.0, 2.0, -3.0), 1.5)
    r = mc.poisson(hip, pts, nrm, 0, 1.0, keep=False)
    assert r["G"] == 256
    v, f = r["vertices"], r["faces"]
    rad = np.linalg.norm(v[:, :3] - np.array([1.0, 2.0, -3.0]), axis=1)
    assert np.abs(rad - 1.5).max() <= 1.5 * r["h"] and np.all(v[:, 3] == 1.0)
    use = mc.edge_use(f)
    assert all(use[(b, a)] == c for (a, b), c in use.items())
    assert abs(mc.signed_volume(v, f) - 4.0 / 3.0 * np.pi * 1.5 ** 3) <= 0.01 * 4.0 / 3.0 * np.pi * 1.5 ** 3
    # genus 1: Euler characteristic 0 (V - E + F with E = 3F / 2 on a closed triangle mesh)
    pts, nrm = _torus(rng, 40000, 1.0, 0.35)
    r = mc.poisson(hip, pts, nrm, 7, 1.0, keep=False)
    v, f = r["vertices"], r["faces"]
    use = mc.edge_use(f)
    assert all(use[(b, a)] == c for (a, b), c in use.items())
    used = np.unique(f)
    assert len(used) - 3 * len(f) // 2 + len(f) == 0
    assert abs(mc.signed_volume(v, f) - 2 * np.pi ** 2 * 1.0 * 0.35 ** 2) <= 0.02 * 2 * np.pi ** 2 * 0.35 ** 2
    d = np.abs(np.hypot(np.hypot(v[:, 0], v[:, 1]) - 1.0, v[:, 2]) - 0.35)
    assert d.max() <= 1.5 * r["h"]


def test_grid_from_the_average_spacing_keeps_the_references_approximation_bound(hip):
    """cgal_poisson.cpp:77 measures the samples' average 6-nearest-neighbour spacing and :52 / :99 asks the mesher for an approximation
    error of at most 0.375 x that spacing.  The grid is chosen from the same yardstick (device k-NN == scipy's k-d tree), and on analytic
    surfaces every vertex of the output lies within that bound of the true surface."""
    rng = np.random.default_rng(4)
    for name, (pts, nrm), dist in (
            ("sphere", _sphere(rng, 30000, (1.0, 2.0, -3.0), 1.5), lambda v: np.abs(np.linalg.norm(v[:, :3] - np.array([1.0, 2.0, -3.0]), axis=1) - 1.5)),
            ("torus", _torus(rng, 60000, 1.0, 0.35), lambda v: np.abs(np.hypot(np.hypot(v[:, 0], v[:, 1]) - 1.0, v[:, 2]) - 0.35)),
            ("small sphere", _sphere(rng, 4000, (0.0, 0.0, 0.0), 0.5), lambda v: np.abs(np.linalg.norm(v[:, :3], axis=1) - 0.5))):
        r = mc.poisson(hip, pts, nrm, 0, 1.0, keep=False)
        sp = mo.average_spacing(pts)
        assert abs(r["spacing"] - sp) <= 1e-5 * sp, name
        G, origin, h = mo.poisson_grid(pts, 0)
        assert r["G"] == G and np.float32(r["h"]) == h and r["ratio_kept"] == 1 and r["h"] <= 0.75 * sp, name
        assert G == 32 or 1.5 * float(np.ptp(pts[:, :3] / pts[:, 3:4], axis=0).max()) / (G // 2 - 1) > 0.75 * sp, name   # the coarsest grid that keeps the ratio
        err = dist(r["vertices"])
        assert err.max() <= 0.375 * sp, (name, float(err.max()), 0.375 * sp, r["h"])
    # a forced coarse grid reports that the ratio is not kept (and the default caps at 512 nodes per axis)
    pts, nrm = _sphere(rng, 30000, (0, 0, 0), 1.0)
    assert mc.poisson(hip, pts, nrm, 5, 1.0, keep=False)["ratio_kept"] == 0


def test_the_normals_lengths_are_confidences_whatever_their_scale(hip):
    """triangulatePixels hands over normals scaled by a pdf (util.cpp:322-327): lengths of 1e-6 .. 1e-4 on real frames, far below the 2^-16
    quantum of the fixed-point splat.  Only their ratios matter to the level set: the library multiplies them by an exact power of two
    first (median size into [0.5, 1)), so a cloud whose normals are all scaled by 2^-17 gives the SAME bytes as the unit-length one, a scale
    that is not a power of two the same surface, and one wild normal does not push the others below the quantum."""
    rng = np.random.default_rng(123)
    pts, nrm = _sphere(rng, 12000, (0.1, 0.2, -0.3), 0.9)
    conf = rng.uniform(0.3, 1.0, size=(len(nrm), 1)).astype(np.float32)
    nrm = (nrm * conf).astype(np.float32)
    ref = mc.poisson(hip, pts, nrm, 6, 1.0)
    k0 = mo.normal_scale_log2(nrm)
    assert ref["normal_scale_log2"] == k0 and k0 in (0, 1)                                   # the median of the largest components is about 0.5
    small = (nrm * np.float32(2.0 ** -17)).astype(np.float32)
    r = mc.poisson(hip, pts, small, 6, 1.0)
    assert r["normal_scale_log2"] == mo.normal_scale_log2(small) == k0 + 17
    assert np.array_equal(r["splat"], ref["splat"]) and np.array_equal(r["chi"], ref["chi"]) and r["iso"] == ref["iso"]
    assert r["vertices"].tobytes() == ref["vertices"].tobytes() and r["faces"].tobytes() == ref["faces"].tobytes()
    assert np.array_equal(mo.poisson_splat(pts, small, *mo.poisson_grid(pts, 6)), r["splat"])
    # pdf-sized normals, not a power of two: the same surface to within the fixed point's rounding
    tiny = (nrm * np.float32(3.7e-6)).astype(np.float32)
    r2 = mc.poisson(hip, pts, tiny, 6, 1.0)
    assert np.abs(r2["splat"][:3]).max() > 2 ** 12                                           # the field is not quantised away
    assert abs(len(r2["vertices"]) - len(ref["vertices"])) <= 0.005 * len(ref["vertices"]) + 2
    rad = np.linalg.norm(r2["vertices"][:, :3] - np.array([0.1, 0.2, -0.3]), axis=1)
    assert np.abs(rad - 0.9).max() <= 1.5 * r2["h"]
    use = mc.edge_use(r2["faces"])
    assert all(use[(b, a)] == c for (a, b), c in use.items())
    # the scale follows the median, not the maximum: a few samples a thousand times as confident as the rest, one unusable normal (beyond
    # 1e4: it votes for nothing) -- the others are not pushed below the quantum, and the splat still equals the oracle's
    wild = tiny.copy()
    wild[5:25] *= np.float32(1000.0)
    wild[30] = [2.0e4, 0.0, 0.0]
    r3 = mc.poisson(hip, pts, wild, 6, 1.0)
    assert r3["normal_scale_log2"] == mo.normal_scale_log2(wild) == r2["normal_scale_log2"]
    assert np.array_equal(mo.poisson_splat(pts, wild, *mo.poisson_grid(pts, 6)), r3["splat"])
    assert np.abs(r3["splat"][:3]).max() > 2 ** 20 and np.median(np.abs(r3["splat"][:3][r3["splat"][:3] != 0])) > 2 ** 6


def test_an_open_patch_is_meshed_where_the_samples_are_and_nowhere_else(hip):
    """Samples of a spherical cap only (what one side of a scene looks like to recon.cpp:114-121): the solved field is flat away from the
    samples and its level set there is a closing sheet plus fuzz.  mvs_poisson_surface meshes the cells within MVS_POISSON_SUPPORT_DEFAULT
    average spacings of a node that collected sample weight -- equal, face for face, to the oracle's surface nets under the oracle's mask --
    and mvs_poisson_surface_ex(support 0) still returns the closed surface."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(77)
    pts, nrm = _sphere(rng, 30000, (0.0, 0.0, 0.0), 1.0)
    keep = pts[:, 2] / pts[:, 3] > 0.2
    pts, nrm = pts[keep], nrm[keep]
    closed = mc.poisson(hip, pts, nrm, 7, 1.0, support=0.0)
    r = mc.poisson(hip, pts, nrm, 7, 1.0)
    assert closed["support_nodes"] == 0 and r["support_nodes"] == int(np.ceil(8.0 * r["spacing"] / r["h"])) > 0
    assert np.array_equal(closed["chi"], r["chi"]) and closed["iso"] == r["iso"]                # the same field, the same level
    G, origin, h = mo.poisson_grid(pts, 7)
    mask = mo.poisson_support(r["splat"], r["support_nodes"])
    v, f = mo.surface_nets(r["chi"], r["iso"], origin, h, mask)
    assert np.array_equal(f, r["faces"]) and len(v) == len(r["vertices"]) and np.abs(v - r["vertices"]).max() <= 2e-6 * float(np.abs(origin).max() + G * h)
    v0, f0 = mo.surface_nets(closed["chi"], closed["iso"], origin, h)
    assert np.array_equal(f0, closed["faces"])
    # the closed surface has a sheet far from every sample; the trimmed one does not, and it loses nothing near the samples
    tree = cKDTree(pts[:, :3] / pts[:, 3:4])
    reach = (r["support_nodes"] + 2) * np.sqrt(3.0) * r["h"]
    d_closed = tree.query(closed["vertices"][:, :3])[0]
    d_trim = tree.query(r["vertices"][:, :3])[0]
    assert d_trim.max() <= reach < d_closed.max() and (d_closed > reach).sum() > 0.05 * len(d_closed)
    near = set(map(bytes, closed["vertices"][d_closed <= 4.0 * r["spacing"]]))
    assert near <= set(map(bytes, r["vertices"]))
    use = mc.edge_use(closed["faces"])
    assert all(use[(b, a)] == c for (a, b), c in use.items())
    use = mc.edge_use(r["faces"])
    assert any(use[(b, a)] != c for (a, b), c in use.items())                                # open: it has a border
    # the facet criteria pass leaves border edges alone and still reaches the angle bound on what it may touch
    import mvs_amd
    from test_criteria_cpu import facet_angles
    v2, f2, rep = mvs_amd.enforce_facet_criteria(r["vertices"], r["faces"], r["spacing"])
    assert rep["facets_below_angle"] <= 0.002 * len(f2) and (facet_angles(v2, f2).min(1) < 20.0).sum() == rep["facets_below_angle"]


def test_the_default_path_keeps_all_three_facet_criteria_of_the_reference(hip):
    """cgal_poisson.cpp:50-52, 95-97: make_surface_mesh is asked for facet angles >= 20 degrees, facet radii <= 300 average spacings and
    facet distances <= 0.375 average spacings.  mvs_amd.poisson_surface (like poissonSurface of the C++ mirror) runs
    mvs_surface_enforce_criteria with those numbers on the mesh of the HIP solver; recomputed here in numpy, on analytic surfaces."""
    import mvs_amd
    from test_criteria_cpu import circumradii, closed_oriented_manifold, facet_angles
    rng = np.random.default_rng(14)
    for name, (pts, nrm), dist in (
            ("sphere", _sphere(rng, 30000, (1.0, 2.0, -3.0), 1.5), lambda v: np.abs(np.linalg.norm(v[:, :3] - np.array([1.0, 2.0, -3.0]), axis=1) - 1.5)),
            ("torus", _torus(rng, 60000, 1.0, 0.35), lambda v: np.abs(np.hypot(np.hypot(v[:, 0], v[:, 1]) - 1.0, v[:, 2]) - 0.35))):
        raw_v, raw_f = mvs_amd.poisson_surface(pts, nrm, criteria=None, use_precision=True)   # (normals as given: the bytes of mc.poisson)
        r0 = mc.poisson(hip, pts, nrm, 0, 1.0, keep=False)
        assert np.array_equal(raw_v, r0["vertices"]) and np.array_equal(raw_f, r0["faces"]), name      # criteria=None: the mesher's output as it is
        assert facet_angles(raw_v, raw_f).min() < 5.0, name
        rep = {}
        v, f = mvs_amd.poisson_surface(pts, nrm, report=rep, use_precision=True, simplify=False)   # the criteria pass alone first
        sp = rep["average_spacing"]
        assert abs(sp - mo.average_spacing(pts)) <= 1e-5 * sp
        ang = facet_angles(v, f)
        assert ang.min() >= 20.0 - 1e-6 and rep["facets_below_angle"] == 0 and abs(rep["min_angle_deg"] - ang.min()) < 1e-3, (name, float(ang.min()))
        R = circumradii(v, f)
        assert R.max() <= 300.0 * sp and rep["facets_above_radius"] == 0, name
        centres = v[:, :3].astype(np.float64)[f].mean(1)
        assert dist(v.astype(np.float64)).max() <= 0.375 * sp and dist(centres).max() <= 0.375 * sp, name
        assert closed_oriented_manifold(f) and np.all(v[:, 3] == 1.0), name
        assert abs(mc.signed_volume(v, f) / mc.signed_volume(raw_v, raw_f) - 1.0) < 1e-4, name
        assert rep["collapses"] == len(raw_v) - len(v) and 0 < rep["collapses"] < 0.02 * len(raw_v) and rep["flips"] > 0, name
        # the pass is host code over the downloaded mesh: the same bytes as mvs_amd.enforce_facet_criteria on the raw mesh
        v2, f2, _ = mvs_amd.enforce_facet_criteria(raw_v, raw_f, sp)
        assert v2.tobytes() == v.tobytes() and f2.tobytes() == f.tobytes(), name
        # the DEFAULT ends with the simplification pass (round 5): the facet count the criteria ask for, not the grid's -- and all three
        # criteria still hold, recomputed here: angles, radii, distance of vertices and facet centres from the analytic surface
        srep = {}
        vs, fs = mvs_amd.poisson_surface(pts, nrm, report=srep, use_precision=True)
        assert len(fs) * 10 < len(f) and srep["simplify"]["facets_after"] == len(fs) and srep["simplify"]["facets_before"] == len(f), (name, len(f), len(fs))
        assert facet_angles(vs, fs).min() >= 20.0 - 1e-6 and circumradii(vs, fs).max() <= 300.0 * sp, name
        assert dist(vs.astype(np.float64)).max() <= 0.375 * sp and dist(vs[:, :3].astype(np.float64)[fs].mean(1)).max() <= 0.375 * sp, name
        assert closed_oriented_manifold(fs) and abs(mc.signed_volume(vs, fs) / mc.signed_volume(raw_v, raw_f) - 1.0) < 0.03, name
        v3, f3, _ = mvs_amd.simplify_surface(v, f, sp)
        assert v3.tobytes() == vs.tobytes() and f3.tobytes() == fs.tobytes(), name


def test_a_cell_that_collects_more_than_2_pow_31_does_not_wrap(hip):
    """two outliers stretch the box until the real samples share a few cells; with 32-bit accumulators the weight and normal
    fields wrapped silently.  The fields are 64-bit now: the integers equal the oracle's (numpy int64)"""
    rng = np.random.default_rng(11)
    pts, nrm = _sphere(rng, 400000, (0.0, 0.0, 0.0), 0.01)
    pts[0, :3] = (-40.0, -40.0, -40.0)
    pts[1, :3] = (40.0, 40.0, 40.0)
    pts[:, 3] = 1.0
    nrm[5] = (3.0e4, 0.0, 0.0)            # beyond the guard: votes for nothing
    r = mc.poisson(hip, pts, nrm, 5, 1.0)
    G, origin, h = mo.poisson_grid(pts, 5)
    splat = mo.poisson_splat(pts, nrm, G, origin, h)
    assert splat[3].max() > 2 ** 31 and np.array_equal(splat, r["splat"])
    sp = mo.average_spacing(pts)          # the two outliers must neither slow the neighbour search to a crawl nor change its result
    assert abs(r["spacing"] - sp) <= 1e-4 * sp


def test_average_spacing_on_clouds_of_mixed_density_and_coincident_samples(hip):
    """the neighbour search must neither crawl nor go wrong when densities differ by orders of magnitude (a tight cluster inside a sparse
    shell: most samples share a few cells of any uniform grid) or when samples coincide (distance 0 counts, as in CGAL)"""
    import time
    rng = np.random.default_rng(21)
    shell, n1 = _sphere(rng, 20000, (0.0, 0.0, 0.0), 1.0)
    cluster, n2 = _sphere(rng, 60000, (0.3, 0.1, -0.2), 0.002)
    pts, nrm = np.concatenate([shell, cluster]), np.concatenate([n1, n2])
    t0 = time.perf_counter()
    r = mc.poisson(hip, pts, nrm, 6, 1.0, keep=False)
    assert time.perf_counter() - t0 < 30.0
    sp = mo.average_spacing(pts)
    assert sp * (1 - 1e-4) <= r["spacing"] <= sp * 1.05, (r["spacing"], sp)    # exact unless the candidate budget was spent (then an upper bound)
    dup = np.concatenate([shell[:5000]] * 8)                                     # every sample 8 times: the 6 nearest neighbours are copies
    r = mc.poisson(hip, dup, np.concatenate([n1[:5000]] * 8), 6, 1.0, keep=False)
    assert r["spacing"] == 0.0 and mo.average_spacing(dup) == 0.0


def test_a_rerun_gives_the_same_bytes_and_bad_arguments_fail(hip):
    rng = np.random.default_rng(9)
    pts, nrm = _sphere(rng, 5000, (0, 0, 0), 1.0)
    a = mc.poisson(hip, pts, nrm, 6, 1.0)
    b = mc.poisson(hip, pts, nrm, 6, 1.0)
    assert np.array_equal(a["splat"], b["splat"]) and np.array_equal(a["chi"], b["chi"]) and np.array_equal(a["vertices"], b["vertices"]) and np.array_equal(a["faces"], b["faces"])
    s = ctypes.c_void_p()
    p, q = pts.ctypes.data_as(ctypes.c_void_p), nrm.ctypes.data_as(ctypes.c_void_p)
    assert hip.mvs_poisson_surface(p, q, 0, 6, ctypes.c_float(1.0), 0, ctypes.byref(s)) == -1           # MVS_EINVAL: no samples
    assert hip.mvs_poisson_surface(p, q, 5000, 12, ctypes.c_float(1.0), 0, ctypes.byref(s)) == -1       # grid too fine
    assert hip.mvs_poisson_surface(None, q, 5000, 6, ctypes.c_float(1.0), 0, ctypes.byref(s)) == -1
    one = np.tile(pts[:1], (10, 1))
    assert hip.mvs_poisson_surface(one.ctypes.data_as(ctypes.c_void_p), q, 10, 6, ctypes.c_float(1.0), 0, ctypes.byref(s)) == -1   # no extent
    hip.mvs_surface_last_error.restype = ctypes.c_char_p
    assert b"extent" in hip.mvs_surface_last_error()


def test_tessellate_of_the_cpp_mirror_reaches_it(tmp_path):
    exe = os.path.join(ROOT, "mesh-reconstruction_amd", os.environ.get("MVS_BUILD_VARIANT", ""), "bin", "host_selftest")
    r = subprocess.run([exe, "gpu", os.path.join(ROOT, "tests", "data", "tracks"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert "gpu selftest: 0 failures" in r.stdout and "poissonSurface:" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_wrappers_default_is_unit_normals_and_the_references_semantics_are_one_flag_away():
    """both backends of the reference use the normals' lengths as confidences (cgal_poisson.cpp:58-69; pcl.cpp:23 + 198-202).
    mvs_amd.poisson_surface (like host/poisson.cpp) normalises them BY DEFAULT -- a deliberate divergence recorded in DESIGN.md section 9 --
    and use_precision=True is the reference's behaviour.  This pins which is which."""
    import mvs_amd
    rng = np.random.default_rng(11)
    d = rng.normal(size=(6000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.concatenate([d * 0.8, np.ones((len(d), 1))], 1).astype(np.float32)
    lengths = (10.0 ** rng.uniform(-2.0, 0.0, len(d)))[:, None]          # two decades, like triangulatePixels' pdf
    v_unit, f_unit = mvs_amd.poisson_surface(pts, d.astype(np.float32))
    v_def, f_def = mvs_amd.poisson_surface(pts, (d * lengths).astype(np.float32))
    v_conf, f_conf = mvs_amd.poisson_surface(pts, (d * lengths).astype(np.float32), use_precision=True)
    assert v_def.shape == v_unit.shape and np.array_equal(f_def, f_unit) and np.abs(v_def - v_unit).max() < 1e-4   # the lengths did not matter
    assert v_conf.shape != v_unit.shape or not np.array_equal(f_conf, f_unit)                                          # here they did
    v_same, f_same = mvs_amd.poisson_surface(pts, d.astype(np.float32), use_precision=True)                            # unit lengths: no difference
    np.testing.assert_array_equal(f_same, f_unit)


def test_config5_outer_iteration_closes_points_filter_poisson_mesh_render():
    """recon.cpp:114-136 + 42: the point blocks of a few zatisi main frames -> filterPoints -> poissonSurface -> the mesh the next
    iteration renders.  (Synthetic frames: the cloud is whatever the flows give; what is checked is that the stages connect --
    finite vertices, valid indices, a mesh the rasteriser accepts and sees.)"""
    import c5_common
    import mvs_amd
    seq = c5_common.Sequence()
    with mvs_amd.Context(seq.W, seq.H) as ctx:
        ctx.load_mesh(seq.verts, seq.faces)
        blocks = [c5_common.process_main_frame(ctx, seq, f)[2] for f in seq.mains[10:14]]
        cloud = np.concatenate(blocks)
        assert len(cloud) > 2000
        cloud = cloud[::max(1, len(cloud) // 50000)]
        xyz = cloud[:, :3] / cloud[:, 3:4]
        extent = float(np.percentile(xyz, 95, axis=0).max() - np.percentile(xyz, 5, axis=0).min())
        keep = ctx.filter_points(cloud[:, :4], 0.01 * extent)
        pts, nrm = cloud[keep, :4], cloud[keep, 4:7]
        assert 0 < len(pts) <= len(cloud)
        # triangulatePixels scales the normals by its pdf (util.cpp:322-327): lengths far below the splat's 2^-16 quantum and two decades apart
        length = np.linalg.norm(nrm, axis=1)
        assert 0.0 < np.median(length) < 1e-3 and np.percentile(length, 99) > 10.0 * np.percentile(length, 1)
        # the surface-nets mesh as it leaves the GPU: one vertex per patch of a cell, so no edge has more than two facets even on this cloud
        # (round 4, one vertex per cell: 67 such edges, and 203 facets below 20 degrees after the criteria pass that nothing could repair)
        raw_v, raw_f = mvs_amd.poisson_surface(pts, nrm, criteria=None)
        assert mc.facets_per_edge(raw_f) == 2
        rep = {}
        v, f = mvs_amd.poisson_surface(pts, nrm, report=rep)
        assert len(v) > 100 and len(f) > 100 and np.isfinite(v).all() and f.min() >= 0 and f.max() < len(v)
        # ... and the surface is the samples' sheet: not fuzz in empty space (round 4: 16 M vertices before the normals' scale, the support
        # mask and unit normals), most of it within a few spacings of a sample, facets below the angle bound an exception
        # (measured over the SURFACE -- facet centres weighted by facet area -- not over the vertices: since round 5 the mesh is simplified, and the
        # vertices that stay are mostly those on the support's border and on non-manifold edges, which the pass never removes)
        from scipy.spatial import cKDTree
        tri = v[:, :3].astype(np.float64)[f]
        area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        dist = cKDTree(pts[:, :3] / pts[:, 3:4]).query(tri.mean(1))[0] / rep["average_spacing"]
        order = np.argsort(dist)
        median = dist[order][np.searchsorted(np.cumsum(area[order]), 0.5 * area.sum())]
        assert len(v) < 10 * len(pts) and median < 1.0 and area[dist < 3.0].sum() > 0.8 * area.sum() and dist.max() < 2.0 * np.sqrt(3.0) * (rep["support_nodes"] + 2)
        assert rep["facets_below_angle"] < 0.0005 * rep["simplify"]["facets_before"]   # (counted by the criteria pass, before the simplification; measured: 56 of 325 k)
        assert rep["simplify"]["facets_after"] == len(f) and len(f) * 5 < rep["simplify"]["facets_before"]
        ctx.load_mesh(v, f)
        d = ctx.depth(seq.cams[seq.mains[12]])
        assert (d != mvs_amd.BACKGROUND_DEPTH).any()


def test_golden_vectors(hip):
    """the HIP Poisson surface against tests/golden/meshing_small.npz directly (so that kernel and oracle cannot drift together unnoticed)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "meshing_small.npz"))
    r = mc.poisson(hip, g["poisson_points"], g["poisson_normals"], 5, 1.0)
    assert r["G"] == int(g["poisson_G"]) and np.array_equal(r["origin"], g["poisson_origin"]) and np.float32(r["h"]) == g["poisson_h"]
    assert np.array_equal(r["splat"], g["poisson_splat"])
    rng_ = float(np.ptp(g["poisson_chi"]))
    assert np.abs(r["chi"] - g["poisson_chi"]).max() <= 1e-5 * rng_ and abs(r["iso"] - float(g["poisson_level"])) <= 1e-5 * rng_
    # the level set of a field that differs in its last bits: the same surface up to the cells those bits decide
    assert abs(len(r["vertices"]) - len(g["poisson_vertices"])) <= 0.01 * len(g["poisson_vertices"]) + 2
    assert abs(mc.signed_volume(r["vertices"], r["faces"]) - mc.signed_volume(g["poisson_vertices"], g["poisson_faces"])) <= 1e-3 * abs(mc.signed_volume(g["poisson_vertices"], g["poisson_faces"]))


def test_poisson_warmup_builds_the_plans_a_later_call_finds():
    """mvs_poisson_warmup(grid_log2): the ~2 s of rocFFT's run-time kernel compilation for a grid size paid when the caller chooses;
    argument errors are error codes"""
    import time
    import mvs_amd
    lib = mvs_amd.load_library()
    assert lib.mvs_poisson_warmup(4) == mvs_amd.MVS_EINVAL if hasattr(mvs_amd, "MVS_EINVAL") else lib.mvs_poisson_warmup(4) == -1
    assert lib.mvs_poisson_warmup(10) == -1
    assert lib.mvs_poisson_warmup(6) == 0          # 64^3
    t0 = time.perf_counter()
    assert lib.mvs_poisson_warmup(6) == 0          # cached: nothing to build
    assert time.perf_counter() - t0 < 0.05
    rng = np.random.default_rng(5)
    d = rng.normal(size=(3000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.concatenate([d, np.ones((len(d), 1))], 1).astype(np.float32)
    v, f = mvs_amd.poisson_surface(pts, d.astype(np.float32), grid_log2=6)
    assert len(v) > 100 and len(f) > 100
