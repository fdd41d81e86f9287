"""ctypes access to the surface-meshing entry points (host/alpha_shapes.cpp in libmvs_host.so; csrc/poisson.hip in libmvs_hip.so)."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import mvs_amd  # noqa: E402

LIBDIR = mvs_amd.LIB_DIR   # (lib, or san/lib under MVS_BUILD_VARIANT=san)


def host_lib():
    ctypes.CDLL(os.path.join(LIBDIR, "libmvs_hip.so"), mode=ctypes.RTLD_GLOBAL)
    return ctypes.CDLL(os.path.join(LIBDIR, "libmvs_host.so"))


def alpha_shape(lib, pts, forced=0.0):
    pts = np.ascontiguousarray(pts, np.float32)
    n, c = pts.shape
    cnt, al, comp = ctypes.c_int(), ctypes.c_float(), ctypes.c_int()
    p = pts.ctypes.data_as(ctypes.c_void_p)
    assert lib.mvs_alpha_shape_faces(p, n, c, ctypes.c_float(forced), None, 0, ctypes.byref(cnt), ctypes.byref(al), ctypes.byref(comp)) == 0
    faces = np.zeros((cnt.value, 3), np.int32)
    assert lib.mvs_alpha_shape_faces(p, n, c, ctypes.c_float(forced), faces.ctypes.data_as(ctypes.c_void_p), cnt.value, ctypes.byref(cnt), ctypes.byref(al),
                                     ctypes.byref(comp)) == 0
    return faces, al.value, comp.value


def delaunay_cells(lib, pts):
    pts = np.ascontiguousarray(pts, np.float32)
    n, c = pts.shape
    cnt = ctypes.c_int()
    p = pts.ctypes.data_as(ctypes.c_void_p)
    assert lib.mvs_delaunay3_cells(p, n, c, None, 0, ctypes.byref(cnt)) == 0
    out = np.zeros((cnt.value, 4), np.int32)
    assert lib.mvs_delaunay3_cells(p, n, c, out.ctypes.data_as(ctypes.c_void_p), cnt.value, ctypes.byref(cnt)) == 0
    return out


def canonical_faces(f):
    """oriented triangles as a set, each rotated so that its smallest index comes first"""
    f = np.asarray(f).reshape(-1, 3)
    if len(f) == 0:
        return set()
    k = np.argmin(f, 1)
    r = np.arange(len(f))
    return set(map(tuple, np.stack([f[r, (k + i) % 3] for i in range(3)], 1).tolist()))


def signed_volume(vertices, faces):
    v = np.asarray(vertices, np.float64)[:, :3]
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    return float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)


def edge_use(faces):
    """directed edge -> count; a closed oriented surface uses every directed edge as often as its reverse"""
    from collections import Counter
    c = Counter()
    for a, b, d in np.asarray(faces).tolist():
        c[(a, b)] += 1
        c[(b, d)] += 1
        c[(d, a)] += 1
    return c


def facets_per_edge(faces):
    """the largest number of facets that share an (undirected) edge: 2 on a manifold surface, 1 on its border"""
    f = np.asarray(faces, np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    e.sort(1)
    return int(np.unique(e[:, 0] << 32 | e[:, 1], return_counts=True)[1].max())


def poisson(hip, pts, nrm, grid_log2=0, smooth=1.0, keep=True, support=None):
    """support: None = mvs_poisson_surface (the default support radius), a number = mvs_poisson_surface_ex with that many spacings (0: no trimming)"""
    pts = np.ascontiguousarray(pts, np.float32)
    nrm = np.ascontiguousarray(nrm, np.float32)
    s = ctypes.c_void_p()
    hip.mvs_surface_last_error.restype = ctypes.c_char_p
    if support is None:
        rc = hip.mvs_poisson_surface(pts.ctypes.data_as(ctypes.c_void_p), nrm.ctypes.data_as(ctypes.c_void_p), len(pts), grid_log2, ctypes.c_float(smooth), int(keep),
                                     ctypes.byref(s))
    else:
        rc = hip.mvs_poisson_surface_ex(pts.ctypes.data_as(ctypes.c_void_p), nrm.ctypes.data_as(ctypes.c_void_p), len(pts), grid_log2, ctypes.c_float(smooth),
                                        ctypes.c_float(support), int(keep), ctypes.byref(s))
    if rc != 0:
        raise RuntimeError(hip.mvs_surface_last_error().decode())
    nv, nf, G = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    hip.mvs_surface_counts(s, ctypes.byref(nv), ctypes.byref(nf))
    v = np.zeros((nv.value, 4), np.float32)
    f = np.zeros((nf.value, 3), np.int32)
    hip.mvs_surface_fetch(s, v.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p))
    origin = np.zeros(3, np.float32)
    h, iso = ctypes.c_float(), ctypes.c_float()
    hip.mvs_surface_grid(s, ctypes.byref(G), origin.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h), ctypes.byref(iso), None, None)
    avg, node, kept = ctypes.c_float(), ctypes.c_float(), ctypes.c_int()
    assert hip.mvs_surface_spacing(s, ctypes.byref(avg), ctypes.byref(node), ctypes.byref(kept)) == 0 and node.value == h.value
    nodes = ctypes.c_int()
    assert hip.mvs_surface_support(s, ctypes.byref(nodes)) == 0
    nscale = ctypes.c_int()
    assert hip.mvs_surface_normal_scale(s, ctypes.byref(nscale)) == 0
    out = {"vertices": v, "faces": f, "G": G.value, "origin": origin, "h": h.value, "iso": iso.value, "spacing": avg.value, "ratio_kept": kept.value,
           "support_nodes": nodes.value, "normal_scale_log2": nscale.value}
    if keep:
        chi = np.zeros((G.value,) * 3, np.float32)
        splat = np.zeros((4,) + (G.value,) * 3, np.int64)
        assert hip.mvs_surface_grid(s, None, None, None, None, chi.ctypes.data_as(ctypes.c_void_p), splat.ctypes.data_as(ctypes.c_void_p)) == 0
        out["chi"], out["splat"] = chi, splat
    hip.mvs_surface_free(s)
    return out
