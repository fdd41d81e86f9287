"""ctypes binding of libmvs_hip.so (include/mvs.h) for tests/ and bench.py.

Harness only: the product is the C-ABI library; its real host side is the C++ mirror of recon.hpp in
mesh-reconstruction_amd/host/.  This module fails loudly when the HIP library is missing -- there is
no CPU fallback (the oracle lives in oracle/ and is never imported from here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.normpath(os.path.join(_HERE, "..", ".."))
# MVS_BUILD_VARIANT=san: the address / undefined-behaviour sanitized host build (`make sanitize`: san/lib, san/bin) instead of lib / bin
_VARIANT = os.environ.get("MVS_BUILD_VARIANT", "")
LIB_DIR = os.path.join(PKG_ROOT, _VARIANT, "lib") if _VARIANT else os.path.join(PKG_ROOT, "lib")
BIN_DIR = os.path.join(PKG_ROOT, _VARIANT, "bin") if _VARIANT else os.path.join(PKG_ROOT, "bin")
LIB_PATH = os.path.join(LIB_DIR, "libmvs_hip.so")

MVS_SWEEP_VOLUME = 1
MVS_SWEEP_FUSED_ARGMIN = 2
MVS_SWEEP_FORCE_GENERIC = 4
MVS_SWEEP_NO_RECT = 8
MVS_SHARD_ROWS, MVS_SHARD_VIEWS, MVS_SHARD_VIEWS_SCATTER = 0, 1, 2
SHARD_MODES = {"rows": 0, "views": 1, "views_scatter": 2}
MVS_SAMPLER_FIXED, MVS_SAMPLER_EXACT_F32 = 0, 1
SAMPLERS = {"fixed": MVS_SAMPLER_FIXED, "exact": MVS_SAMPLER_EXACT_F32}
MVS_K_SWEEP, MVS_K_ARGMIN, MVS_K_PLAN, MVS_K_RASTER, MVS_K_PROJECT, MVS_K_FLOW = 0, 1, 2, 3, 4, 5
MVS_K_COUNT = 8
BACKGROUND_DEPTH = np.float32(1.0)

# every symbol include/mvs.h declares: (name, restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_fp, _u8p, _i32p, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
ABI = [
    ("mvs_create", _vp, [_i, _i, _i]),
    ("mvs_destroy", None, [_vp]),
    ("mvs_last_error", C.c_char_p, [_vp]),
    ("mvs_set_stream", _i, [_vp, _vp]),
    ("mvs_synchronize", _i, [_vp]),
    ("mvs_host_alloc", _vp, [_sz]),
    ("mvs_host_free", None, [_vp]),
    ("mvs_width", _i, [_vp]),
    ("mvs_height", _i, [_vp]),
    ("mvs_load_mesh", _i, [_vp, _fp, _i, _i32p, _i]),
    ("mvs_depth", _i, [_vp, _fp, _fp]),
    ("mvs_depth_probe", _i, [_vp, _fp, _i, _i32p, _i32p, _fp]),
    ("mvs_projected", _i, [_vp, _fp, _u8p, _fp, _u8p]),
    ("mvs_mix_background", _i, [_vp, _u8p, _u8p, _fp, _u8p]),
    ("mvs_compare", _i, [_vp, _u8p, _u8p, _fp]),
    ("mvs_flow_remap", _i, [_vp, _fp, _i, _u8p, _u8p]),
    ("mvs_flow", _i, [_vp, _u8p, _u8p, _i, _fp]),
    ("mvs_triangulate", _i, [_vp, _i, C.POINTER(_fp), _fp, _fp, _fp, _fp, C.POINTER(_i)]),
    ("mvs_process_frame", _i, [_vp, _fp, _u8p, _i, _fp, C.POINTER(_u8p), _i, _fp, C.POINTER(_i), _fp]),
    ("mvs_process_frame_slots", _i, [_vp, _fp, _i, _i, _fp, _i32p, _i, _fp, C.POINTER(_i), _fp]),
    ("mvs_filter_points", _i, [_vp, _fp, _i, _f, _i32p, C.POINTER(_i)]),
    ("mvs_sweep", _i, [_vp, _fp, _u8p, _i, _fp, C.POINTER(_u8p), _i, _f, _f, _fp, _fp, _fp]),
    ("mvs_warp_by_depth", _i, [_vp, _fp, _fp, _fp, _u8p, _u8p]),
    ("mvs_sweep_set_sampler", _i, [_vp, _i]),
    ("mvs_sweep_sampler", _i, [_vp]),
    ("mvs_sweep_set_main", _i, [_vp, _fp, _u8p]),
    ("mvs_sweep_set_views", _i, [_vp, _i, _fp, C.POINTER(_u8p)]),
    ("mvs_sweep_set_planes", _i, [_vp, _i, _f, _f]),
    ("mvs_sweep_set_main_device", _i, [_vp, _fp, _vp]),
    ("mvs_sweep_set_views_device", _i, [_vp, _i, _fp, C.POINTER(_vp)]),
    ("mvs_sweep_run", _i, [_vp, _i, _i, C.c_uint]),
    ("mvs_sweep_run_planes", _i, [_vp, _i, _i, _i, _i, C.c_uint]),
    ("mvs_sweep_plane_granularity", _i, []),
    ("mvs_sweep_run_rows", _i, [_vp, _i, _i, _i, _i, C.c_uint]),
    ("mvs_sweep_row_granularity", _i, []),
    ("mvs_sweep_row_granularity_of", _i, [_vp]),
    ("mvs_sweep_plan_shape", _i, [_vp]),
    ("mvs_resize_u8", _i, [_vp, _vp, _i, _i, _i, _vp, _i, _i]),
    ("mvs_set_texture_filter", _i, [_vp, _i]),
    ("mvs_texture_filter", _i, [_vp]),
    ("mvs_frame_store", _i, [_vp, _i]),
    ("mvs_frame_upload", _i, [_vp, _i, _vp]),
    ("mvs_frame_upload_device", _i, [_vp, _i, _vp]),
    ("mvs_sweep_handles", _i, [_vp, _i, _fp, _i, _vp, _vp, _i, _f, _f, _vp, _vp]),
    ("mvs_sweep_batch", _i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, C.c_float, C.c_float, _vp, _vp]),
    ("mvs_sweep_batch_async", _i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, C.c_float, C.c_float, _vp, _vp]),
    ("mvs_sweep_batch_wait", _i, [_vp]),
    ("mvs_sweep_argmin", _i, [_vp]),
    ("mvs_sweep_refine_depth", _i, [_vp]),
    ("mvs_sweep_argmin_partial", _i, [_vp, _vp, _i, _i, _vp]),
    ("mvs_sweep_combine_partials", _i, [_vp, _vp, _i]),
    ("mvs_sweep_volume_device", _vp, [_vp, C.POINTER(_sz)]),
    ("mvs_sweep_use_volume", _i, [_vp, _vp, _sz]),
    ("mvs_sweep_depth_device", _vp, [_vp]),
    ("mvs_sweep_cost_device", _vp, [_vp]),
    ("mvs_sweep_index_device", _vp, [_vp]),
    ("mvs_sweep_fetch", _i, [_vp, _fp, _fp, _i32p, _u32p]),
    ("mvs_sweep_view_matrices", _i, [_vp, _fp]),
    ("mvs_comm_create", _vp, [C.POINTER(_i), _i, _i, _i]),
    ("mvs_comm_destroy", None, [_vp]),
    ("mvs_comm_size", _i, [_vp]),
    ("mvs_comm_context", _vp, [_vp, _i]),
    ("mvs_comm_last_error", C.c_char_p, [_vp]),
    ("mvs_comm_set_mode", _i, [_vp, _i]),
    ("mvs_comm_mode", _i, [_vp]),
    ("mvs_comm_set_plane_groups", _i, [_vp, _i]),
    ("mvs_comm_set_planes", _i, [_vp, _i, _f, _f]),
    ("mvs_comm_set_main", _i, [_vp, _fp, _u8p]),
    ("mvs_comm_set_views", _i, [_vp, _i, _fp, C.POINTER(_u8p)]),
    ("mvs_comm_run", _i, [_vp, C.c_uint]),
    ("mvs_comm_fetch", _i, [_vp, _fp, _fp]),
    ("mvs_comm_run_async", _i, [_vp, C.c_uint]),
    ("mvs_comm_wait", _i, [_vp]),
    ("mvs_comm_pending", _i, [_vp]),
    ("mvs_comm_peer_access", _i, [_vp, _i]),
    ("mvs_comm_device", _i, [_vp, _i]),
    ("mvs_sweep_set_plan_cache", _i, [_vp, _i]),
    ("mvs_sweep_sharded", _i, [_vp, _fp, _u8p, _i, _fp, C.POINTER(_u8p), _i, _f, _f, _fp, _fp]),
    ("mvs_profile_enable", _i, [_vp, _i]),
    ("mvs_profile_read", _i, [_vp, _fp, C.POINTER(_i), _i]),
    ("mvs_device_info", C.c_char_p, [_vp]),
    ("mvs_poisson_warmup", _i, [_i]),
    ("mvs_poisson_surface", _i, [_vp, _vp, _i, _i, _f, _i, _vp]),
    ("mvs_poisson_surface_ex", _i, [_vp, _vp, _i, _i, _f, _f, _i, _vp]),
    ("mvs_surface_support", _i, [_vp, _vp]),
    ("mvs_surface_normal_scale", _i, [_vp, _vp]),
    ("mvs_surface_counts", _i, [_vp, _vp, _vp]),
    ("mvs_surface_fetch", _i, [_vp, _vp, _vp]),
    ("mvs_surface_grid", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mvs_surface_spacing", _i, [_vp, _vp, _vp, _vp]),
    ("mvs_surface_enforce_criteria", _i, [_vp, _f, _f, _f, _vp]),
    ("mvs_surface_simplify", _i, [_vp, _f, _f, _vp]),
    ("mvs_surface_from_mesh", _i, [_vp, _i, _vp, _i, _f, _vp]),
    ("mvs_surface_free", None, [_vp]),
    ("mvs_surface_last_error", C.c_char_p, []),
]

_lib = None


class MvsError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libmvs_hip.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("MVS_HIP_LIBRARY") or LIB_PATH   # MVS_HIP_LIBRARY: A/B timing of two builds in one GPU session (tools/)
    if not os.path.exists(path):
        raise MvsError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, res, args in ABI:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # test hook, not part of mvs.h
    if hasattr(lib, "mvs_test_rcp"):
        lib.mvs_test_rcp.restype = _i
        lib.mvs_test_rcp.argtypes = [_vp, C.c_uint, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    if hasattr(lib, "mvs_test_onecall_bands"):
        lib.mvs_test_onecall_bands.restype = _i
        lib.mvs_test_onecall_bands.argtypes = [_vp]
    _lib = lib
    return lib


_host_lib = None


def alpha_shape_faces(points, forced_alpha=0.0):
    """alphaShapeFaces (recon.hpp:33-34) through libmvs_host.so (host code, no GPU): points N x 3 or N x 4 (homogeneous)
    -> (faces F x 3 int32 of row indices, normals out of the solid; the alpha chosen; solid components at that alpha)"""
    global _host_lib
    if _host_lib is None:
        load_library()  # libmvs_host.so links libmvs_hip.so
        path = os.path.join(os.path.dirname(LIB_PATH), "libmvs_host.so")
        if not os.path.exists(path):
            raise MvsError("host library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        _host_lib = C.CDLL(path)
        _host_lib.mvs_alpha_shape_faces.restype = _i
        _host_lib.mvs_alpha_shape_faces.argtypes = [_vp, _i, _i, _f, _vp, _i, _vp, _vp, _vp]
    pts = np.ascontiguousarray(points, np.float32)
    if pts.ndim != 2 or pts.shape[1] not in (3, 4):
        raise ValueError("points must be N x 3 or N x 4")
    count, alpha, comps = C.c_int(), C.c_float(), C.c_int()
    rc = _host_lib.mvs_alpha_shape_faces(pts.ctypes.data_as(_vp), pts.shape[0], pts.shape[1], float(forced_alpha), None, 0, C.byref(count), C.byref(alpha), C.byref(comps))
    if rc != 0:
        raise MvsError("mvs_alpha_shape_faces failed (%d)" % rc)
    faces = np.zeros((count.value, 3), np.int32)
    rc = _host_lib.mvs_alpha_shape_faces(pts.ctypes.data_as(_vp), pts.shape[0], pts.shape[1], float(forced_alpha), faces.ctypes.data_as(_vp), count.value, C.byref(count),
                                         C.byref(alpha), C.byref(comps))
    if rc != 0:
        raise MvsError("mvs_alpha_shape_faces failed (%d)" % rc)
    return faces, alpha.value, comps.value


class CriteriaReport(C.Structure):
    """mvs_criteria_report (include/mvs.h)"""
    _fields_ = [("collapses", C.c_int), ("flips", C.c_int), ("facets_below_angle", C.c_int), ("facets_above_radius", C.c_int),
                ("min_angle_deg", C.c_float), ("max_circumradius", C.c_float), ("facets_trimmed", C.c_int)]


# cgal_poisson.cpp:50-52: sm_angle (degrees), sm_radius and sm_distance (in average spacings)
REFERENCE_FACET_CRITERIA = (20.0, 300.0, 0.375)


POISSON_SUPPORT_DEFAULT = 8.0   # MVS_POISSON_SUPPORT_DEFAULT (include/mvs.h)


def poisson_surface(points, normals, grid_log2=0, smooth_cells=1.0, criteria=REFERENCE_FACET_CRITERIA, report=None, support_spacings=POISSON_SUPPORT_DEFAULT,
                    use_precision=False, simplify=True):
    """poissonSurface (recon.hpp:37) through mvs_poisson_surface + mvs_surface_enforce_criteria: points N x 4 homogeneous, normals N x 3
    (out of the solid) -> (vertices V x 4 float32 with w = 1, faces F x 3 int32).  criteria = (min angle in degrees, max facet radius and
    max facet distance in units of the samples' average spacing), the reference's by default; None: the surface-nets mesh as it is.
    support_spacings: the level set is meshed only within that many average spacings of the samples (0: everywhere; include/mvs.h).
    use_precision: keep the normals' lengths as confidences -- what BOTH backends of the reference do (cgal_poisson.cpp:58-69 hands them to
    CGAL as they are; pcl.cpp:23 defines USE_PRECISION, so pcl.cpp:198-202 sets setConfidence(true)).  The default, False, normalises them to
    unit length first: a deliberate divergence from the reference, measured on the pipeline's own clouds (host/poisson.cpp).
    simplify (with criteria): after the criteria pass, remove the vertices the criteria do not need (mvs_surface_simplify: edge collapses under
    the same angle bound and an accumulated distance of at most criteria[2] spacings) -- the reference's mesher returns as few facets as the
    curvature allows, the grid mesher the grid's density; False: the criteria pass's mesh as it is.
    report: a dict that receives the fields of mvs_criteria_report, the average spacing, the support radius in nodes and, under "simplify",
    the fields of mvs_simplify_report"""
    lib = load_library()
    pts = np.ascontiguousarray(points, np.float32)
    nrm = np.ascontiguousarray(normals, np.float32)
    if pts.ndim != 2 or pts.shape[1] != 4 or nrm.shape != (len(pts), 3):
        raise ValueError("points must be N x 4 and normals N x 3")
    if not use_precision:
        with np.errstate(invalid="ignore", over="ignore"):
            length = np.sqrt((nrm.astype(np.float64) ** 2).sum(1))
            ok = (length > 0.0) & (length < 1e30)
        nrm = np.where(ok[:, None], nrm.astype(np.float64) / np.where(ok, length, 1.0)[:, None], nrm).astype(np.float32)
    s = C.c_void_p()
    if lib.mvs_poisson_surface_ex(pts.ctypes.data_as(_vp), nrm.ctypes.data_as(_vp), len(pts), int(grid_log2), float(smooth_cells), float(support_spacings), 0,
                                  C.byref(s)) != 0:
        raise MvsError(lib.mvs_surface_last_error().decode())
    try:
        spacing = C.c_float()
        lib.mvs_surface_spacing(s, C.byref(spacing), None, None)
        if criteria is not None and spacing.value > 0.0:
            rep = CriteriaReport()
            rc = lib.mvs_surface_enforce_criteria(s, float(criteria[0]), float(criteria[1]) * spacing.value, float(criteria[2]) * spacing.value, C.byref(rep))
            if rc != 0:
                raise MvsError("mvs_surface_enforce_criteria failed (%d)" % rc)
            if report is not None:
                report.update({name: getattr(rep, name) for name, _ in CriteriaReport._fields_})
            if simplify:
                srep = SimplifyReport()
                rc = lib.mvs_surface_simplify(s, float(criteria[0]), float(criteria[2]) * spacing.value, C.byref(srep))
                if rc != 0:
                    raise MvsError("mvs_surface_simplify failed (%d)" % rc)
                if report is not None:
                    report["simplify"] = {name: getattr(srep, name) for name, _ in SimplifyReport._fields_}
        if report is not None:
            nodes = C.c_int()
            lib.mvs_surface_support(s, C.byref(nodes))
            report["average_spacing"] = spacing.value
            report["support_nodes"] = nodes.value
        nv, nf = C.c_int(), C.c_int()
        lib.mvs_surface_counts(s, C.byref(nv), C.byref(nf))
        v = np.zeros((nv.value, 4), np.float32)
        f = np.zeros((nf.value, 3), np.int32)
        lib.mvs_surface_fetch(s, v.ctypes.data_as(_vp), f.ctypes.data_as(_vp))
    finally:
        lib.mvs_surface_free(s)
    return v, f


class SimplifyReport(C.Structure):
    _fields_ = [("collapses", C.c_int), ("vertices_before", C.c_int), ("vertices_after", C.c_int), ("facets_before", C.c_int), ("facets_after", C.c_int),
                ("max_accumulated_distance", C.c_float)]


def simplify_surface(vertices, faces, average_spacing, criteria=REFERENCE_FACET_CRITERIA):
    """mvs_surface_from_mesh + mvs_surface_simplify on a caller's mesh (host code, no GPU): -> (vertices, faces, report dict)"""
    lib = load_library()
    v = np.ascontiguousarray(vertices, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    s = C.c_void_p()
    if lib.mvs_surface_from_mesh(v.ctypes.data_as(_vp), len(v), f.ctypes.data_as(_vp), len(f), float(average_spacing), C.byref(s)) != 0:
        raise MvsError("mvs_surface_from_mesh failed")
    try:
        rep = SimplifyReport()
        rc = lib.mvs_surface_simplify(s, float(criteria[0]), float(criteria[2]) * average_spacing, C.byref(rep))
        if rc != 0:
            raise MvsError("mvs_surface_simplify failed (%d)" % rc)
        nv, nf = C.c_int(), C.c_int()
        lib.mvs_surface_counts(s, C.byref(nv), C.byref(nf))
        vo = np.zeros((nv.value, 4), np.float32)
        fo = np.zeros((nf.value, 3), np.int32)
        lib.mvs_surface_fetch(s, vo.ctypes.data_as(_vp), fo.ctypes.data_as(_vp))
        return vo, fo, {name: getattr(rep, name) for name, _ in SimplifyReport._fields_}
    finally:
        lib.mvs_surface_free(s)


def enforce_facet_criteria(vertices, faces, average_spacing, criteria=REFERENCE_FACET_CRITERIA):
    """mvs_surface_from_mesh + mvs_surface_enforce_criteria on a caller's mesh (host code, no GPU): -> (vertices, faces, report dict)"""
    lib = load_library()
    v = np.ascontiguousarray(vertices, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    if v.ndim != 2 or v.shape[1] != 4 or f.ndim != 2 or f.shape[1] != 3:
        raise ValueError("vertices must be V x 4 and faces F x 3")
    s = C.c_void_p()
    if lib.mvs_surface_from_mesh(v.ctypes.data_as(_vp), len(v), f.ctypes.data_as(_vp), len(f), float(average_spacing), C.byref(s)) != 0:
        raise MvsError("mvs_surface_from_mesh: bad argument")
    try:
        rep = CriteriaReport()
        rc = lib.mvs_surface_enforce_criteria(s, float(criteria[0]), float(criteria[1]) * average_spacing, float(criteria[2]) * average_spacing, C.byref(rep))
        if rc != 0:
            raise MvsError("mvs_surface_enforce_criteria failed (%d)" % rc)
        nv, nf = C.c_int(), C.c_int()
        lib.mvs_surface_counts(s, C.byref(nv), C.byref(nf))
        v2 = np.zeros((nv.value, 4), np.float32)
        f2 = np.zeros((nf.value, 3), np.int32)
        lib.mvs_surface_fetch(s, v2.ctypes.data_as(_vp), f2.ctypes.data_as(_vp))
    finally:
        lib.mvs_surface_free(s)
    return v2, f2, {name: getattr(rep, name) for name, _ in CriteriaReport._fields_}


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _u8(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _ptr(a, t):
    return a.ctypes.data_as(t)


class _DeviceArray:
    """minimal __cuda_array_interface__ carrier so torch can alias a library-owned device buffer without a copy"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                         "strides": None}


def pinned_array(shape, dtype):
    """numpy array on page-locked host memory from mvs_host_alloc (freed when the array and its views are gone)"""
    lib = load_library()
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    ptr = lib.mvs_host_alloc(max(n, 1))
    if not ptr:
        raise MvsError("mvs_host_alloc(%d) failed" % n)
    buf = (C.c_char * max(n, 1)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    import weakref
    weakref.finalize(buf, lib.mvs_host_free, ptr)
    return arr


class Comm:
    """mvs_comm: one main view swept on several GPUs of one node (row bands by default; the views split with RCCL on request)."""

    def __init__(self, devices, width, height, sampler=None):
        self.lib = load_library()
        self.W, self.H = int(width), int(height)
        devs = (C.c_int * max(len(devices), 1))(*devices)
        self.h = self.lib.mvs_comm_create(devs if len(devices) else None, len(devices), self.W, self.H)
        if not self.h:
            raise MvsError("mvs_comm_create failed: %s" % self.lib.mvs_comm_last_error(None).decode())
        if sampler is not None:
            for r in range(self.size()):
                rc = self.lib.mvs_sweep_set_sampler(self.lib.mvs_comm_context(self.h, r), SAMPLERS[sampler])
                if rc:
                    raise MvsError("mvs_sweep_set_sampler failed on rank %d" % r)

    def size(self):
        return self.lib.mvs_comm_size(self.h)

    def set_mode(self, mode, plane_groups=None):
        """"rows" (default: row bands, no collective), "views" (plane-group all-reduce pipeline) or "views_scatter" (reduce-scatter)"""
        rc = self.lib.mvs_comm_set_mode(self.h, SHARD_MODES[mode] if isinstance(mode, str) else int(mode))
        if rc == 0 and plane_groups is not None:
            rc = self.lib.mvs_comm_set_plane_groups(self.h, int(plane_groups))
        if rc:
            raise MvsError("libmvs_hip error %d: %s" % (rc, self.lib.mvs_comm_last_error(self.h).decode()))

    def mode(self):
        return self.lib.mvs_comm_mode(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mvs_comm_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sweep(self, main_cam, main_img, side_cams, side_imgs, nplanes, z_lo=-1.0, z_hi=1.0):
        W, H = self.W, self.H
        V = len(side_imgs)
        cam = _f32(main_cam, (4, 4))
        img = _u8(main_img, (H, W))
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((1, 4, 4), np.float32)
        frames = [_u8(s, (H, W)) for s in side_imgs]
        arr = (_u8p * max(V, 1))(*[_ptr(f, _u8p) for f in frames])
        depth = np.empty((H, W), np.float32)
        cost = np.empty((H, W), np.float32)
        rc = self.lib.mvs_sweep_sharded(self.h, _ptr(cam, _fp), _ptr(img, _u8p), V, _ptr(cams, _fp), arr, int(nplanes), float(z_lo), float(z_hi),
                                        _ptr(depth, _fp), _ptr(cost, _fp))
        if rc:
            raise MvsError("libmvs_hip error %d: %s" % (rc, self.lib.mvs_comm_last_error(self.h).decode()))
        return depth, cost

    def _check(self, rc):
        if rc:
            raise MvsError("libmvs_hip error %d: %s" % (rc, self.lib.mvs_comm_last_error(self.h).decode()))

    def context(self, rank):
        """the borrowed mvs_ctx handle of a rank (for the C entry points that take one)"""
        return self.lib.mvs_comm_context(self.h, int(rank))

    def set(self, main_cam, main_img, side_cams, side_imgs, nplanes, z_lo=-1.0, z_hi=1.0):
        """the resident form: planes, main view and ALL side views uploaded to every rank once (mvs_comm_set_planes / _set_main / _set_views)"""
        W, H = self.W, self.H
        V = len(side_imgs)
        cam = _f32(main_cam, (4, 4))
        img = _u8(main_img, (H, W))
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((1, 4, 4), np.float32)
        frames = [_u8(s, (H, W)) for s in side_imgs]
        arr = (_u8p * max(V, 1))(*[_ptr(f, _u8p) for f in frames])
        self._check(self.lib.mvs_comm_set_planes(self.h, int(nplanes), float(z_lo), float(z_hi)))
        self._check(self.lib.mvs_comm_set_main(self.h, _ptr(cam, _fp), _ptr(img, _u8p)))
        self._check(self.lib.mvs_comm_set_views(self.h, V, _ptr(cams, _fp), arr))

    def run(self, flags=0):
        """one sweep of what is resident in the current mode; the maps stay on rank 0's GPU (mvs_comm_run)"""
        self._check(self.lib.mvs_comm_run(self.h, int(flags)))

    def run_async(self):
        """queue one sweep of what is resident (mvs_comm_run_async): up to two in flight in rows mode; wait() publishes the oldest"""
        self._check(self.lib.mvs_comm_run_async(self.h, 0))

    def wait(self):
        self._check(self.lib.mvs_comm_wait(self.h))

    def pending(self):
        return self.lib.mvs_comm_pending(self.h)

    def peer_access(self):
        """per rank: 1 = its band copies travel GPU to GPU to rank 0, 0 = staged through host memory (mvs_comm_peer_access)"""
        return [self.lib.mvs_comm_peer_access(self.h, r) for r in range(self.size())]

    def devices(self):
        return [self.lib.mvs_comm_device(self.h, r) for r in range(self.size())]

    def note(self):
        """mvs_comm_last_error: after creation, "no error" or the note naming ranks without peer access"""
        return self.lib.mvs_comm_last_error(self.h).decode()

    def fetch(self, want_cost=True):
        depth = np.empty((self.H, self.W), np.float32)
        cost = np.empty((self.H, self.W), np.float32) if want_cost else None
        self._check(self.lib.mvs_comm_fetch(self.h, _ptr(depth, _fp), _ptr(cost, _fp) if want_cost else None))
        return (depth, cost) if want_cost else depth


class Context:
    """One GPU context (mvs_ctx).  Mirrors the life cycle of the reference's RenderGLX (render_glx.cpp:152-227)."""

    def __init__(self, width, height, device=0, sampler=None):
        self.lib = load_library()
        self.W, self.H = int(width), int(height)
        self.h = self.lib.mvs_create(int(device), self.W, self.H)
        if not self.h:
            raise MvsError("mvs_create failed: %s" % self.lib.mvs_last_error(None).decode())
        if sampler is not None:
            self.set_sampler(sampler)

    def set_sampler(self, sampler):
        """"fixed" (default of the library: 1/32-texel positions, 8-bit weight table, cells count<<24 | sum) or "exact"
        (f32 bilinear rounded to u8, cells count<<16 | sum) -- include/mvs.h"""
        self._check(self.lib.mvs_sweep_set_sampler(self.h, SAMPLERS[sampler] if isinstance(sampler, str) else int(sampler)))

    def sampler(self):
        return {v: k for k, v in SAMPLERS.items()}[self.lib.mvs_sweep_sampler(self.h)]

    def close(self):
        if getattr(self, "h", None):
            self.lib.mvs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MvsError("libmvs_hip error %d: %s" % (rc, self.lib.mvs_last_error(self.h).decode()))

    def info(self):
        return self.lib.mvs_device_info(self.h).decode()

    def synchronize(self):
        self._check(self.lib.mvs_synchronize(self.h))

    def set_stream(self, stream_handle):
        self._check(self.lib.mvs_set_stream(self.h, C.c_void_p(stream_handle)))

    # ---- sweep ----------------------------------------------------------------------------------
    def sweep(self, main_cam, main_img, side_cams, side_imgs, nplanes, z_lo=-1.0, z_hi=1.0, want_cost=False,
              want_volume=False):
        """mvs_sweep on host buffers -> depth (H,W) [, cost (H,W)] [, volume (D,H,W)]"""
        W, H = self.W, self.H
        V = len(side_imgs)
        cam = _f32(main_cam, (4, 4))
        img = _u8(main_img, (H, W))
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((0, 4, 4), np.float32)
        frames = [_u8(s, (H, W)) for s in side_imgs]
        arr = (_u8p * max(V, 1))(*[_ptr(f, _u8p) for f in frames])
        depth = np.empty((H, W), np.float32)
        cost = np.empty((H, W), np.float32) if want_cost else None
        vol = np.empty((nplanes, H, W), np.float32) if want_volume else None
        self.D = int(nplanes)   # the views and planes stay resident: sweep_run / sweep_fetch may follow
        self._check(self.lib.mvs_sweep(self.h, _ptr(cam, _fp), _ptr(img, _u8p), V, _ptr(cams, _fp), arr, int(nplanes),
                                       float(z_lo), float(z_hi), _ptr(depth, _fp),
                                       _ptr(cost, _fp) if want_cost else None, _ptr(vol, _fp) if want_volume else None))
        out = [depth]
        if want_cost:
            out.append(cost)
        if want_volume:
            out.append(vol)
        return out[0] if len(out) == 1 else tuple(out)

    def warp_by_depth(self, main_cam, depth, side_cam, frame):
        cam, sc = _f32(main_cam, (4, 4)), _f32(side_cam, (4, 4))
        d, f = _f32(depth, (self.H, self.W)), _u8(frame, (self.H, self.W))
        out = np.empty((self.H, self.W, 2), np.uint8)
        self._check(self.lib.mvs_warp_by_depth(self.h, _ptr(cam, _fp), _ptr(d, _fp), _ptr(sc, _fp), _ptr(f, _u8p), _ptr(out, _u8p)))
        return out

    def sweep_set(self, main_cam, main_img, side_cams, side_imgs, nplanes, z_lo=-1.0, z_hi=1.0):
        W, H = self.W, self.H
        V = len(side_imgs)
        cam = _f32(main_cam, (4, 4))
        img = _u8(main_img, (H, W))
        self._check(self.lib.mvs_sweep_set_main(self.h, _ptr(cam, _fp), _ptr(img, _u8p)))
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((0, 4, 4), np.float32)
        frames = [_u8(s, (H, W)) for s in side_imgs]
        arr = (_u8p * max(V, 1))(*[_ptr(f, _u8p) for f in frames])
        self._check(self.lib.mvs_sweep_set_views(self.h, V, _ptr(cams, _fp), arr))
        self._check(self.lib.mvs_sweep_set_planes(self.h, int(nplanes), float(z_lo), float(z_hi)))
        self.V, self.D = V, int(nplanes)

    def sweep_set_views(self, side_cams, side_imgs):
        """mvs_sweep_set_views alone (new side views for the main view already set)"""
        W, H = self.W, self.H
        V = len(side_imgs)
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((0, 4, 4), np.float32)
        frames = [_u8(s, (H, W)) for s in side_imgs]
        arr = (_u8p * max(V, 1))(*[_ptr(f, _u8p) for f in frames])
        self._check(self.lib.mvs_sweep_set_views(self.h, V, _ptr(cams, _fp), arr))
        self.V = V

    def sweep_set_planes(self, nplanes, z_lo=-1.0, z_hi=1.0):
        self._check(self.lib.mvs_sweep_set_planes(self.h, int(nplanes), float(z_lo), float(z_hi)))
        self.D = int(nplanes)

    def sweep_set_main_device(self, main_cam, main_ptr):
        """mvs_sweep_set_main_device: main_ptr = address of H*W u8 in the memory of the context's GPU (e.g. tensor.data_ptr()); stream-ordered"""
        cam = _f32(main_cam, (4, 4))
        self._check(self.lib.mvs_sweep_set_main_device(self.h, _ptr(cam, _fp), C.c_void_p(int(main_ptr))))

    def sweep_set_views_device(self, side_cams, side_ptrs):
        """mvs_sweep_set_views_device: side_ptrs = device addresses of the side frames (H*W u8 each); stream-ordered, no synchronisation"""
        V = len(side_ptrs)
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((0, 4, 4), np.float32)
        arr = (_vp * max(V, 1))(*[C.c_void_p(int(q)) for q in side_ptrs])
        self._check(self.lib.mvs_sweep_set_views_device(self.h, V, _ptr(cams, _fp), arr))
        self.V = V

    def sweep_run(self, view_first=0, view_count=None, flags=MVS_SWEEP_VOLUME):
        if view_count is None:
            view_count = self.V - view_first
        self._check(self.lib.mvs_sweep_run(self.h, int(view_first), int(view_count), int(flags)))

    def sweep_run_planes(self, view_first, view_count, plane_first, plane_count, flags=MVS_SWEEP_VOLUME):
        self._check(self.lib.mvs_sweep_run_planes(self.h, int(view_first), int(view_count), int(plane_first), int(plane_count),
                                                  int(flags)))

    def plane_granularity(self):
        return self.lib.mvs_sweep_plane_granularity()

    def sweep_run_rows(self, row_first, row_count, view_first=0, view_count=None, flags=MVS_SWEEP_VOLUME):
        if view_count is None:
            view_count = self.V - view_first
        self._check(self.lib.mvs_sweep_run_rows(self.h, int(view_first), int(view_count), int(row_first), int(row_count), int(flags)))

    def row_granularity(self):
        """row-band boundaries must be multiples of this (depends on the sampler set on the context)"""
        return self.lib.mvs_sweep_row_granularity_of(self.h)

    def set_plan_cache(self, enable):
        self._check(self.lib.mvs_sweep_set_plan_cache(self.h, 1 if enable else 0))

    def plan_shape(self):
        return self.lib.mvs_sweep_plan_shape(self.h)

    def sweep_refine_depth(self):
        """sub-plane parabola refinement of the depth map (needs the volume and a depth selection of the same run)"""
        self._check(self.lib.mvs_sweep_refine_depth(self.h))

    # ---- frame store + batched sweep (a sequence on one GPU) ------------------------------------
    def frame_store(self, capacity):
        self._check(self.lib.mvs_frame_store(self.h, int(capacity)))
        self._store_keep = {}

    def frame_upload(self, slot, frame):
        f = _u8(frame, (self.H, self.W))
        self._store_keep[int(slot)] = f   # the copy is asynchronous: keep the array alive until the next synchronising call
        self._check(self.lib.mvs_frame_upload(self.h, int(slot), f.ctypes.data_as(C.c_void_p)))

    def frame_upload_device(self, slot, frame_ptr):
        """mvs_frame_upload_device: frame_ptr = device address of H*W u8 (stream-ordered)"""
        self._check(self.lib.mvs_frame_upload_device(self.h, int(slot), C.c_void_p(int(frame_ptr))))

    def sweep_handles(self, main_slot, main_cam, side_slots, side_cams, nplanes, z_lo=-1.0, z_hi=1.0, want_cost=False, out=None):
        """mvs_sweep_handles: one main view whose frames are slots of the frame store -> depth [H,W] (, cost)"""
        ss = np.ascontiguousarray(side_slots, dtype=np.int32).reshape(-1)
        S = len(ss)
        mc = _f32(main_cam, (4, 4))
        sc = _f32(np.asarray(side_cams, dtype=np.float32).reshape(S, 4, 4)) if S else np.zeros((1, 4, 4), np.float32)
        depth = out if out is not None else np.empty((self.H, self.W), np.float32)
        cost = np.empty((self.H, self.W), np.float32) if want_cost else None
        self._check(self.lib.mvs_sweep_handles(self.h, int(main_slot), _ptr(mc, _fp), S, ss.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), int(nplanes),
                                               float(z_lo), float(z_hi), depth.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p) if want_cost else None))
        self._store_keep = {}
        self.V, self.D = S, int(nplanes)
        return (depth, cost) if want_cost else depth

    def sweep_batch_async(self, main_slots, main_cams, side_slots, side_cams, nplanes, out, z_lo=-1.0, z_hi=1.0, cost_out=None):
        """mvs_sweep_batch_async: queues the batch; `out` [M,H,W] f32 (page-locked: pinned_array) is complete after sweep_batch_wait()"""
        ms = np.ascontiguousarray(main_slots, dtype=np.int32)
        ss = np.ascontiguousarray(side_slots, dtype=np.int32)
        M, S = ss.shape
        mc = _f32(np.asarray(main_cams, dtype=np.float32).reshape(M, 4, 4))
        sc = _f32(np.asarray(side_cams, dtype=np.float32).reshape(M, S, 4, 4))
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size >= M * self.H * self.W
        assert cost_out is None or (cost_out.dtype == np.float32 and cost_out.flags.c_contiguous and cost_out.size >= M * self.H * self.W)
        # the copy stream writes into these arrays after this call has returned: they stay referenced until the wait (a caller that
        # drops its page-locked array -- mvs_host_free through the finalizer -- would hand the DMA freed memory: ADVICE r04)
        if not hasattr(self, "_batch_keep"):
            self._batch_keep = []
        self._batch_keep.append((out, cost_out))
        # the library keeps two batches in flight and waits for the oldest itself when a third is queued (batch_slot[2]): the arrays of
        # anything older are complete and need no reference from here (a caller that never calls the wait would otherwise pin them all)
        del self._batch_keep[:-2]
        self._check(self.lib.mvs_sweep_batch_async(self.h, M, ms.ctypes.data_as(C.c_void_p), mc.ctypes.data_as(C.c_void_p), S, ss.ctypes.data_as(C.c_void_p),
                                                   sc.ctypes.data_as(C.c_void_p), int(nplanes), float(z_lo), float(z_hi), out.ctypes.data_as(C.c_void_p),
                                                   cost_out.ctypes.data_as(C.c_void_p) if cost_out is not None else None))

    def sweep_batch_wait(self):
        try:
            self._check(self.lib.mvs_sweep_batch_wait(self.h))
        finally:
            self._batch_keep = []

    def sweep_batch(self, main_slots, main_cams, side_slots, side_cams, nplanes, z_lo=-1.0, z_hi=1.0, want_cost=False, out=None):
        """mvs_sweep_batch: main_slots [M], main_cams [M,4,4], side_slots [M,S], side_cams [M,S,4,4] -> depth [M,H,W] (, cost [M,H,W])"""
        ms = np.ascontiguousarray(main_slots, dtype=np.int32)
        ss = np.ascontiguousarray(side_slots, dtype=np.int32)
        M, S = ss.shape
        mc = _f32(np.asarray(main_cams, dtype=np.float32).reshape(M, 4, 4))
        sc = _f32(np.asarray(side_cams, dtype=np.float32).reshape(M, S, 4, 4))
        depth = out if out is not None else np.empty((M, self.H, self.W), np.float32)
        cost = np.empty((M, self.H, self.W), np.float32) if want_cost else None
        self._check(self.lib.mvs_sweep_batch(self.h, M, ms.ctypes.data_as(C.c_void_p), mc.ctypes.data_as(C.c_void_p), S, ss.ctypes.data_as(C.c_void_p),
                                             sc.ctypes.data_as(C.c_void_p), int(nplanes), float(z_lo), float(z_hi), depth.ctypes.data_as(C.c_void_p),
                                             cost.ctypes.data_as(C.c_void_p) if want_cost else None))
        self._store_keep = {}
        return (depth, cost) if want_cost else depth

    def depth_device_array(self):
        """zero-copy [H, W] f32 view of the device depth map for torch.as_tensor(..., device='cuda') (valid until the
        context is closed; the buffer exists after the first sweep run)"""
        ptr = self.lib.mvs_sweep_depth_device(self.h)
        if not ptr:
            raise MvsError("no depth map on the device yet")
        return _DeviceArray(ptr, (self.H, self.W), "<f4")

    def sweep_argmin(self):
        self._check(self.lib.mvs_sweep_argmin(self.h))

    def sweep_argmin_partial(self, volume_slice_ptr, plane_first, plane_count, partial_out_ptr):
        self._check(self.lib.mvs_sweep_argmin_partial(self.h, C.c_void_p(volume_slice_ptr), int(plane_first), int(plane_count),
                                                      C.c_void_p(partial_out_ptr)))

    def sweep_combine_partials(self, partials_ptr, nparts):
        self._check(self.lib.mvs_sweep_combine_partials(self.h, C.c_void_p(partials_ptr), int(nparts)))

    def sweep_volume_device(self):
        n = _sz(0)
        p = self.lib.mvs_sweep_volume_device(self.h, C.byref(n))
        if not p:
            raise MvsError("no volume: %s" % self.lib.mvs_last_error(self.h).decode())
        return p, n.value

    def sweep_use_volume(self, device_ptr, nbytes):
        self._check(self.lib.mvs_sweep_use_volume(self.h, C.c_void_p(device_ptr), nbytes))

    def sweep_result_pointers(self):
        return (self.lib.mvs_sweep_depth_device(self.h), self.lib.mvs_sweep_cost_device(self.h),
                self.lib.mvs_sweep_index_device(self.h))

    def sweep_fetch(self, want_volume=False):
        W, H = self.W, self.H
        depth = np.empty((H, W), np.float32)
        cost = np.empty((H, W), np.float32)
        idx = np.empty((H, W), np.int32)
        vol = np.empty((self.D, H, W), np.uint32) if want_volume else None
        self._check(self.lib.mvs_sweep_fetch(self.h, _ptr(depth, _fp), _ptr(cost, _fp), _ptr(idx, _i32p),
                                             _ptr(vol, _u32p) if want_volume else None))
        return depth, cost, idx, vol

    def sweep_view_matrices(self):
        q = np.empty((self.V, 3, 4), np.float32)
        self._check(self.lib.mvs_sweep_view_matrices(self.h, _ptr(q, _fp)))
        return q

    # ---- profiling ------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.mvs_profile_enable(self.h, 1 if on else 0))

    def profile_read(self, reset=True):
        ms = (C.c_float * MVS_K_COUNT)()
        n = (C.c_int * MVS_K_COUNT)()
        self._check(self.lib.mvs_profile_read(self.h, ms, n, 1 if reset else 0))
        return list(ms), list(n)

    # ---- renderer / helpers ---------------------------------------------------------------------
    def load_mesh(self, verts4, faces3):
        v = _f32(verts4)
        f = np.ascontiguousarray(faces3, dtype=np.int32)
        assert v.ndim == 2 and v.shape[1] == 4 and f.ndim == 2 and f.shape[1] == 3
        self._check(self.lib.mvs_load_mesh(self.h, _ptr(v, _fp), v.shape[0], _ptr(f, _i32p), f.shape[0]))

    def depth(self, cam):
        cam = _f32(cam, (4, 4))
        out = np.empty((self.H, self.W), np.float32)
        self._check(self.lib.mvs_depth(self.h, _ptr(cam, _fp), _ptr(out, _fp)))
        return out

    def depth_probe(self, cam, rows, cols):
        cam = _f32(cam, (4, 4))
        rows = np.ascontiguousarray(rows, np.int32).ravel()
        cols = np.ascontiguousarray(cols, np.int32).ravel()
        if rows.shape != cols.shape:
            raise ValueError("rows and cols differ in length")
        out = np.empty(rows.shape[0], np.float32)
        self._check(self.lib.mvs_depth_probe(self.h, _ptr(cam, _fp), rows.shape[0], _ptr(rows, _i32p), _ptr(cols, _i32p), _ptr(out, _fp)))
        return out

    def resize(self, img, dw, dh):
        """cv::resize(img, Size(dw, dh)) INTER_LINEAR on u8 (configuration.cpp:233)"""
        img = np.ascontiguousarray(img, np.uint8)
        sh, sw = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, ch), np.uint8)
        self._check(self.lib.mvs_resize_u8(self.h, img.ctypes.data_as(C.c_void_p), sw, sh, ch, out.ctypes.data_as(C.c_void_p), int(dw), int(dh)))
        return out

    def set_texture_filter(self, name):
        """"mipmap" (default: the reference's GL_LINEAR_MIPMAP_LINEAR request) or "level0" for Render::projected's frame texture"""
        self._check(self.lib.mvs_set_texture_filter(self.h, {"mipmap": 0, "level0": 1}[name]))

    def projected(self, cam, frame, projector):
        cam = _f32(cam, (4, 4))
        prj = _f32(projector, (4, 4))
        frame = _u8(frame, (self.H, self.W))
        out = np.empty((self.H, self.W, 3), np.uint8)
        self._check(self.lib.mvs_projected(self.h, _ptr(cam, _fp), _ptr(frame, _u8p), _ptr(prj, _fp), _ptr(out, _u8p)))
        return out

    def mix_background(self, img3, bg, depth):
        img3 = _u8(img3, (self.H, self.W, 3))
        bg = _u8(bg, (self.H, self.W))
        depth = _f32(depth, (self.H, self.W)).copy()
        out = np.empty((self.H, self.W), np.uint8)
        self._check(self.lib.mvs_mix_background(self.h, _ptr(img3, _u8p), _ptr(bg, _u8p), _ptr(depth, _fp),
                                                _ptr(out, _u8p)))
        return out, depth

    def compare(self, prev, nxt):
        prev = _u8(prev, (self.H, self.W))
        nxt = _u8(nxt, (self.H, self.W))
        out = np.empty((self.H, self.W), np.float32)
        self._check(self.lib.mvs_compare(self.h, _ptr(prev, _u8p), _ptr(nxt, _u8p), _ptr(out, _fp)))
        return out

    def flow_remap(self, flow, image):
        flow = _f32(flow)
        assert flow.shape[:2] == (self.H, self.W)
        image = _u8(image, (self.H, self.W))
        out = np.empty((self.H, self.W), np.uint8)
        self._check(self.lib.mvs_flow_remap(self.h, _ptr(flow, _fp), flow.shape[2], _ptr(image, _u8p), _ptr(out, _u8p)))
        return out

    def flow(self, prev, nxt, use_farneback):
        prev = _u8(prev, (self.H, self.W))
        nxt = _u8(nxt, (self.H, self.W))
        out = np.empty((self.H, self.W, 4), np.float32)
        self._check(self.lib.mvs_flow(self.h, _ptr(prev, _u8p), _ptr(nxt, _u8p), 1 if use_farneback else 0,
                                      _ptr(out, _fp)))
        return out

    def triangulate(self, flows, main_cam, side_cams, depth, copy=True):
        """mvs_triangulate -> (N, 7) rows (x, y, z, w, nx, ny, nz) in pixel scan order (copy=False: a view of the context's
        reusable output buffer, valid until the next triangulate / process_frame call -- see process_frame)"""
        V = len(flows)
        fl = [_f32(f, (self.H, self.W, 4)) for f in flows]
        arr = (_fp * max(V, 1))(*[_ptr(f, _fp) for f in fl])
        cam = _f32(main_cam, (4, 4))
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((1, 4, 4), np.float32)
        depth = _f32(depth, (self.H, self.W))
        if getattr(self, "_pf_out", None) is None:
            self._pf_out = np.zeros((self.H * self.W, 7), np.float32)
        out = self._pf_out
        n = C.c_int(0)
        self._check(self.lib.mvs_triangulate(self.h, V, arr, _ptr(cam, _fp), _ptr(cams, _fp), _ptr(depth, _fp), _ptr(out, _fp),
                                             C.byref(n)))
        return out[:n.value].copy() if copy else out[:n.value]

    def process_frame(self, main_cam, main_frame, side_cams, side_frames, use_farneback=False, want_depth=False, copy=True):
        """mvs_process_frame: recon.cpp:65-117 for one main frame -> (N, 7) points [, depth after mixBackground].
        The library writes into a caller-owned H*W x 7 buffer; this wrapper keeps ONE such buffer per context (a fresh 58 MB
        numpy array per call at 1080p costs more in page faults than the whole GPU pipeline).  copy=True returns an
        independent array; copy=False returns a view of that buffer, valid until the next call -- what a C++ caller reusing
        its cv::Mat gets, and what the timing scripts use."""
        V = len(side_frames)
        cam = _f32(main_cam, (4, 4))
        mf = _u8(main_frame, (self.H, self.W))
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((1, 4, 4), np.float32)
        frames = [_u8(f, (self.H, self.W)) for f in side_frames]
        arr = (_u8p * max(V, 1))(*[_ptr(f, _u8p) for f in frames])
        if getattr(self, "_pf_out", None) is None:
            self._pf_out = np.zeros((self.H * self.W, 7), np.float32)   # zeros: touch every page once, here
        out = self._pf_out
        depth = np.empty((self.H, self.W), np.float32) if want_depth else None
        n = C.c_int(0)
        self._check(self.lib.mvs_process_frame(self.h, _ptr(cam, _fp), _ptr(mf, _u8p), V, _ptr(cams, _fp), arr,
                                               1 if use_farneback else 0, _ptr(out, _fp), C.byref(n),
                                               _ptr(depth, _fp) if want_depth else None))
        pts = out[:n.value].copy() if copy else out[:n.value]
        return (pts, depth) if want_depth else pts

    def process_frame_slots(self, main_cam, main_slot, side_cams, side_slots, use_farneback=False, want_depth=False, copy=True):
        """mvs_process_frame_slots: process_frame() with the frames taken from the frame store (frame_store / frame_upload)."""
        V = len(side_slots)
        cam = _f32(main_cam, (4, 4))
        cams = _f32(np.asarray(side_cams, dtype=np.float32).reshape(V, 4, 4)) if V else np.zeros((1, 4, 4), np.float32)
        slots = np.ascontiguousarray(np.asarray(list(side_slots) if V else [0], dtype=np.int32))
        if getattr(self, "_pf_out", None) is None:
            self._pf_out = np.zeros((self.H * self.W, 7), np.float32)
        out = self._pf_out
        depth = np.empty((self.H, self.W), np.float32) if want_depth else None
        n = C.c_int(0)
        self._check(self.lib.mvs_process_frame_slots(self.h, _ptr(cam, _fp), int(main_slot), V, _ptr(cams, _fp), _ptr(slots, _i32p),
                                                     1 if use_farneback else 0, _ptr(out, _fp), C.byref(n),
                                                     _ptr(depth, _fp) if want_depth else None))
        pts = out[:n.value].copy() if copy else out[:n.value]
        return (pts, depth) if want_depth else pts

    def filter_points(self, points4, alpha):
        """mvs_filter_points -> ascending indices of the points that survive"""
        pts = _f32(points4)
        assert pts.ndim == 2 and pts.shape[1] == 4
        keep = np.empty(max(pts.shape[0], 1), np.int32)
        n = C.c_int(0)
        self._check(self.lib.mvs_filter_points(self.h, _ptr(pts, _fp), pts.shape[0], float(alpha), _ptr(keep, _i32p), C.byref(n)))
        return keep[:n.value].copy()

    def onecall_bands(self):
        """test hook: the row bands the last one-call sweep() of this context was pipelined in (0: the unbanded path)"""
        return self.lib.mvs_test_onecall_bands(self.h)

    def test_rcp(self, exp_bits):
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self.lib.mvs_test_rcp(self.h, exp_bits, C.byref(a), C.byref(b)))
        return a.value, b.value
