"""The facet-criteria report of poissonSurface on a cloud the pipeline itself produces (config-5 outer iteration: point blocks of a few zatisi
main frames -> filterPoints -> Poisson).  python tools/criteria_on_c5.py [--where] [--use-precision]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import c5_common
import mvs_amd

seq = c5_common.Sequence()
with mvs_amd.Context(seq.W, seq.H) as ctx:
    ctx.load_mesh(seq.verts, seq.faces)
    cloud = np.concatenate([c5_common.process_main_frame(ctx, seq, f)[2] for f in seq.mains[10:14]])
    cloud = cloud[::max(1, len(cloud) // 50000)]
    xyz = cloud[:, :3] / cloud[:, 3:4]
    extent = float(np.percentile(xyz, 95, axis=0).max() - np.percentile(xyz, 5, axis=0).min())
    keep = ctx.filter_points(cloud[:, :4], 0.01 * extent)
    pts, nrm = cloud[keep, :4], cloud[keep, 4:7]
    precision = "--use-precision" in sys.argv   # keep the pdf's lengths as confidences (pcl.cpp:198-202's USE_PRECISION) instead of unit normals
    raw_v, raw_f = mvs_amd.poisson_surface(pts, nrm, criteria=None, use_precision=precision)
    rep = {}
    v, f = mvs_amd.poisson_surface(pts, nrm, report=rep, use_precision=precision)
    print("samples", len(pts), "raw", len(raw_v), len(raw_f), "->", len(v), len(f), rep)
    if "--where" in sys.argv:
        from scipy.spatial import cKDTree
        sp = rep["average_spacing"]
        d, _ = cKDTree(pts[:, :3] / pts[:, 3:4]).query(raw_v[::50, :3])
        print("raw vertices: distance to the nearest sample in spacings: percentiles 10/50/90/99", np.round(np.percentile(d / sp, [10, 50, 90, 99]), 2),
              "share within 3 spacings %.3f" % (d < 3 * sp).mean())
        lo, hi = xyz.min(0), xyz.max(0)
        print("cloud extent", np.round(hi - lo, 3), "robust extent", round(extent, 3), "spacing", sp)
