"""One-call mvs_sweep at BASELINE c3 (host frames in, host depths out) for 1..8 row bands of its upload pipeline (sweep.hip: BandPipeline),
pageable and pinned host buffers, general and rectified cameras.  Usage: python tools/time_onecall.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mesh-reconstruction_amd", "python"))
os.environ["MVS_TEST_HOOKS"] = "1"
import numpy as np

import mvs_amd
from mvs_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H, D, V = 1920, 1080, 128, 16
main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V)
turned = side_cams.copy()
for v in range(V):
    a = 2.0 * np.pi * v / V
    cy, sy, cp, sp = np.cos(0.012), np.sin(0.012), np.cos(-0.012), np.sin(-0.012)
    R = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    turned[v] = synth.camera_at([0.15 * np.cos(a), 0.15 * np.sin(a), 0.0], W, H, rot=R)
out = {}
for name, cams in (("rectified", side_cams), ("general", turned)):
    ref = None
    for bands in (1, 2, 3, 4, 5, 6, 8):
        os.environ["MVS_ONECALL_BANDS"] = str(bands)
        with mvs_amd.Context(W, H, sampler="fixed") as ctx:
            for _ in range(3):
                depth, cost = ctx.sweep(main_cam, main_img, cams, sides, D, want_cost=True)
            if ref is None:
                ref = (depth.copy(), cost.copy())
            assert np.array_equal(depth, ref[0]) and np.array_equal(cost, ref[1]), (name, bands)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                ctx.sweep(main_cam, main_img, cams, sides, D)
                ts.append(time.perf_counter() - t0)
            ctx.set_plan_cache(False)
            tc = []
            for _ in range(reps):
                t0 = time.perf_counter()
                ctx.sweep(main_cam, main_img, cams, sides, D)
                tc.append(time.perf_counter() - t0)
            out["%s bands=%d" % (name, bands)] = {"ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * float(np.min(ts)),
                                                  "ms_median_planning_every_call": 1e3 * float(np.median(tc)), "used": ctx.onecall_bands()}
# page-locked buffers (mvs_host_alloc) on both sides: the copies are asynchronous DMA, the pipeline's best case
import ctypes as C
lib = mvs_amd.load_library()
pm = mvs_amd.pinned_array((H, W), np.uint8)
pm[:] = main_img
ps = [mvs_amd.pinned_array((H, W), np.uint8) for _ in range(V)]
for a, b in zip(ps, sides):
    a[:] = b
pd, pc = mvs_amd.pinned_array((H, W), np.float32), mvs_amd.pinned_array((H, W), np.float32)
u8p, fp = C.POINTER(C.c_ubyte), C.POINTER(C.c_float)
arr = (u8p * V)(*[a.ctypes.data_as(u8p) for a in ps])
cam = np.ascontiguousarray(main_cam, np.float32)
for name, cams in (("rectified pinned", side_cams), ("general pinned", turned)):
    cc = np.ascontiguousarray(cams, np.float32)
    for bands in (1, 2, 3, 4, 6, 8):
        os.environ["MVS_ONECALL_BANDS"] = str(bands)
        with mvs_amd.Context(W, H, sampler="fixed") as ctx:
            def call():
                rc = lib.mvs_sweep(ctx.h, cam.ctypes.data_as(fp), pm.ctypes.data_as(u8p), V, cc.ctypes.data_as(fp), arr, D, -1.0, 1.0, pd.ctypes.data_as(fp), pc.ctypes.data_as(fp), None)
                assert rc == 0, rc
            for _ in range(3):
                call()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                call()
                ts.append(time.perf_counter() - t0)
            out["%s bands=%d" % (name, bands)] = {"ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * float(np.min(ts)), "used": ctx.onecall_bands()}
print(json.dumps(out, indent=1))
