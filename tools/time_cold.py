"""The cold step alone (bench.py's `cold_step`): a new (main, views) set whose raw u8 frames are resident in HBM -- view matrices, quad
images from the raw frames, region plan, sweep with depth selection -- for a rocprofv3 kernel trace of what the step is made of, and
the same through mvs_sweep_handles over the frame store (plan + sweep + depth download).
usage: python3 tools/time_cold.py [c2|c3] [steps] [--general]"""
import sys
import time

sys.path.insert(0, "mesh-reconstruction_amd/python")
import numpy as np
import torch

import mvs_amd
from mvs_amd import synth

cfg = {"c1": (640, 480, 32, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16)}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
W, H, D, V = cfg
main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.15)
if "--general" in sys.argv:
    cams = []
    for vi in range(V):
        ang = 2.0 * np.pi * vi / V
        yaw, pitch = 0.012 * np.cos(ang), 0.012 * np.sin(ang)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        rot = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        cams.append(synth.camera_at([0.15 * np.cos(ang), 0.15 * np.sin(ang), 0.0], W, H, rot=rot))
    side_cams = np.stack(cams)
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
raw_main = torch.as_tensor(main_img, device="cuda")
raw_sides = [torch.as_tensor(s, device="cuda") for s in sides]
ptrs = [t.data_ptr() for t in raw_sides]
torch.cuda.synchronize()
with mvs_amd.Context(W, H) as ctx:
    ctx.sweep_set_planes(D)

    def once():
        ctx.sweep_set_main_device(main_cam, raw_main.data_ptr())
        ctx.sweep_set_views_device(side_cams, ptrs)
        ctx.sweep_run(0, V, both)
    for _ in range(30):
        once()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    ctx.synchronize()
    cold = (time.perf_counter() - t0) / steps * 1e3
    for _ in range(5):
        ctx.sweep_run(0, V, both)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.sweep_run(0, V, both)
    ctx.synchronize()
    warm = (time.perf_counter() - t0) / steps * 1e3
    print("cold step %.4f ms, resident step %.4f ms (plan shape %d)" % (cold, warm, ctx.plan_shape()))
    ctx.frame_store(V + 1)
    ctx.frame_upload(0, main_img)
    for v in range(V):
        ctx.frame_upload(1 + v, sides[v])
    out = mvs_amd.pinned_array((H, W), np.float32)
    for _ in range(5):
        ctx.sweep_handles(0, main_cam, list(range(1, V + 1)), side_cams, D, out=out)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.sweep_handles(0, main_cam, list(range(1, V + 1)), side_cams, D, out=out)
    print("mvs_sweep_handles %.4f ms per main view (plan + sweep with depth selection, no volume + depth download to pinned memory)" % ((time.perf_counter() - t0) / steps * 1e3))
