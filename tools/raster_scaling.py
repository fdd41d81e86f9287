#!/usr/bin/env python3
"""Render::depth / Render::projected time against mesh size (640x480): the reference's later iterations render Poisson surfaces
with 10^5..10^6 faces, 200 + N_main times per outer iteration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mvs_amd, scenes
from mvs_amd import synth
W, H = 640, 480
cam, side = synth.camera_at([0, 0, 0], W, H), synth.camera_at([0.15, 0, 0], W, H)
img = np.random.default_rng(0).integers(0, 256, (H, W), dtype=np.uint8)
for n in (64, 128, 256, 512, 724):
    verts, faces = scenes.heightfield_mesh(n)
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        rows, cols = np.array([H // 2], np.int32), np.array([W // 2], np.int32)
        for _ in range(3): ctx.depth_probe(cam, rows, cols)
        t0 = time.perf_counter()
        for _ in range(20): ctx.depth_probe(cam, rows, cols)          # raster only: one pixel comes back
        td = (time.perf_counter() - t0) / 20 * 1e3
        for _ in range(2): ctx.projected(cam, img, side)
        t0 = time.perf_counter()
        for _ in range(10): ctx.projected(cam, img, side)
        tp = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"faces": int(faces.shape[0]), "depth_ms": td, "projected_ms": tp}))
