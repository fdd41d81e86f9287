#!/usr/bin/env python3
"""the stage the reference itself executes per main frame (recon.cpp:65-117) --
depth, then per side view projected + mixBackground + calculateFlow, then triangulatePixels -- at 640x480 with the
cameras of tracks/koberec.yaml, 4 side views, synthetic frames (the clip is missing from the checkout).
GPU: mvs_process_frame.  CPU: the oracle port of the same stages on the host (single thread, like the reference).
Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import mvs_amd  # noqa: E402
import orc  # noqa: E402
import scenes  # noqa: E402
import tracks_yaml  # noqa: E402


def main():
    no_cpu = "--no-cpu" in sys.argv
    farneback = "--farneback" in sys.argv
    t = tracks_yaml.load("koberec.yaml")
    W, H = t["width"], t["height"]
    for a in sys.argv[1:]:   # --size=1920x1080: the projections are NDC, so the same cameras serve any frame size
        if a.startswith("--size="):
            W, H = (int(v) for v in a[len("--size="):].split("x"))
    cams = t["cameras"]
    main_i, side_is = 40, [30, 35, 45, 50]
    main = cams[main_i]
    sides = np.stack([cams[i] for i in side_is])
    # proxy mesh: a plane through the bundle cloud, in front of the koberec cameras
    b = t["bundles"]
    xyz = b[:, :3] / b[:, 3:4]
    c = xyz.mean(0)
    ext = 1.5 * np.abs(xyz - c).max()
    n = 48
    g = np.linspace(-ext, ext, n)
    X, Y = np.meshgrid(g, g)
    verts = np.stack([c[0] + X.ravel(), c[1] + Y.ravel(), np.full(n * n, c[2]), np.ones(n * n)], 1).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    a_, b_, c_, d_ = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a_, b_, c_], 1), np.stack([b_, d_, c_], 1)]).astype(np.int32)
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:H, 0:W]

    def frame(k):
        return (127 + 60 * np.sin((xx + 3 * k) / 19.0) * np.cos((yy - 2 * k) / 23.0) + rng.normal(0, 3, (H, W))).clip(0, 255).astype(np.uint8)
    main_img = frame(0)
    side_imgs = [frame(k + 1) for k in range(4)]

    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        pts = ctx.process_frame(main, main_img, sides, side_imgs, farneback)
        reps = 40
        samples = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.process_frame(main, main_img, sides, side_imgs, farneback, copy=False)   # the C ABI call into a reused buffer
            samples.append((time.perf_counter() - t0) * 1e3)
        gpu_ms = float(np.median(samples))
        info = ctx.info()

    if no_cpu:
        print(json.dumps({"gpu_ms_per_main_frame": gpu_ms, "gpu_ms_min_max": [min(samples), max(samples)], "farneback": farneback}))
        return
    o = orc.load()
    soup = o.load_mesh(verts, faces)
    t0 = time.perf_counter()
    d = o.depth(soup, main, W, H)
    flows = []
    for cam, img in zip(sides, side_imgs):
        mixed, d = o.mix_background(o.projected(soup, main, img, cam), main_img, d)
        flows.append(o.calculate_flow(main_img, mixed, farneback))
    o_pts = o.triangulate_pixels(flows, main, sides, d)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    same = pts.shape == o_pts.shape and np.array_equal(pts[:, :4], o_pts[:, :4], equal_nan=True)
    n_nan = int(np.isnan(o_pts[:, :4]).any(axis=1).sum())
    print(json.dumps({"workload": "reference stage: koberec.yaml cameras, %dx%d, 1 main + 4 side frames, %s flow" % (W, H, "Farneback" if farneback else "variational"),
                      "points": int(pts.shape[0]), "gpu_ms_per_main_frame": gpu_ms, "gpu_ms_min_max": [min(samples), max(samples)], "gpu_reps": reps, "cpu_port_ms_per_main_frame": cpu_ms,
                      "cpu_threads": 1, "speedup": cpu_ms / gpu_ms, "positions_bit_identical": bool(same), "points_with_nan_in_both": n_nan, "device": info}))


if __name__ == "__main__":
    main()
