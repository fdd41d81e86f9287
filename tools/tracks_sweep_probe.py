#!/usr/bin/env python3
"""Sweep rate and planner choice on the cameras of each bundled tracks file (640x480, 128 planes, 4 neighbours at +-5/+-10% of
the sequence), frames resident: checks that real baselines / rotations do not fall onto the oversize -> generic path."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import tempfile
import numpy as np
import torch  # noqa: F401 (HIP runtime first)
PLAN = os.path.join(tempfile.gettempdir(), "mvs_plan_tracks.bin")
os.environ["MVS_TEST_HOOKS"] = "1"
os.environ["MVS_PLAN_DUMP"] = PLAN
import mvs_amd
from mvs_amd import tracks
rng = np.random.default_rng(2)
for name in ("koberec.yaml", "koberec-.yaml", "zatisi.yaml", "koule-tr.yaml"):
    t = tracks.load(name)
    W, H, cams = t["width"], t["height"], t["cameras"]
    n = len(cams); m = n // 2; s = max(1, n // 12)
    ids = [m - 2 * s, m - s, m + s, m + 2 * s]
    imgs = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(5)]
    D, V = 128, 4
    flags = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    with mvs_amd.Context(W, H) as ctx:
        ctx.sweep_set(cams[m], imgs[0], np.stack([cams[i] for i in ids]), imgs[1:], D)
        for _ in range(3): ctx.sweep_run(0, V, flags)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ctx.sweep_run(0, V, flags)
        ctx.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
        depth, cost, idx, vol = ctx.sweep_fetch(want_volume=True)
        raw = np.fromfile(PLAN, dtype=np.uint32)[4:].reshape(-1, 2)   # the fixed sampler's region plan (MVS_PLAN_DUMP)
        mode, rw, rh = (raw[:, 1] >> 16) & 7, raw[:, 1] & 0xff, (raw[:, 1] >> 8) & 0xff
        staged = (mode == 1) | (mode == 2)
        print(json.dumps({"tracks": name, "frames": n, "plan_shape": ctx.plan_shape(), "ms": ms, "T_samples_per_s": W * H * D * V / ms / 1e9,
                          "in_frame_fraction": float((idx >= 0).mean()), "mean_views_in_frame": float((vol >> 24).mean()),
                          "regions_skip_fast_border_generic_pct": [round(100 * float((mode == k).mean()), 2) for k in range(4)],
                          "region_width_mean_max": [float(rw[staged].mean()), int(rw[staged].max())] if staged.any() else None,
                          "region_height_mean_max": [float(rh[staged].mean()), int(rh[staged].max())] if staged.any() else None,
                          "regions_wider_than_half_row_pct": round(100 * float((rw[staged] > 112).mean()), 2) if staged.any() else None}))
