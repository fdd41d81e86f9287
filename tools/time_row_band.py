#!/usr/bin/env python3
"""Single-GPU projection of the row sharding (bench.py --shard rows): time mvs_sweep_run_rows for each band of an
8-way split of one main view, and check the zero-copy torch view of the depth map the all-gather uses.
No collective is executed here (one GPU); the slowest band bounds the multi-GPU step from below."""
import json, os, sys, time
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # master switch of the library's environment hooks (csrc/hooks.hpp)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")  # the library masks the experiment bits of `flags` otherwise
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch
import mvs_amd
from mvs_amd import synth, dist as mdist

def run(name, W, H, D, V, world, data):
    if data == "scene":
        main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.15)
    else:
        main_cam, main_img, side_cams, sides = synth.noise_views(W, H, V, seed=synth.SEED_NOISE)
    flags = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    with mvs_amd.Context(W, H) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        for _ in range(3):
            ctx.sweep_run(0, V, flags)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.sweep_run(0, V, flags)
        ctx.synchronize()
        full_ms = (time.perf_counter() - t0) / 10 * 1e3
        full_depth = ctx.sweep_fetch()[0].copy()
        alias = torch.as_tensor(ctx.depth_device_array(), device="cuda")
        alias_ok = bool(np.array_equal(alias.cpu().numpy(), full_depth))
        bands = mdist.equal_row_bands(H, world, ctx.row_granularity())
        band_ms = []
        for a, n in bands:
            for _ in range(3):
                ctx.sweep_run_rows(a, n, 0, V, flags)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.sweep_run_rows(a, n, 0, V, flags)
            ctx.synchronize()
            band_ms.append((time.perf_counter() - t0) / 20 * 1e3)
        same = bool(np.array_equal(ctx.sweep_fetch()[0], full_depth))
    print(json.dumps({"config": name, "data": data, "world": world, "full_ms": full_ms, "bands": bands, "band_ms": band_ms,
                      "slowest_band_ms": max(band_ms), "ideal_ms": full_ms / world, "depth_gather_bytes": 4 * W * H,
                      "bands_reproduce_full_depth": same, "torch_alias_matches": alias_ok}))

run("c3", 1920, 1080, 128, 16, 8, "scene")
run("c3", 1920, 1080, 128, 16, 2, "scene")
run("c4", 3840, 2160, 256, 32, 8, "noise")
