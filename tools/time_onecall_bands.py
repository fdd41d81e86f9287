import sys, time, os
sys.path.insert(0, 'mesh-reconstruction_amd/python')
import numpy as np, torch, mvs_amd
from mvs_amd import synth
W, H, D, V = 1920, 1080, 128, 16
mc, mi, sc, si = synth.noise_views(W, H, V)
with mvs_amd.Context(W, H) as ctx:
    frames = [mvs_amd.pinned_array((H, W), np.uint8) for _ in range(V + 1)]
    for dst, src in zip(frames, [mi] + list(si)):
        dst[...] = src
    mi, si = frames[0], frames[1:]
    ctx._pinned_depth = mvs_amd.pinned_array((H, W), np.float32)
    for env in (None, "1", None, "1"):
        if env: os.environ["MVS_SWEEP_NO_PIPELINE"] = env
        else: os.environ.pop("MVS_SWEEP_NO_PIPELINE", None)
        ctx.sweep(mc, mi, sc, si, D); ctx.sweep(mc, mi, sc, si, D)
        t0 = time.perf_counter()
        for _ in range(10): ctx.sweep(mc, mi, sc, si, D)
        dt = (time.perf_counter() - t0) / 10
        print("pinned, NO_PIPELINE=%s: %.3f ms, bands %d" % (env, dt * 1e3, ctx.lib.mvs_test_onecall_bands(ctx.h)))
    # resident band sweeps
    ctx.sweep_set(mc, mi, sc, si, D)
    fo = mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    for nb in (1, 4):
        bands = [(H * b // nb // 8 * 8, (H * (b + 1) // nb // 8 * 8 if b < nb - 1 else H)) for b in range(nb)]
        for _ in range(3):
            for a, e in bands: ctx.sweep_run_rows(a, e - a, 0, V, fo)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            for a, e in bands: ctx.sweep_run_rows(a, e - a, 0, V, fo)
        ctx.synchronize()
        print("resident, %d band(s), fused only: %.3f ms" % (nb, (time.perf_counter() - t0) / 10 * 1e3))
