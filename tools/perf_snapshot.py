"""One table of resident sweep times (volume + fused depth selection, frames in HBM) for the BASELINE shapes and both samplers, ring geometry
(plane-independent w) and the same with that shortcut disabled (the path general cameras take): python tools/perf_snapshot.py > out.json"""
import json, os, sys, time
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # master switch of the library's environment hooks (csrc/hooks.hpp)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")  # the library masks the experiment bits of `flags` otherwise
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch  # noqa: F401 (HIP runtime first)
import mvs_amd
from mvs_amd import synth

SHAPES = {"c1": (640, 480, 32, 4), "c5-size": (640, 480, 128, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16), "c4": (3840, 2160, 256, 32)}
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN


def timeit(ctx, V, flags, n):
    for _ in range(max(3, n // 2)):
        ctx.sweep_run(0, V, flags)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.sweep_run(0, V, flags)
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {}
for name, (W, H, D, V) in SHAPES.items():
    views = synth.noise_views(W, H, V) if name == "c4" else synth.make_views(W, H, V, radius=0.15)[:4]
    mc, mi, sc, si = views
    n = 5 if name == "c4" else 40
    for sampler in ("fixed", "exact"):
        with mvs_amd.Context(W, H, sampler=sampler) as ctx:
            ctx.sweep_set(mc, mi, sc, si, D)
            timeit(ctx, V, both, n)  # clocks
            ms, ms_general, ms_fused = timeit(ctx, V, both, n), timeit(ctx, V, both | mvs_amd.MVS_SWEEP_NO_RECT | (4 << 8), n), timeit(ctx, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN, n)
        samples = float(W) * H * D * V
        out["%s %s" % (name, sampler)] = {"shape": [W, H, D, V], "ms": round(ms, 4), "T_samples_per_s": round(samples / ms / 1e9, 3),
                                          "ms_general_cameras": round(ms_general, 4), "ms_depth_only_no_volume": round(ms_fused, 4)}
        print("%-8s %-5s %8.3f ms  %6.2f T samples/s   general %8.3f ms   no volume %8.3f ms" % (name, sampler, ms, samples / ms / 1e9, ms_general, ms_fused), file=sys.stderr)
print(json.dumps(out, indent=1))
