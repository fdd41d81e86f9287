"""In-process A/B timings of the sweep kernels (timing experiments only: some variants give wrong results).  Most variants of the fixed
sampler exist only in a library built with -DMVS_FX_EXPERIMENTS (make -B CXXFLAGS="... -DMVS_FX_EXPERIMENTS"), which itself runs 3-5 %
slower than the production build; with the production build they time the unmodified kernel.
debug bits (flags >> 8): 1 skip LDS staging*, 2 linear tile order, 4 never use the plane-independent-w path, 8 no region look-ahead, 16 treat BORDER regions as FAST*, 32 skip the sample loop*, 64 no per-view barrier*, 128 no chunk epilogue* (* = experiments build only for the fixed sampler); flags >> 16: forced plane splits"""
import sys, os
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # master switch of the library's environment hooks (csrc/hooks.hpp)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")  # the library masks the experiment bits of `flags` otherwise
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch  # noqa: F401 (HIP runtime first)
import mvs_amd
from mvs_amd import synth

cfg = {"c1": (640, 480, 32, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16), "c5": (640, 480, 128, 4)}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
W, H, D, V = cfg
mc, mi, sc, si, gt = synth.make_views(W, H, V, radius=0.15)
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN


def timeit(ctx, flags, n=10):
    for _ in range(3):
        ctx.sweep_run(0, V, flags)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.sweep_run(0, V, flags)
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for sampler in ("fixed", "exact"):
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(mc, mi, sc, si, D)
        timeit(ctx, both, 30)  # clocks up
        print(sampler, "volume+fused %.3f ms" % timeit(ctx, both), " fused only %.3f" % timeit(ctx, mvs_amd.MVS_SWEEP_FUSED_ARGMIN),
              " volume only %.3f" % timeit(ctx, mvs_amd.MVS_SWEEP_VOLUME), " no staging %.3f" % timeit(ctx, both | (1 << 8)),
              " general %.3f" % timeit(ctx, both | (4 << 8)), " general, no staging %.3f" % timeit(ctx, both | (5 << 8)),
              " linear tiles %.3f" % timeit(ctx, both | (2 << 8)), (" without look-ahead %.3f" % timeit(ctx, both | (8 << 8)) + " border as fast %.3f" % timeit(ctx, both | (16 << 8)) + " no sample loop %.3f" % timeit(ctx, both | (32 << 8)) + " no loop, no staging %.3f" % timeit(ctx, both | (33 << 8)) + " no per-view barrier %.3f" % timeit(ctx, both | (64 << 8))) if sampler == "fixed" else "")
        if sampler == "fixed":
            fo, vo = mvs_amd.MVS_SWEEP_FUSED_ARGMIN, mvs_amd.MVS_SWEEP_VOLUME
            print("   no sample loop: fused only %.3f  volume only %.3f  | no loop, no staging: fused only %.3f  volume only %.3f | no loop, no look-ahead %.3f" %
                  (timeit(ctx, fo | (32 << 8)), timeit(ctx, vo | (32 << 8)), timeit(ctx, fo | (33 << 8)), timeit(ctx, vo | (33 << 8)), timeit(ctx, both | (40 << 8))))
            print("   skeleton (no loop, no staging, fused only) %.3f: border as fast %.3f  + no epilogue %.3f  + no per-view barrier %.3f" %
                  (timeit(ctx, fo | (33 << 8)), timeit(ctx, fo | ((33 + 16) << 8)), timeit(ctx, fo | ((33 + 16 + 128) << 8)), timeit(ctx, fo | ((33 + 16 + 128 + 64) << 8))))
        print("   splits:", " ".join("%d: %.3f" % (s, timeit(ctx, both | (s << 16))) for s in (1, 2, 4, 8)))
