"""In-process timings of the exact sampler's rectified kernel (sweep_exact_rect) against sweep_tiled on the SURVEY 8d ring.
    python tools/exp_xrect.py [c2|c3|c5] [--exp]      (--exp: needs a -DMVS_XR_EXPERIMENTS build, tools/build_variant.sh xrexp -DMVS_XR_EXPERIMENTS csrc/sweep_xrect.hip)"""
import os, sys, time
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # master switch of the library's environment hooks (csrc/hooks.hpp)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch  # noqa: F401
import mvs_amd
from mvs_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c3"
W, H, D, V = {"c1": (640, 480, 32, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16), "c5": (640, 480, 128, 4)}[name]
mc, mi, sc, si, gt = synth.make_views(W, H, V, radius=0.15)
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
with mvs_amd.Context(W, H, sampler="exact") as ctx:
    ctx.sweep_set(mc, mi, sc, si, D)

    def t(flags, n=20):
        for _ in range(3):
            ctx.sweep_run(0, V, flags)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.sweep_run(0, V, flags)
        ctx.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t(both, 30)
    print(name, "exact sampler: rectified %.3f ms (shape %d), fused only %.3f, volume only %.3f | sweep_tiled %.3f ms" %
          (t(both), ctx.plan_shape(), t(mvs_amd.MVS_SWEEP_FUSED_ARGMIN), t(mvs_amd.MVS_SWEEP_VOLUME), t(both | mvs_amd.MVS_SWEEP_NO_RECT)))
    print("   splits:", " ".join("%d: %.3f" % (s, t(both | (s << 16))) for s in (1, 2, 4, 8)))
    if "--exp" in sys.argv:
        for label, bits in (("full", 0), ("no copies", 1), ("no sampling", 2), ("neither", 3), ("every copy from one box", 8), ("per-row values for free", 16)):
            print("   %-28s %.3f ms" % (label, t(both | (bits << 8))))
