"""In-process timings of the rectified-view kernel (sweep_fx_rect) against the general tiled kernel on the SURVEY 8d ring.
    python tools/exp_rect.py [c1|c2|c3|c4|c5] [--check]
Environment: MVS_RECT_SLOTS=S (LDS slots = look-ahead + 1)."""
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

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c3"
cfg = {"c1": (640, 480, 32, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16), "c4": (3840, 2160, 256, 32), "c5": (640, 480, 128, 4)}[name]
W, H, D, V = cfg
mc, mi, sc, si, gt = synth.make_views(W, H, V, radius=0.15)
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
NR = mvs_amd.MVS_SWEEP_NO_RECT


def timeit(ctx, flags, n=20):
    for _ in range(3):
        ctx.sweep_run(0, V, flags)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.sweep_run(0, V, flags)
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with mvs_amd.Context(W, H, sampler="fixed") as ctx:
    ctx.sweep_set(mc, mi, sc, si, D)
    timeit(ctx, both, 30)
    print(name, "plan shape", ctx.plan_shape(), "slots", os.environ.get("MVS_RECT_SLOTS", "default"))
    print("rect:    volume+fused %.3f ms  fused only %.3f  volume only %.3f" % (timeit(ctx, both), timeit(ctx, mvs_amd.MVS_SWEEP_FUSED_ARGMIN), timeit(ctx, mvs_amd.MVS_SWEEP_VOLUME)))
    print("general: volume+fused %.3f ms  fused only %.3f  volume only %.3f" % (timeit(ctx, both | NR), timeit(ctx, mvs_amd.MVS_SWEEP_FUSED_ARGMIN | NR), timeit(ctx, mvs_amd.MVS_SWEEP_VOLUME | NR)))
    print("rect splits:", " ".join("%d: %.3f" % (s, timeit(ctx, both | (s << 16))) for s in (1, 2, 4, 8)))
    if "--check" in sys.argv:
        ctx.sweep_run(0, V, both)
        a = ctx.sweep_fetch(want_volume=False)
        ctx.sweep_run(0, V, both | NR)
        b = ctx.sweep_fetch(want_volume=False)
        print("depth equal", np.array_equal(a[0], b[0]), "index differs at", int(np.count_nonzero(a[2] != b[2])), "cost differs at", int(np.count_nonzero(a[1] != b[1])))
    if "--exp" in sys.argv:   # needs a build with -DMVS_RX_EXPERIMENTS (make -B CXXFLAGS="... -DMVS_RX_EXPERIMENTS")
        for name_, bits in (("full", 0), ("every plane taken for FULL", 32), ("no copies", 1), ("every copy from one box (L2 hits)", 8), ("one box, no sampling", 10), ("no sampling", 2), ("no barrier", 4), ("no epilogue", 16), ("no copies, no sampling", 3), ("no sampling, no epilogue", 18),
                            ("no copies, no barrier", 5), ("skeleton: none of the four", 23)):
            print("   %-36s %.3f ms  (fused only %.3f, volume only %.3f)" % (name_, timeit(ctx, both | (bits << 8)), timeit(ctx, mvs_amd.MVS_SWEEP_FUSED_ARGMIN | (bits << 8)), timeit(ctx, mvs_amd.MVS_SWEEP_VOLUME | (bits << 8))))
