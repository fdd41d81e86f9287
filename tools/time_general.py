"""The general tiled kernel of the fixed sampler at c3 / c2 / c5-size: ring geometry with MVS_SWEEP_NO_RECT (hoisted reciprocal) and cameras
turned by 12 mrad (the bench's general_camera_path), volume + fused / fused only, with a bit-identity check against the rectified kernel on the ring."""
import os, sys, time
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # master switch of the library's environment hooks (csrc/hooks.hpp)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch  # noqa: F401
import mvs_amd
from mvs_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
W, H, D, V = {"c1": (640, 480, 32, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16), "c5": (640, 480, 128, 4)}[name]
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


def rot_cams():
    cams = []
    for vi in range(V):
        a = 2.0 * np.pi * vi / V
        yaw, pitch = 0.012 * np.cos(a), 0.012 * np.sin(a)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        rot = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        cams.append(synth.camera_at([0.15 * np.cos(a), 0.15 * np.sin(a), 0.0], W, H, rot=rot))
    return np.stack(cams)


with mvs_amd.Context(W, H, sampler="fixed") as ctx:
    ctx.sweep_set(mc, mi, sc, si, D)
    timeit(ctx, both, 30)
    print(name, "ring, general kernel: volume+fused %.3f ms  fused only %.3f  volume only %.3f   (rect %.3f)" %
          (timeit(ctx, both | NR), timeit(ctx, mvs_amd.MVS_SWEEP_FUSED_ARGMIN | NR), timeit(ctx, mvs_amd.MVS_SWEEP_VOLUME | NR), timeit(ctx, both)))
    ctx.sweep_run(0, V, both)
    a = ctx.sweep_fetch(want_volume=False)
    ctx.sweep_run(0, V, both | NR)
    b = ctx.sweep_fetch(want_volume=False)
    print("   general == rect:", all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])))
    ctx.sweep_set(mc, mi, rot_cams(), si, D)
    timeit(ctx, both, 10)
    print(name, "rotated cameras:      volume+fused %.3f ms  fused only %.3f  volume only %.3f   (plan shape %d)" %
          (timeit(ctx, both), timeit(ctx, mvs_amd.MVS_SWEEP_FUSED_ARGMIN), timeit(ctx, mvs_amd.MVS_SWEEP_VOLUME), ctx.plan_shape()))
