"""Where a main frame's time goes in the frame-store path of c5 (uploads, batch calls), against mvs_sweep per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch  # noqa: F401
import mvs_amd
from mvs_amd import tracks

W, H, D = 640, 480, 128
cams = tracks.load("zatisi.yaml")["cameras"]
n = len(cams)
rng = np.random.default_rng(1)
frames = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(n)]
sides = lambda f: [min(n - 1, max(0, f + o)) for o in (-10, -5, 5, 10)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with mvs_amd.Context(W, H) as ctx:
    ctx.frame_store(n)
    out = mvs_amd.pinned_array((B, H, W), np.float32)
    for rep in range(3):
        t0 = time.perf_counter()
        for f in range(n):
            ctx.frame_upload(f, frames[f])
        ctx.synchronize()
        t1 = time.perf_counter()
        tb = []
        for i in range(0, n, B):
            mb = list(range(i, min(n, i + B)))
            a = time.perf_counter()
            ctx.sweep_batch(mb, np.stack([cams[f] for f in mb]), np.array([sides(f) for f in mb], np.int32),
                            np.stack([np.stack([cams[j] for j in sides(f)]) for f in mb]), D, out=out[:len(mb)])
            tb.append(time.perf_counter() - a)
        t2 = time.perf_counter()
        print("rep %d: uploads %.2f ms (%.3f per frame), batches %.2f ms (%.3f per frame; per call min %.3f median %.3f max %.3f ms)" %
              (rep, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / n, (t2 - t1) * 1e3, (t2 - t1) * 1e3 / n, min(tb) * 1e3, sorted(tb)[len(tb) // 2] * 1e3, max(tb) * 1e3))
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    mb = list(range(40, 40 + B))
    for _ in range(5):
        ctx.sweep_batch(mb, np.stack([cams[f] for f in mb]), np.array([sides(f) for f in mb], np.int32), np.stack([np.stack([cams[j] for j in sides(f)]) for f in mb]), D, out=out)
    ms, cnt = ctx.profile_read(reset=True)
    print("kernel classes (ms per call): sweep %.3f plan %.3f" % (ms[mvs_amd.MVS_K_SWEEP] / 5, ms[mvs_amd.MVS_K_PLAN] / 5))
    t0 = time.perf_counter()
    for f in range(40, 70):
        ctx.sweep(cams[f], frames[f], np.stack([cams[j] for j in sides(f)]), [frames[j] for j in sides(f)], D)
    print("mvs_sweep per main frame %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
