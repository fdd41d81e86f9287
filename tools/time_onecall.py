"""PCIe-inclusive rate of the one-call host-buffer form mvs_sweep() at c3 (uploads 17 frames, downloads depth), from pageable and
from page-locked (mvs_host_alloc) caller buffers."""
import sys, time
sys.path.insert(0, 'mesh-reconstruction_amd/python')
import numpy as np, mvs_amd
from mvs_amd import synth
W, H, D, V = 1920, 1080, 128, 16
mc, mi, sc, si = synth.noise_views(W, H, V)
for kind in ("pageable", "pinned"):
    with mvs_amd.Context(W, H) as ctx:
        if kind == "pinned":
            frames = [mvs_amd.pinned_array((H, W), np.uint8) for _ in range(V + 1)]
            for dst, src in zip(frames, [mi] + list(si)):
                dst[...] = src
            mi, si = frames[0], frames[1:]
            ctx._pinned_depth = mvs_amd.pinned_array((H, W), np.float32)
        ctx.sweep(mc, mi, sc, si, D)
        ctx.sweep(mc, mi, sc, si, D)
        t0 = time.perf_counter()
        for _ in range(10): ctx.sweep(mc, mi, sc, si, D)
        dt = (time.perf_counter() - t0) / 10
    print("one-call mvs_sweep (host buffers, %s): %.2f ms -> %.3g samples/s" % (kind, dt * 1e3, W * H * D * V / dt))
