"""PCIe-inclusive rate of the one-call host-buffer form mvs_sweep() at c3 (uploads 17 frames, downloads depth)."""
import sys, time
sys.path.insert(0, 'mesh-reconstruction_amd/python')
import numpy as np, mvs_amd
from mvs_amd import synth
W, H, D, V = 1920, 1080, 128, 16
mc, mi, sc, si = synth.noise_views(W, H, V)
with mvs_amd.Context(W, H) as ctx:
    ctx.sweep(mc, mi, sc, si, D)
    t0 = time.perf_counter()
    for _ in range(5): ctx.sweep(mc, mi, sc, si, D)
    dt = (time.perf_counter() - t0) / 5
print("one-call mvs_sweep (host buffers, pageable): %.2f ms -> %.3g samples/s" % (dt * 1e3, W * H * D * V / dt))
