import sys, time
sys.path.insert(0, 'mesh-reconstruction_amd/python')
import numpy as np, mvs_amd
from mvs_amd import synth
W, H = 640, 480
sc = synth.Scene(freq_scale=W / 1920.0)
a, b = sc.render([0, 0, 0], W, H), sc.render([0.05, 0, 0], W, H)
with mvs_amd.Context(W, H) as ctx:
    for fb in (True, False):
        for _ in range(3): ctx.flow(a, b, fb)
        ctx.profile_enable(True); ctx.profile_read(True)
        t0 = time.perf_counter()
        for _ in range(20): ctx.flow(a, b, fb)
        dt = (time.perf_counter() - t0) / 20 * 1e3
        ms, n = ctx.profile_read(True)
        print("farneback" if fb else "variational", "wall %.2f ms, device %.2f ms" % (dt, ms[5] / n[5]))
