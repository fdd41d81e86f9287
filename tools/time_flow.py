"""mvs_flow (calculateFlow, flow.cpp:19-42) through the C ABI: wall and device time per call for both algorithms.
usage: python3 tools/time_flow.py [W H] [farneback|variational|both]   (default 640 480 both; run under rocprofv3 --kernel-trace --stats for per-kernel figures)"""
import sys, time
sys.path.insert(0, 'mesh-reconstruction_amd/python')
import numpy as np, mvs_amd
from mvs_amd import synth
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
which = sys.argv[3] if len(sys.argv) > 3 else "both"
sc = synth.Scene(freq_scale=W / 1920.0)
a, b = sc.render([0, 0, 0], W, H), sc.render([0.05, 0, 0], W, H)
with mvs_amd.Context(W, H) as ctx:
    for fb in (True, False):
        if which != "both" and (which == "farneback") != fb:
            continue
        for _ in range(3): ctx.flow(a, b, fb)
        ctx.profile_enable(True); ctx.profile_read(True)
        t0 = time.perf_counter()
        for _ in range(20): ctx.flow(a, b, fb)
        dt = (time.perf_counter() - t0) / 20 * 1e3
        ms, n = ctx.profile_read(True)
        print("%dx%d" % (W, H), "farneback" if fb else "variational", "wall %.2f ms, device %.2f ms" % (dt, ms[5] / n[5]))
