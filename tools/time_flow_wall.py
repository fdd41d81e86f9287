import sys, time
sys.path.insert(0, 'mesh-reconstruction_amd/python')
import numpy as np, mvs_amd
from mvs_amd import synth
W, H = int(sys.argv[1]), int(sys.argv[2])
sc = synth.Scene(freq_scale=W / 1920.0)
a, b = sc.render([0, 0, 0], W, H), sc.render([0.05, 0, 0], W, H)
with mvs_amd.Context(W, H) as ctx:
    for _ in range(5): ctx.flow(a, b, True)
    ts=[]
    for _ in range(30):
        t0=time.perf_counter(); ctx.flow(a, b, True); ts.append((time.perf_counter()-t0)*1e3)
    ts.sort(); print(W,H,"wall median %.2f min %.2f max %.2f"%(ts[15],ts[0],ts[-1]))
