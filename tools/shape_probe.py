import json, os, sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mesh-reconstruction_amd", "python"))
import numpy as np, mvs_amd
from mvs_amd import synth
def t(ctx, V, flags, n=10):
    for _ in range(3): ctx.sweep_run(0, V, flags)
    ctx.synchronize(); ts=[]
    for _ in range(3):
        t0=time.perf_counter()
        for _ in range(n): ctx.sweep_run(0, V, flags)
        ctx.synchronize(); ts.append((time.perf_counter()-t0)/n*1e3)
    return min(ts)
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
for name,(W,H,D,V) in {"c1":(640,480,32,4),"c2":(1280,720,64,8),"c3":(1920,1080,128,16)}.items():
    mc, mi, sc, si, _ = synth.make_views(W,H,V,radius=0.15)
    with mvs_amd.Context(W,H) as ctx:
        ctx.sweep_set(mc, mi, sc, si, D)
        a = t(ctx, V, both); shape = ctx.plan_shape()
        b = t(ctx, V, both | (8<<8))
        print(json.dumps({"config":name,"auto_ms":a,"auto_shape":shape,"forced_4x16_ms":b}))
