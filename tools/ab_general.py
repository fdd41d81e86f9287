import json, os, sys, time
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # master switch of the library's environment hooks (csrc/hooks.hpp)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")  # the library masks the experiment bits of `flags` otherwise
sys.path.insert(0, os.path.join(os.getcwd(), "mesh-reconstruction_amd", "python"))
import numpy as np, mvs_amd
from mvs_amd import synth
def t(ctx, V, flags, n=15):
    for _ in range(3): ctx.sweep_run(0, V, flags)
    ctx.synchronize(); ts=[]
    for _ in range(4):
        t0=time.perf_counter()
        for _ in range(n): ctx.sweep_run(0, V, flags)
        ctx.synchronize(); ts.append((time.perf_counter()-t0)/n*1e3)
    return min(ts)
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
W,H,D,V = 1920,1080,128,16
mc, mi, sc, si, _ = synth.make_views(W,H,V,radius=0.15)
with mvs_amd.Context(W,H) as ctx:
    ctx.sweep_set(mc, mi, sc, si, D)
    # high-byte debug bits: 4 = never take the plane-independent-w path; 16 = general path with the compiler-scheduled pair
    # loop instead of the hand-pipelined reads (the shipped default in the 2 x 32 shape)
    r = {"wconst": t(ctx,V,both), "general_pipelined": t(ctx,V,both|(4<<8)), "general_compiler_loop": t(ctx,V,both|((4|16)<<8)),
         "general_pipelined_again": t(ctx,V,both|(4<<8))}
    ctx.sweep_run(0,V,both|(4<<8)); a=[x.copy() for x in ctx.sweep_fetch()[:3]]
    ctx.sweep_run(0,V,both|((4|16)<<8)); b=ctx.sweep_fetch()[:3]
    r["identical"]=all(np.array_equal(x,y) for x,y in zip(a,b))
print(json.dumps(r))
