import sys, time
sys.path.insert(0, 'mesh-reconstruction_amd/python')
import numpy as np, mvs_amd
from mvs_amd import synth
W,H,D,V=1920,1080,128,16
mc,mi,sc,si,gt=synth.make_views(W,H,V)
ctx=mvs_amd.Context(W,H); ctx.sweep_set(mc,mi,sc,si,D)
ctx.profile_enable(True)
for flags,name in [(1,'grouped'),(1|0x200,'linear'),(1,'grouped'),(1|0x200,'linear'),(1|0x100,'nostage grouped'),(1|0x300,'nostage linear'),(1,'grouped'),(1|0x200,'linear')]:
    for _ in range(3): ctx.sweep_run(0,V,flags)
    ctx.profile_read(True)
    for _ in range(10): ctx.sweep_run(0,V,flags)
    ms,n=ctx.profile_read(True)
    print(name, ms[0]/n[0])
