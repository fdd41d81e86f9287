"""One million noisy samples of the unit sphere through mvs_poisson_surface at 512^3 (grid_log2 = 9) and at the automatic grid: wall time per call
(first call of a size includes rocFFT's runtime compilation), radius range, volume.  python tests/perf/poisson_large.py (on the GPU box)"""
import sys,time,ctypes,numpy as np
sys.path.insert(0,"tests"); import torch, meshing_common as mc
hip=ctypes.CDLL("mesh-reconstruction_amd/lib/libmvs_hip.so")
rng=np.random.default_rng(0); n=1000000
u=rng.normal(size=(n,3)); u/=np.linalg.norm(u,axis=1,keepdims=True)
pts=np.hstack([u*(1+0.002*rng.normal(size=(n,1))),np.ones((n,1))]).astype(np.float32); nrm=u.astype(np.float32)
for lg in (9,9,0,0):
    t=time.time(); r=mc.poisson(hip,pts,nrm,lg,1.0,keep=False); dt=time.time()-t
    v=r["vertices"]; rad=np.linalg.norm(v[:,:3],axis=1)
    print("G", r["G"], "%.3f s"%dt, len(v), len(r["faces"]), "radius %.4f .. %.4f (cell %.4f)"%(rad.min(),rad.max(),r["h"]), "vol %.4f"%mc.signed_volume(v,r["faces"]))
