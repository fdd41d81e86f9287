import sys, ctypes, numpy as np, time, os
sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch
import meshing_oracle as mo, meshing_common as mc
hip=ctypes.CDLL('mesh-reconstruction_amd/lib/libmvs_hip.so')
rng=np.random.default_rng(3)
n=4000
u=rng.normal(size=(n,3)); u/=np.linalg.norm(u,axis=1,keepdims=True)
pts=np.hstack([u*0.8+np.array([0.3,-0.2,1.0]), np.ones((n,1))]).astype(np.float32)
nrm=u.astype(np.float32)
for lg in (5,6,0):
    t0=time.time(); r=mc.poisson(hip, pts, nrm, lg, 1.0); t1=time.time()
    G,origin,h=mo.poisson_grid(pts,lg)
    print("lg",lg,"G",r['G'],G,"origin",r['origin'],origin,"h",r['h'],h,"time %.3f"%(t1-t0))
    sp=mo.poisson_splat(pts,nrm,G,origin,h)
    print(" splat equal:", np.array_equal(sp, r['splat']), "nonzero", np.count_nonzero(sp[3]))
    chi=mo.poisson_chi(sp,1.0)
    rngc=chi.max()-chi.min()
    print(" chi max abs diff / range: %.2e"%(np.abs(chi-r['chi']).max()/rngc), "iso", r['iso'], mo.trilinear(chi,G,origin,h,pts[:,:3]/pts[:,3:4]).mean())
    v,f=mo.surface_nets(r['chi'], r['iso'], origin, h)
    print(" verts", len(v), len(r['vertices']), "faces", len(f), len(r['faces']), "faces equal", np.array_equal(f,r['faces']), "verts maxdiff", np.abs(v-r['vertices']).max() if len(v)==len(r['vertices']) else None)
    rad=np.linalg.norm(r['vertices'][:,:3]-np.array([0.3,-0.2,1.0]),axis=1)
    print(" radius mean %.4f min %.4f max %.4f (cells: %.3f)"%(rad.mean(),rad.min(),rad.max(),(rad.max()-rad.min())/r['h']), "vol %.4f vs %.4f"%(mc.signed_volume(r['vertices'],r['faces']), 4/3*np.pi*0.8**3))
    e=mc.edge_use(r['faces']); print(" closed:", all(e[(b,a)]==c for (a,b),c in e.items()))
