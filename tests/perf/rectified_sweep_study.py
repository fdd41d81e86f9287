"""Study (round 5, CPU only: the oracle's two samplers): would RECTIFY-THEN-SWEEP serve rotated cameras?  A rotation about the camera centre is a
homography of its image, so a side view can be resampled ONCE per view set into a virtual camera with the main camera's orientation and
intrinsics; if its centre also lies in the main camera's focal plane the view is then "rectified" in sweep_rect.hip's sense (0.61 ms at c3
instead of the general kernel's 1.67).  This script measures what the extra interpolation costs in fidelity: side cameras on the benchmark's
ring, each turned by up to `theta` rad about two axes, frames ray-cast from the turned cameras; (a) contract v2 on the real views, (b) contract
v2 on the virtual views (bilinear resample in f64, rounded to u8), each against the exact sampler on the real views and the analytic depth.
Evaluated `margin` pixels inside the frame so that no sample of the virtual views falls into the wedge the rotation leaves without data (a
product version needs the validity polygon in planner, kernel and oracle).
    python tests/perf/rectified_sweep_study.py [theta = 0.012] [margin = 170]
Measured at 960 x 540, 64 planes, 8 views: theta 0.012 / 0.03 -- plane flips against the exact sampler 2.6 % / 2.3 % (v2 on the
real views: 2.1 % / 2.0 %), by more than one plane 0 / 0.003 %, mean best-cost difference 0.11 grey levels (0.09), depth RMSE against the analytic
surface 0.00928 / 0.00931 (exact sampler 0.00930 / 0.00932): on par.  NOT built: it serves centres in the focal plane only; the tracks the reference
bundles move along the optical axis too (q10 != 0: the sample phase varies across a tile, which is what sweep_rect.hip's fast path cannot have)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import orc
from mvs_amd import synth
oracle = orc.load()
W, H, D, V = 960, 540, 64, 8
theta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.012
sc = synth.Scene(synth.SEED_SCENE, W / 1920.0)
def rot(yaw, pitch):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    return np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
def render_rot(c, R):
    aspect = W / H
    col = (np.arange(W) * 2 + 1) / W - 1.0
    row = 1.0 - (np.arange(H) * 2 + 1) / H
    dxc = (col * synth.FOVX / 2.0)[None, :].repeat(H, 0); dyc = (row * synth.FOVX / (2.0 * aspect))[:, None].repeat(W, 1)
    dc = np.stack([dxc, dyc, -np.ones_like(dxc)], -1)          # camera-space ray
    dw = dc @ R                                                # world = R^T d  (row vectors: d @ R)
    t = np.full((H, W), 3.0)
    for _ in range(20):
        t = (sc.height(c[0] + t * dw[..., 0], c[1] + t * dw[..., 1]) - c[2]) / dw[..., 2]
    X, Y = c[0] + t * dw[..., 0], c[1] + t * dw[..., 1]
    return np.clip(np.rint(sc.albedo(X, Y)), 0, 255).astype(np.uint8)
main_cam = synth.camera_at([0, 0, 0], W, H)
main_img, gt = sc.render([0, 0, 0], W, H, want_depth=True)
rng = np.random.default_rng(1)
centers, Rs = [], []
for v in range(V):
    a = 2 * np.pi * v / V
    centers.append(np.array([0.15 * np.cos(a), 0.15 * np.sin(a), 0.0]))
    Rs.append(rot(rng.uniform(-theta, theta), rng.uniform(-theta, theta)))
real_cams = np.stack([synth.camera_at(c, W, H, rot=R) for c, R in zip(centers, Rs)])
real_imgs = [render_rot(c, R) for c, R in zip(centers, Rs)]
virt_cams = np.stack([synth.camera_at(c, W, H) for c in centers])
def pix(cam, X):   # world points (N,3) -> pixel coords (x, y) top-down, centre of pixel (0,0) = (0,0)
    s = np.concatenate([X, np.ones((len(X), 1))], 1) @ cam.astype(np.float64).T
    return (s[:, 0] / s[:, 3]) * W / 2 + W / 2 - 0.5, -(s[:, 1] / s[:, 3]) * H / 2 + H / 2 - 0.5
virt_imgs, valid = [], []
yy, xx = np.mgrid[0:H, 0:W]
for v in range(V):
    # a world point on the ray of each virtual pixel: unproject ndc at z = 0
    ndc = np.stack([(2 * xx.ravel() + 1) / W - 1, 1 - (2 * yy.ravel() + 1) / H, np.zeros(W * H), np.ones(W * H)], 1)
    Xh = ndc @ np.linalg.inv(virt_cams[v].astype(np.float64)).T
    X = Xh[:, :3] / Xh[:, 3:4]
    px, py = pix(real_cams[v], X)
    x0, y0 = np.floor(px).astype(int), np.floor(py).astype(int)
    fx, fy = px - x0, py - y0
    ok = (x0 >= 0) & (y0 >= 0) & (x0 + 1 < W) & (y0 + 1 < H)
    x0c, y0c = np.clip(x0, 0, W - 2), np.clip(y0, 0, H - 2)
    im = real_imgs[v].astype(np.float64)
    val = (im[y0c, x0c] * (1 - fx) + im[y0c, x0c + 1] * fx) * (1 - fy) + (im[y0c + 1, x0c] * (1 - fx) + im[y0c + 1, x0c + 1] * fx) * fy
    virt_imgs.append(np.where(ok, np.clip(np.rint(val), 0, 255), 0).astype(np.uint8).reshape(H, W))
    valid.append(ok.reshape(H, W))
nt = os.cpu_count() or 8
t0 = time.time()
d_e, c_e, i_e, _ = oracle.sweep(main_cam, main_img, real_cams, real_imgs, D, nthreads=nt, sampler="exact")
d_f, c_f, i_f, _ = oracle.sweep(main_cam, main_img, real_cams, real_imgs, D, nthreads=nt, sampler="fixed")
d_r, c_r, i_r, _ = oracle.sweep(main_cam, main_img, virt_cams, virt_imgs, D, nthreads=nt, sampler="fixed")
d_x, c_x, i_x, _ = oracle.sweep(main_cam, main_img, virt_cams, virt_imgs, D, nthreads=nt, sampler="exact")
print("sweeps %.1f s" % (time.time() - t0))
m = int(sys.argv[2]) if len(sys.argv) > 2 else 170
inner = np.zeros((H, W), bool); inner[m:-m, m:-m] = True
ok = inner & (i_e >= 0) & (i_f >= 0) & (i_r >= 0)
n = ok.sum()
for name, i, d, c in (("v2 fixed on real views", i_f, d_f, c_f), ("v3 rectified + fixed", i_r, d_r, c_r), ("rectified + exact", i_x, d_x, c_x)):
    flip = (ok & (i != i_e)).sum() / n; far = (ok & (np.abs(i - i_e) > 1)).sum() / n
    rm = np.sqrt(np.mean((d[ok].astype(np.float64) - gt[ok]) ** 2)); rme = np.sqrt(np.mean((d_e[ok].astype(np.float64) - gt[ok]) ** 2))
    print("%-24s flips vs exact %.4f  >1 plane %.5f  mean|dcost| %.4f  rmse gt %.6f (exact %.6f)" % (name, flip, far, np.mean(np.abs(c[ok] - c_e[ok])), rm, rme))
