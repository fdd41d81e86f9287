"""Histogram of the fixed sampler's region plan (diagnostic): MVS_PLAN_DUMP makes the library write the descriptors after planning.
usage: python tools/plan_hist.py [c1|c2|c3]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch  # noqa: F401 (HIP runtime first)

path = os.path.join(tempfile.gettempdir(), "mvs_plan.bin")
os.environ["MVS_TEST_HOOKS"] = "1"
os.environ["MVS_PLAN_DUMP"] = path
import mvs_amd
from mvs_amd import synth

cfg = {"c1": (640, 480, 32, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16)}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
W, H, D, V = cfg
mc, mi, sc, si, gt = synth.make_views(W, H, V, radius=0.15)
with mvs_amd.Context(W, H, sampler="fixed") as ctx:
    ctx.sweep_set(mc, mi, sc, si, D)
    ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
    ctx.synchronize()
raw = np.fromfile(path, dtype=np.uint32)
tx, ty, nch, nv = raw[:4]
d = raw[4:].reshape(-1, 2)
rw, rh, mode = d[:, 1] & 0xff, (d[:, 1] >> 8) & 0xff, (d[:, 1] >> 16) & 7
names = ["SKIP", "FAST", "BORDER", "GENERIC"]
print("tiles %dx%d chunks %d views %d regions %d" % (tx, ty, nch, nv, len(d)))
for m in range(4):
    sel = mode == m
    if sel.any():
        print("%-8s %6.2f %%   rw mean %.1f max %d   rh mean %.1f max %d   quads/region %.0f" %
              (names[m], 100 * sel.mean(), rw[sel].mean(), rw[sel].max(), rh[sel].mean(), rh[sel].max(), (rw[sel].astype(int) * rh[sel]).mean()))
st = (mode == 1) | (mode == 2)
print("rh histogram (staged):", np.bincount(rh[st])[:33])
print("rw histogram /16 (staged):", np.bincount(rw[st] // 16))
pairs = rh[st][:-1].astype(int) + rh[st][1:]
print("consecutive staged regions fitting the 32-row ring together: %.1f %%" % (100 * (pairs <= 32).mean()))
