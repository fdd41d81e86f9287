"""Per-section cycle counts of the fixed-sampler sweep's view loop (s_memtime, wavefront 0 of every workgroup).  Needs a library built
with -DMVS_FX_EXPERIMENTS (make CXXFLAGS="... -DMVS_FX_EXPERIMENTS"); the production build ignores MVS_FX_PROF."""
import sys, os
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # master switch of the library's environment hooks (csrc/hooks.hpp)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")  # the library masks the experiment bits of `flags` otherwise
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import torch  # noqa: F401 (HIP runtime first)
os.environ["MVS_FX_PROF"] = "1"
import mvs_amd
from mvs_amd import synth

cfg = {"c1": (640, 480, 32, 4), "c2": (1280, 720, 64, 8), "c3": (1920, 1080, 128, 16), "c5": (640, 480, 128, 4)}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
W, H, D, V = cfg
mc, mi, sc, si, gt = synth.make_views(W, H, V, radius=0.15)
if "--ring" not in sys.argv:  # general cameras (turned by 12 mrad, as bench.py's general_camera_path): the ring itself is served by sweep_fx_rect
    import numpy as np
    cams = []
    for vi in range(V):
        a = 2.0 * np.pi * vi / V
        yaw, pitch = 0.012 * np.cos(a), 0.012 * np.sin(a)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        rot = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        cams.append(synth.camera_at([0.15 * np.cos(a), 0.15 * np.sin(a), 0.0], W, H, rot=rot))
    sc = np.stack(cams)
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN | mvs_amd.MVS_SWEEP_NO_RECT
with mvs_amd.Context(W, H, sampler="fixed") as ctx:
    ctx.sweep_set(mc, mi, sc, si, D)
    for flags in (both, both, both, both | (32 << 8), both | (33 << 8)):
        sys.stderr.write("flags %#x: " % flags)
        sys.stderr.flush()
        ctx.sweep_run(0, V, flags)
        ctx.synchronize()
