"""The general kernel's separable path at BASELINE c3: side cameras with the main camera's orientation, on the benchmark's ring but moved along
the optical axis by up to +-0.1 (not sweep_fx_rect's case), with the path on and off (MVS_NO_SEP=1).  python tools/time_sep.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mesh-reconstruction_amd", "python"))
os.environ["MVS_TEST_HOOKS"] = "1"
import numpy as np

import mvs_amd
from mvs_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
out = {}
for name, (W, H, D, V) in (("c3", (1920, 1080, 128, 16)), ("c2", (1280, 720, 64, 8)), ("c5 shape", (640, 480, 128, 4))):
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V)
    cams = side_cams.copy()
    for v in range(V):
        a = 2.0 * np.pi * v / V
        cams[v] = synth.camera_at([0.15 * np.cos(a), 0.15 * np.sin(a), 0.1 * np.sin(1.7 * v + 0.3)], W, H)
    ref = None
    for label, env in (("separable", None), ("general form", "1")):
        if env:
            os.environ["MVS_NO_SEP"] = env
        else:
            os.environ.pop("MVS_NO_SEP", None)
        with mvs_amd.Context(W, H, sampler="fixed") as ctx:
            ctx.sweep_set(main_cam, main_img, cams, sides, D)
            for _ in range(5):
                ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
            d, c, i, _ = ctx.sweep_fetch()
            if ref is None:
                ref = (d.copy(), c.copy(), i.copy())
            assert np.array_equal(d, ref[0]) and np.array_equal(c, ref[1]) and np.array_equal(i, ref[2])
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
            ctx.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / reps
            out["%s %s" % (name, label)] = {"ms_per_step": ms, "samples_per_s": float(W) * H * D * V / (ms * 1e-3), "plan_shape": ctx.plan_shape() if hasattr(ctx, "plan_shape") else None}
print(json.dumps(out, indent=1))
