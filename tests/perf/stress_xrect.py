"""Randomised hunt for disagreements between the rectified-view kernel (sweep_exact_rect), sweep_tiled and the oracle (the exact sampler): random
sizes (ragged tiles), plane counts, view counts, depth ranges and in-plane camera shifts from a fraction of a pixel to most of the frame
(planes fully in frame, partly in frame, out of frame and with rows that do not advance in step, in every mixture), view subsets, forced plane splits.  Not part of the suite proper:
python tests/perf/stress_xrect.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401 (HIP runtime first)
import mvs_amd
from mvs_amd import synth
import orc

oracle = orc.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
NR = mvs_amd.MVS_SWEEP_NO_RECT

bad = rect_cases = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(8, 900)), int(rng.integers(8, 300))
    D, V = int(rng.integers(1, 80)), int(rng.integers(1, 12))
    fov = float(rng.uniform(0.6, 1.4))
    main_cam = synth.camera_at([0.0, 0.0, 0.0], W, H, fovx=fov)
    spread = float(rng.choice([0.002, 0.05, 0.3, 1.0]))
    side_cams = np.stack([synth.camera_at([rng.uniform(-spread, spread), rng.uniform(-spread, spread), 0.0], W, H, fovx=fov) for _ in range(V)])
    main_img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    sides = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(V)]
    z = (float(rng.uniform(-1.0, -0.2)), float(rng.uniform(0.2, 1.0)))
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, z[0], z[1], want_volume=True, nthreads=8, sampler="exact")
    with mvs_amd.Context(W, H, sampler="exact") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D, z[0], z[1])
        ctx.sweep_run(0, V, both)
        rect_cases += ctx.plan_shape() == 5
        for flags in (both, both | (1 << 16), both | (3 << 16), both | NR):
            ctx.sweep_run(0, V, flags)
            depth, cost, idx, vol = ctx.sweep_fetch(want_volume=True)
            nv, ni = int(np.count_nonzero(vol != ref[3])), int(np.count_nonzero(idx != ref[2]))
            if nv or ni:
                bad += 1
                print("seed %d %dx%d D=%d V=%d spread %g flags=%#x plan %d: %d cells, %d indices differ" % (seed, W, H, D, V, spread, flags, ctx.plan_shape(), nv, ni), flush=True)
        if V > 1:  # a view subset: linearity of the packed cells
            v0 = int(rng.integers(0, V - 1))
            vn = int(rng.integers(1, V - v0 + 1))
            ctx.sweep_run(v0, vn, mvs_amd.MVS_SWEEP_VOLUME)
            a = ctx.sweep_fetch(want_volume=True)[3]
            ctx.sweep_run(v0, vn, mvs_amd.MVS_SWEEP_VOLUME | NR)
            b = ctx.sweep_fetch(want_volume=True)[3]
            if not np.array_equal(a, b):
                bad += 1
                print("seed %d: view subset [%d, %d) differs between the kernels" % (seed, v0, v0 + vn), flush=True)
print("stress (rectified views): seeds %d..%d, %d disagreements; %d of %d cases served by sweep_exact_rect" % (first, first + count - 1, bad, rect_cases, count))
