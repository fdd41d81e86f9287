"""Randomised hunt for disagreements between the fixed sampler's tiled kernel, its un-tiled kernel and the oracle: random sizes, plane
counts, view counts (sequences mixing SKIP / FAST / BORDER / GENERIC / wide regions), forced plane splits.  Not part of the test
suite proper (minutes of GPU time; it lives under tests/ because it uses the oracle): python tests/perf/stress_fx.py [first_seed] [count] [fixed|exact]"""
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
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sampler = sys.argv[3] if len(sys.argv) > 3 else "fixed"
both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN


def random_camera(rng, W, H, spread, angle):
    c = rng.uniform(-spread, spread, 3) * np.array([1.0, 1.0, 0.3])
    yaw, pitch = rng.uniform(-angle, angle, 2)
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    R = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    return synth.camera_at(c, W, H, rot=R, fovx=rng.uniform(0.6, 1.4))


bad = 0
seen = []
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(2, 700)), int(rng.integers(2, 260))
    D, V = int(rng.integers(1, 70)), int(rng.integers(1, 14))
    if seed % 3 == 0:   # every camera with the same orientation (translations in all three directions, own intrinsics): the separable path of sweep_fx_tiled
        main_cam = random_camera(rng, W, H, 0.05, 0.0)
        side_cams = np.stack([random_camera(rng, W, H, rng.choice([0.1, 0.4, 1.5]), 0.0) if rng.uniform() < 0.85 else random_camera(rng, W, H, 0.3, 0.2) for _ in range(V)])
    else:
        main_cam = random_camera(rng, W, H, 0.05, 0.05)
        side_cams = np.stack([random_camera(rng, W, H, rng.choice([0.1, 0.4, 1.5]), rng.choice([0.02, 0.3, 0.7])) for _ in range(V)])
    main_img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    sides = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(V)]
    z = (float(rng.uniform(-1.0, -0.2)), float(rng.uniform(0.2, 1.0)))
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, z[0], z[1], want_volume=True, nthreads=8, sampler=sampler)
    seen.append(float(((ref[3] >> (24 if sampler == "fixed" else 16)) > 0).mean()))
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D, z[0], z[1])
        for flags in (both, both | (1 << 16), both | (8 << 8), both | mvs_amd.MVS_SWEEP_FORCE_GENERIC):
            ctx.sweep_run(0, V, flags)
            depth, cost, idx, vol = ctx.sweep_fetch(want_volume=True)
            nv, ni = int(np.count_nonzero(vol != ref[3])), int(np.count_nonzero(idx != ref[2]))
            if nv > max(1, vol.size * 1e-6) or ni > max(1, idx.size * 1e-5):
                bad += 1
                print("seed %d %dx%d D=%d V=%d flags=%#x: %d cells, %d indices differ" % (seed, W, H, D, V, flags, nv, ni), flush=True)
print("stress (%s sampler): seeds %d..%d, %d disagreements; cells with a view in frame: mean %.2f, cases above 0.5: %d, all empty: %d" %
      (sampler, first, first + count - 1, bad, float(np.mean(seen)), int(np.sum(np.array(seen) > 0.5)), int(np.sum(np.array(seen) == 0.0))))
