"""Random sequences of sweep calls on ONE long-lived context -- new view sets (ring cameras, rotated cameras, mixtures, subsets), new planes, sampler
switches, MVS_SWEEP_NO_RECT, the one-call entry, frame-store handles, device-resident setters -- each result compared with what a FRESH context gives
for the same inputs: whatever state a context keeps between calls (region plans and their reuse snapshot, rectified-view tables, lazily built padded /
f16 images, the frame store) must never leak into a result.  Not part of the suite proper (tests/test_handles_gpu.py runs a short version):
python tests/perf/fuzz_api.py [first_seed] [count] [steps per seed]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np
import torch  # noqa: F401 (HIP runtime first)
import mvs_amd
from mvs_amd import synth

both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def run(seed, steps, W=320, H=200, verbose=False):
    rng = np.random.default_rng(seed)
    fov = 0.9
    main_cam = synth.camera_at([0.0, 0.0, 0.0], W, H, fovx=fov)
    pool = []          # (camera, frame, rectified?)
    for k in range(10):
        ang = 2 * np.pi * k / 10
        pos = [0.12 * np.cos(ang), 0.12 * np.sin(ang), 0.0]
        cam = synth.camera_at(pos, W, H, fovx=fov)
        if k % 3 == 2:   # a rotated (general) camera
            c, s = np.cos(0.012), np.sin(0.012)
            R = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], np.float32)
            cam = (cam @ R).astype(np.float32)
        pool.append((cam, rng.integers(0, 256, (H, W), dtype=np.uint8)))
    mains = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(3)]
    plane_sets = [(16, -0.4, 0.6), (32, -0.4, 0.6), (48, -0.8, 0.9), (32, 0.1, 0.7)]
    bad = 0
    with mvs_amd.Context(W, H) as ctx:
        ctx.frame_store(len(pool) + len(mains))
        for i, (_, fr) in enumerate(pool):
            ctx.frame_upload(i, fr)
        for i, m in enumerate(mains):
            ctx.frame_upload(len(pool) + i, m)
        for step in range(steps):
            op = rng.choice(["set", "set", "views", "planes", "onecall", "handles", "sampler", "device"])
            nv = int(rng.integers(1, 6))
            ids = list(rng.choice(len(pool), nv, replace=False))
            if rng.random() < 0.5:
                ids = [i for i in ids if i % 3 != 2] or [0]          # every view rectified: the rectified-view kernels
            mi = int(rng.integers(0, len(mains)))
            D, zlo, zhi = plane_sets[int(rng.integers(0, len(plane_sets)))]
            sampler = ["fixed", "exact"][int(rng.integers(0, 2))] if op == "sampler" else None
            flags = both | (mvs_amd.MVS_SWEEP_NO_RECT if rng.random() < 0.2 else 0)
            cams = np.stack([pool[i][0] for i in ids])
            frames = [pool[i][1] for i in ids]
            # ---- the long-lived context
            if sampler:
                ctx.set_sampler(sampler)
                cur_sampler = sampler
            else:
                cur_sampler = getattr(ctx, "_fuzz_sampler", "fixed")
            ctx._fuzz_sampler = cur_sampler
            if op == "handles" and cur_sampler != "fixed":
                op = "set"
            if op in ("set", "sampler"):
                ctx.sweep_set(main_cam, mains[mi], cams, frames, D, zlo, zhi)
                ctx.sweep_run(0, len(ids), flags)
                got = ctx.sweep_fetch(want_volume=True)
                state = (mi, ids, (D, zlo, zhi))
            elif op == "views" and getattr(ctx, "_fuzz_state", None):
                mi, _, (D, zlo, zhi) = ctx._fuzz_state
                ctx.sweep_set_views(cams, frames)
                ctx.sweep_run(0, len(ids), flags)
                got = ctx.sweep_fetch(want_volume=True)
                state = (mi, ids, (D, zlo, zhi))
            elif op == "planes" and getattr(ctx, "_fuzz_state", None):
                mi, ids, _ = ctx._fuzz_state
                cams = np.stack([pool[i][0] for i in ids]); frames = [pool[i][1] for i in ids]
                ctx.sweep_set_planes(D, zlo, zhi)
                ctx.sweep_run(0, len(ids), flags)
                got = ctx.sweep_fetch(want_volume=True)
                state = (mi, ids, (D, zlo, zhi))
            elif op == "onecall":
                d, c = ctx.sweep(main_cam, mains[mi], cams, frames, D, zlo, zhi, want_cost=True)
                got = (d, c, None, None)
                state = None
            elif op == "handles":
                d, c = ctx.sweep_handles(len(pool) + mi, main_cam, ids, cams, D, zlo, zhi, want_cost=True)
                got = (d, c, None, None)
                state = None
            elif op == "device":
                t_main = torch.from_numpy(mains[mi]).cuda()
                t_side = [torch.from_numpy(f).cuda() for f in frames]
                ctx.sweep_set_planes(D, zlo, zhi)
                ctx.sweep_set_main_device(main_cam, t_main.data_ptr())
                ctx.sweep_set_views_device(cams, [t.data_ptr() for t in t_side])
                ctx.sweep_run(0, len(ids), flags)
                got = ctx.sweep_fetch(want_volume=True)
                torch.cuda.synchronize()
                state = (mi, ids, (D, zlo, zhi))
            else:
                continue
            ctx._fuzz_state = state
            # ---- a fresh context, the plain way
            with mvs_amd.Context(W, H, sampler=cur_sampler) as ref:
                ref.sweep_set(main_cam, mains[mi], cams, frames, D, zlo, zhi)
                ref.sweep_run(0, len(ids), both)
                want = ref.sweep_fetch(want_volume=True)
            same = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)
            if got[2] is not None:
                same = same and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
            if not same or verbose:
                print("seed %d step %d %s sampler %s views %s main %d planes %s flags %#x: %s" % (seed, step, op, cur_sampler, ids, mi, (D, zlo, zhi), flags, "ok" if same else "DIFFERS"), flush=True)
            bad += not same
    return bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    total = sum(run(s, steps) for s in range(first, first + count))
    print("api fuzz: seeds %d..%d x %d steps, %d results differ from a fresh context's" % (first, first + count - 1, steps, total))
