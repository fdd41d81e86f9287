"""The same idea as fuzz_api.py for the per-frame pipeline: random sequences of loadMesh (two meshes), depth, projected, calculateFlow (both
algorithms), mvs_process_frame and mvs_process_frame_slots (1..4 side views, both flows; the store filled lazily and emptied now and then), filterPoints and a small sweep on ONE long-lived context, every output compared
with a fresh context's -- the stages share scratch arenas (flow arena = triangulation and filter scratch, the raster's temporaries, the lanes), and
none of that may show in a result.  python tests/perf/fuzz_pipeline.py [first_seed] [count] [steps]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401 (HIP runtime first)
import mvs_amd
import c5_common
import scenes


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def run(seed, steps, verbose=False):
    rng = np.random.default_rng(seed)
    seq = c5_common.Sequence()
    W, H = seq.W, seq.H
    meshes = [(seq.verts, seq.faces), scenes.proxy_plane(seq.bundles, seq.cams[10], n=20, scale=0.9)]
    bad = 0
    stored = {"sized": False, "have": set()}   # what the long-lived context's frame store holds
    ref_marker = [None]                        # the fresh context of the current step

    def do(ctx, op, a):
        if op == "depth":
            return crc(ctx.depth(seq.cams[a["f"]]))
        if op == "projected":
            return crc(ctx.projected(seq.cams[a["f"]], seq.frame(a["g"]), seq.cams[a["g"]]))
        if op == "flow":
            return crc(ctx.flow(seq.frame(a["f"]), seq.frame(a["g"]), a["fb"]))
        if op == "process":
            ids = seq.sides(a["f"])[:a["n"]]
            return crc(ctx.process_frame(seq.cams[a["f"]], seq.frame(a["f"]), np.stack([seq.cams[j] for j in ids]), [seq.frame(j) for j in ids], a["fb"]))
        if op == "process_slots":   # the long-lived context takes the frames from its frame store (filled lazily, emptied now and then); the fresh one from the host
            ids = seq.sides(a["f"])[:a["n"]]
            if ctx is not ref_marker[0]:
                if not stored["sized"] or a["reset"]:
                    ctx.frame_store(seq.n)
                    stored["sized"] = True
                    stored["have"] = set()
                for j in [a["f"]] + ids:
                    if j not in stored["have"]:
                        ctx.frame_upload(j, seq.frame(j))
                        stored["have"].add(j)
                return crc(ctx.process_frame_slots(seq.cams[a["f"]], a["f"], np.stack([seq.cams[j] for j in ids]), ids, a["fb"]))
            return crc(ctx.process_frame(seq.cams[a["f"]], seq.frame(a["f"]), np.stack([seq.cams[j] for j in ids]), [seq.frame(j) for j in ids], a["fb"]))
        if op == "filter":
            pts = np.random.default_rng(a["s"]).normal(size=(a["n"], 4)).astype(np.float32)
            pts[:, 3] = 1.0
            return crc(ctx.filter_points(pts, a["alpha"]))
        if op == "sweep":
            ids = seq.sides(a["f"])[:a["n"]]
            return crc(*ctx.sweep(seq.cams[a["f"]], seq.frame(a["f"]), np.stack([seq.cams[j] for j in ids]), [seq.frame(j) for j in ids], 16, want_cost=True))
        raise ValueError(op)

    with mvs_amd.Context(W, H) as ctx:
        cur = 0
        ctx.load_mesh(*meshes[cur])
        for step in range(steps):
            op = str(rng.choice(["mesh", "depth", "projected", "flow", "process", "process", "process_slots", "process_slots", "filter", "sweep", "poisson"]))
            if op == "poisson":   # context-free, but it is what made a previously captured hipGraph replay wrongly: keep it in the mix
                d = np.random.default_rng(int(rng.integers(0, 1 << 30))).normal(size=(1500, 3))
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                mvs_amd.poisson_surface(np.concatenate([d, np.ones((1500, 1))], 1).astype(np.float32), d.astype(np.float32), grid_log2=int(rng.choice([4, 5])), criteria=None)
                continue
            if op == "mesh":
                cur = int(rng.integers(0, 2))
                ctx.load_mesh(*meshes[cur])
                continue
            a = {"f": int(rng.choice(seq.mains[3:-3])), "g": int(rng.integers(0, seq.n)), "fb": bool(rng.integers(0, 2)), "n": int(rng.integers(1, 5)),
                 "s": int(rng.integers(0, 1 << 30)), "alpha": float(rng.choice([0.02, 0.1, 0.3])), "reset": bool(rng.integers(0, 6) == 0)}
            if op == "filter":
                a["n"] = int(rng.choice([500, 5000, 40000]))
            got = do(ctx, op, a)
            with mvs_amd.Context(W, H) as ref:
                ref.load_mesh(*meshes[cur])
                ref_marker[0] = ref
                want = do(ref, op, a)
                ref_marker[0] = None
            if got != want or verbose:
                print("seed %d step %d %s %s mesh %d: %s" % (seed, step, op, a, cur, "ok" if got == want else "DIFFERS"), flush=True)
            bad += got != want
    return bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    total = sum(run(s, steps) for s in range(first, first + count))
    print("pipeline fuzz: seeds %d..%d x %d steps, %d results differ from a fresh context's" % (first, first + count - 1, steps, total))
