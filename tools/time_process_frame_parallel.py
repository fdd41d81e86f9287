"""mvs_process_frame over a SEQUENCE: the main frames of recon.cpp's `fa` loop are independent of each other, and one 640 x 480 main frame keeps a
fraction of an MI355X busy (a chain of small launches: DESIGN.md sections 4, 6) -- so N host threads, each with a context of its own on the same GPU, each
taking every N-th main frame.  Prints ms per main frame (wall, all threads) for N = 1, 2, 4, 8 and checks that every thread's points equal the single-context result.
python tools/time_process_frame_parallel.py [farneback]"""
import os, sys, threading, time, zlib
import torch  # noqa: F401 (HIP runtime first)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, c5_common, mvs_amd
fb = len(sys.argv) > 1 and sys.argv[1] == "farneback"
seq = c5_common.Sequence()
mains = seq.mains[2:26]                       # 24 main frames with their 4 neighbours each
inputs = {f: (seq.frame(f), np.stack([seq.cams[j] for j in seq.sides(f)]), [seq.frame(j) for j in seq.sides(f)]) for f in mains}
ref = {}
with mvs_amd.Context(seq.W, seq.H) as ctx:
    ctx.load_mesh(seq.verts, seq.faces)
    for f in mains:
        mf, cams, frames = inputs[f]
        ref[f] = zlib.crc32(np.ascontiguousarray(ctx.process_frame(seq.cams[f], mf, cams, frames, fb)[:, :4]).tobytes())
for n in (1, 2, 4, 8):
    ctxs = [mvs_amd.Context(seq.W, seq.H) for _ in range(n)]
    for c in ctxs:
        c.load_mesh(seq.verts, seq.faces)
    bad = []
    def work(k, reps):
        c = ctxs[k]
        for _ in range(reps):
            for f in mains[k::n]:
                mf, cams, frames = inputs[f]
                pts = c.process_frame(seq.cams[f], mf, cams, frames, fb)
                if zlib.crc32(np.ascontiguousarray(pts[:, :4]).tobytes()) != ref[f]:
                    bad.append(f)
    def run(reps):
        th = [threading.Thread(target=work, args=(k, reps)) for k in range(n)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return time.perf_counter() - t0
    run(1)
    dt = run(3)
    print("%s flow, %d context(s) on one GPU: %.2f ms per main frame (%d frames), results %s" % ("Farneback" if fb else "variational", n, dt / (3 * len(mains)) * 1e3, 3 * len(mains), "equal" if not bad else "DIFFER %s" % bad[:4]), flush=True)
    for c in ctxs:
        c.close()
