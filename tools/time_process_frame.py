"""mvs_process_frame (recon.cpp:65-117 for one main frame: depth, 4 x (projected, mixBackground, calculateFlow), triangulatePixels) on the zatisi cameras at
640 x 480, both flow algorithms; and the same under MVS_SERIAL_FLOWS=1 / MVS_FB_LANES=1 when those are set in the environment.
python tools/time_process_frame.py"""
import os, sys, time
os.environ.setdefault("MVS_TEST_HOOKS", "1")   # the A/B variables above are environment hooks: read only under the master switch
import torch  # noqa: F401 (HIP runtime first)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, c5_common, mvs_amd
seq = c5_common.Sequence()
with mvs_amd.Context(seq.W, seq.H) as ctx:
    ctx.load_mesh(seq.verts, seq.faces)
    f = seq.mains[12]; ids = seq.sides(f)
    cams = np.stack([seq.cams[j] for j in ids]); frames = [seq.frame(j) for j in ids]; mf = seq.frame(f)
    for fb in (False, True):
        for _ in range(3): ctx.process_frame(seq.cams[f], mf, cams, frames, fb, copy=False)
        t0 = time.perf_counter()
        for _ in range(50): ctx.process_frame(seq.cams[f], mf, cams, frames, fb, copy=False)
        print("mvs_process_frame, %d x %d, %d side views, %s flow: %.2f ms per main frame" % (seq.W, seq.H, len(ids), "Farneback" if fb else "variational", (time.perf_counter() - t0) / 50 * 1e3))
    # the same main frame with its frames taken from the frame store (mvs_process_frame_slots): uploaded once, not once per call
    ctx.frame_store(len(ids) + 1)
    for k, fr in enumerate([mf] + frames): ctx.frame_upload(k, fr)
    slots = list(range(1, len(ids) + 1))
    for fb in (False, True):
        for _ in range(3): ctx.process_frame_slots(seq.cams[f], 0, cams, slots, fb, copy=False)
        t0 = time.perf_counter()
        for _ in range(50): ctx.process_frame_slots(seq.cams[f], 0, cams, slots, fb, copy=False)
        print("mvs_process_frame_slots (frames in the frame store), %s flow: %.2f ms per main frame" % ("Farneback" if fb else "variational", (time.perf_counter() - t0) / 50 * 1e3))
