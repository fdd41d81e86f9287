"""mvs_process_frame (recon.cpp:65-117 for one main frame: depth, 4 x (projected, mixBackground, calculateFlow), triangulatePixels) on the zatisi cameras at
640 x 480, both flow algorithms; and the same under MVS_SERIAL_FLOWS=1 / MVS_FB_LANES=1 when those are set in the environment.
python tools/time_process_frame.py [1080p]     (1080p: a synthetic scene at 1920 x 1080 with 4 side views, tests/test_pipeline_gpu.py's, instead of the zatisi cameras)"""
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
if len(sys.argv) > 1 and sys.argv[1] == "1080p":
    import scenes
    from mvs_amd import synth
    W, H = 1920, 1080
    verts, faces = scenes.heightfield_mesh(64, extent=1.4)
    sc = synth.Scene(freq_scale=1.0)
    centres = [[0.0, 0.0, 0.0], [0.15, 0.0, 0.0], [-0.1, 0.12, 0.03], [0.02, -0.14, -0.02], [0.1, 0.1, 0.0]]
    cam = [synth.camera_at(c, W, H) for c in centres]
    img = [sc.render(c, W, H) for c in centres]
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        ctx.frame_store(5)
        for k in range(5): ctx.frame_upload(k, img[k])
        for fb in (False, True):
            for name, call in (("mvs_process_frame", lambda: ctx.process_frame(cam[0], img[0], np.stack(cam[1:]), img[1:], fb, copy=False)),
                               ("mvs_process_frame_slots", lambda: ctx.process_frame_slots(cam[0], 0, np.stack(cam[1:]), [1, 2, 3, 4], fb, copy=False))):
                for _ in range(2): n = len(call())
                t0 = time.perf_counter()
                for _ in range(10): call()
                print("%s, 1920 x 1080, 4 side views, %s flow: %.2f ms per main frame (%d points)" % (name, "Farneback" if fb else "variational", (time.perf_counter() - t0) / 10 * 1e3, n))
