"""Wall time of the per-pair stages through the C ABI (host buffers in/out, as recon.cpp calls them)."""
import sys, time
sys.path.insert(0, 'mesh-reconstruction_amd/python'); sys.path.insert(0, 'tests')
import numpy as np, mvs_amd, scenes
from mvs_amd import synth
for (W, H) in [(640, 480), (1920, 1080)]:
    verts, faces = scenes.heightfield_mesh(128)
    sc = synth.Scene(freq_scale=W / 1920.0)
    main, side = synth.camera_at([0, 0, 0], W, H), synth.camera_at([0.15, 0, 0], W, H)
    a, b = sc.render([0, 0, 0], W, H), sc.render([0.15, 0, 0], W, H)
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        def t(f, n=5):
            f(); f(); ctx.synchronize(); t0 = time.perf_counter()   # two warm-ups: allocations, first launches
            for _ in range(n): r = f()
            return (time.perf_counter() - t0) / n * 1e3, r
        td, depth = t(lambda: ctx.depth(main))
        tp, proj = t(lambda: ctx.projected(main, b, side))
        tm, (mixed, d2) = t(lambda: ctx.mix_background(proj, a, depth))
        tc, _ = t(lambda: ctx.compare(a, mixed))
        tf, fl = t(lambda: ctx.flow(a, mixed, True), 3)
        tv, fv = t(lambda: ctx.flow(a, mixed, False), 3)
        tt, tri = t(lambda: ctx.triangulate([fv], main, side[None], d2, copy=False), 3)   # into a reused buffer, as a C caller would
        ctx.profile_enable(True); ctx.flow(a, mixed, True); ms, n = ctx.profile_read(True); ctx.flow(a, mixed, False); ms2, n2 = ctx.profile_read(True)
        print("%dx%d (%d faces): depth %.2f ms, projected %.2f, mixBackground %.2f, compare %.2f, flow farneback %.2f (device %.2f), "
              "flow variational %.2f (device %.2f), triangulate %.2f ms (%d pts)" % (W, H, faces.shape[0], td, tp, tm, tc, tf, ms[5], tv, ms2[5], tt, tri.shape[0]))
