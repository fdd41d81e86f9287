"""Reproducer of round 4's hipGraph finding: a graph of calculateFlow's kernel sequence that
was instantiated BEFORE the process's first mvs_poisson_surface call (hipFFT / rocFFT: run-time compiled kernels, new code objects, first
use of a large scratch allocation) replayed with wrong results AFTERWARDS.  The library launches eagerly since; the test hook
MVS_FLOW_GRAPH=1 (csrc/hooks.hpp) brings the replay back for this script only.

    python3 tools/graph_repro.py [farneback|variational] [W H]

Steps: (1) eager context: the reference result and the reference work arena; (2) graph context: capture + first replay -- compared;
(3) the first Poisson call of the process; (4) the SAME graph replayed -- compared buffer by buffer in the order the kernels write them,
so that the first buffer that differs names the first kernel whose output is wrong.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

os.environ["MVS_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import mvs_amd  # noqa: E402

alg = sys.argv[1] if len(sys.argv) > 1 else "farneback"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[2].isdigit() else (640, 480)
fb = alg == "farneback"
P = W * H
# the arena in the order calculateFlow's kernels write it (csrc/flow.hip: flow_prepare, farneback_device / variational_device), in units of P floats
# (after a call the buffers hold what the LAST pyramid level, level 0, left in them)
# (round 6: Farneback prepares all pyramid levels at once for frames up to 1280 x 720 -- blur / level images / R hold every level -- and its
# work area is 85 P floats; the variational path keeps 42 P)
WORK = 85 if fb else 42
LEVEL_PIXELS = sum(int(round(W * 0.8 ** k)) * int(round(H * 0.8 ** k)) for k in range(11))   # (the part of the level buffers that is written)
LAYOUT = [("f0, f1 (u8_to_f32_kernel)", WORK, WORK + 2)] + \
         ([("blur of every level (gauss_fused_levels_kernel)", 18, 40), ("level images (resize_levels_kernel)", 40, 40 + 2 * LEVEL_PIXELS / P),
           ("R0, R1 of every level (polyexp_levels_kernel)", 46, 46 + 10 * LEVEL_PIXELS / P), ("coarse flows A / B (upsample_update_kernel, iteration of level 1)", 81, 85),
           ("M (upsample_update_kernel, then farneback_iteration_* every second iteration)", 76, 81),
           ("M ping-pong (farneback_iteration_* every other iteration)", 0, 10)] if fb else
          [("variational work buffers, first half (warp_q5, diff_kernel, var_fixed_point_fused ...)", 0, 11), ("variational work buffers, second half", 11, 22)]) + \
         [("flow (farneback_iteration_* of level 0 / var_finish)", WORK + 2, WORK + 4), ("variance (remap_cubic + compare)", WORK + 4, WORK + 5), ("packed result (pack_flow4)", WORK + 5, WORK + 9)]

yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
tex = lambda x, y: 127 + 50 * np.sin(x / 7.0) * np.cos(y / 9.0) + 40 * np.sin((x + y) / 13.0) + 30 * np.cos((x - 2 * y) / 17.0)  # noqa: E731
a = tex(xx, yy).clip(0, 255).astype(np.uint8)
b = tex(xx - 2.5, yy + 1.5).clip(0, 255).astype(np.uint8)
lib = mvs_amd.load_library()
lib.mvs_test_flow_arena.restype = C.c_int
lib.mvs_test_flow_arena.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]


def arena(ctx):
    out = np.empty((WORK + 9) * P, np.float32)
    rc = lib.mvs_test_flow_arena(ctx.h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
    assert rc == 0, rc
    return out


def differences(x, ref):
    """per buffer, in the order the kernels write them: how many floats differ"""
    return [{"buffer": name, "floats_differing": int(np.count_nonzero(x[int(round(lo * P)):int(round(hi * P))].view(np.uint32) != ref[int(round(lo * P)):int(round(hi * P))].view(np.uint32))), "of": int(round((hi - lo) * P))}
            for name, lo, hi in LAYOUT]


os.environ.pop("MVS_FLOW_GRAPH", None)
with mvs_amd.Context(W, H) as ctx:          # (1) eager
    ref_out = ctx.flow(a, b, fb)
    ref_arena = arena(ctx)
os.environ["MVS_FLOW_GRAPH"] = "2" if "--kernel-memset" in sys.argv else "1"   # 2: the captured hipMemsetAsync nodes replaced by a zero-fill kernel
report = {"memset_nodes": "zero-fill kernel" if "--kernel-memset" in sys.argv else "hipMemsetAsync", "algorithm": alg, "size": [W, H], "device": None}
with mvs_amd.Context(W, H) as gctx:         # (2) graph: capture + first replay
    report["device"] = gctx.info()
    out1 = gctx.flow(a, b, fb)
    report["replay_before_poisson_equal"] = bool(np.array_equal(out1.view(np.uint32), ref_out.view(np.uint32)))
    rng = np.random.default_rng(3)          # (3) the process's first Poisson call
    d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.concatenate([d, np.ones((len(d), 1))], 1).astype(np.float32)
    v, f = mvs_amd.poisson_surface(pts, d.astype(np.float32))
    report["poisson_vertices"] = int(len(v))
    out2 = gctx.flow(a, b, fb)              # (4) the same graph again
    report["replay_after_poisson_equal"] = bool(np.array_equal(out2.view(np.uint32), ref_out.view(np.uint32)))
    diffs = differences(arena(gctx), ref_arena)
    report["buffers_in_write_order"] = diffs
    report["first_buffer_that_differs"] = next((d["buffer"] for d in diffs if d["floats_differing"]), None)
    report["nan_in_result"] = bool(np.isnan(out2).any())
    os.environ.pop("MVS_FLOW_GRAPH", None)
with mvs_amd.Context(W, H) as ectx:         # eager launches after the Poisson call: always right
    report["eager_after_poisson_equal"] = bool(np.array_equal(ectx.flow(a, b, fb).view(np.uint32), ref_out.view(np.uint32)))
print(json.dumps(report))
