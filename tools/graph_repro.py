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
LAYOUT = [("f0, f1 (u8_to_f32_kernel)", 42, 44)] + \
         ([("gauss tmp (gauss_kernel<false>)", 10, 12), ("blur (gauss_kernel<true>)", 12, 14), ("level images (resize_linear_kernel<1>)", 14, 16),
           ("polyexp rows (polyexp_vert)", 16, 22), ("R0, R1 (polyexp_horiz)", 22, 32), ("coarse flows A / B (resize_linear_kernel<2>, iteration of level 1)", 37, 41),
           ("M (update_matrices_kernel, then farneback_iteration_* every second iteration)", 32, 37),
           ("M ping-pong / box sums (farneback_iteration_* every other iteration)", 0, 10)] if fb else
          [("variational work buffers, first half (warp_q5, diff_kernel, var_fixed_point_fused ...)", 0, 11), ("variational work buffers, second half", 11, 22)]) + \
         [("flow (farneback_iteration_* of level 0 / var_finish)", 44, 46), ("variance (remap_cubic + compare)", 46, 47), ("packed result (pack_flow4)", 47, 51)]

yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
tex = lambda x, y: 127 + 50 * np.sin(x / 7.0) * np.cos(y / 9.0) + 40 * np.sin((x + y) / 13.0) + 30 * np.cos((x - 2 * y) / 17.0)  # noqa: E731
a = tex(xx, yy).clip(0, 255).astype(np.uint8)
b = tex(xx - 2.5, yy + 1.5).clip(0, 255).astype(np.uint8)
lib = mvs_amd.load_library()
lib.mvs_test_flow_arena.restype = C.c_int
lib.mvs_test_flow_arena.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]


def arena(ctx):
    out = np.empty(51 * P, np.float32)
    rc = lib.mvs_test_flow_arena(ctx.h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
    assert rc == 0, rc
    return out


def differences(x, ref):
    """per buffer, in the order the kernels write them: how many floats differ"""
    return [{"buffer": name, "floats_differing": int(np.count_nonzero(x[lo * P:hi * P].view(np.uint32) != ref[lo * P:hi * P].view(np.uint32))), "of": (hi - lo) * P}
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
