"""CPU tests of the calculateFlow() restatement (flow.cpp:19-42): Farneback and variational refinement."""
import numpy as np


def _pair(W, H, dx, dy):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)

    def tex(x, y):
        return 127 + 50 * np.sin(x / 7.0) * np.cos(y / 9.0) + 40 * np.sin((x + y) / 13.0) + 30 * np.cos((x - 2 * y) / 17.0)
    return tex(xx, yy).clip(0, 255).astype(np.uint8), tex(xx - dx, yy - dy).clip(0, 255).astype(np.uint8)


def test_farneback_level_geometry(oracle):
    """11 levels (k = 10..0), coarsest 69x52 at 640x480, 206x116 at 1080p"""
    import ctypes as C
    f = oracle.lib.orc_farneback_levels
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    for (W, H, cw, ch) in [(640, 480, 69, 52), (1280, 720, 137, 77), (1920, 1080, 206, 116), (3840, 2160, 412, 232)]:
        lw, lh, ls = (C.c_int * 64)(), (C.c_int * 64)(), (C.c_double * 64)()
        L = f(W, H, 10, 0.8, lw, lh, ls)
        assert L == 10 and (lw[10], lh[10]) == (cw, ch) and (lw[0], lh[0]) == (W, H)
    lw, lh, ls = (C.c_int * 64)(), (C.c_int * 64)(), (C.c_double * 64)()
    assert f(100, 60, 10, 0.8, lw, lh, ls) == 2  # 60 * 0.8^3 < 32 stops the pyramid


def test_farneback_tracks_translation(oracle):
    W, H = 320, 240
    for dx, dy in [(0.5, -0.3), (3.0, 2.0), (8.0, -5.0)]:
        a, b = _pair(W, H, dx, dy)
        c = oracle.farneback(a, b)[40:-40, 40:-40]
        # direction and magnitude (the +1e-3 regulariser of the 2x2 solve biases magnitudes low on smooth texture)
        assert abs(np.median(c[..., 0]) / dx - 0.8) < 0.15 and abs(np.median(c[..., 1]) / dy - 0.8) < 0.15
        assert np.mean(np.sign(c[..., 0]) == np.sign(dx)) > 0.97
    a, _ = _pair(W, H, 0, 0)
    # identical frames: zero flow away from the last row/column (which OpenCV's UpdateMatrices treats as out of range)
    same = np.abs(oracle.farneback(a, a))
    assert np.median(same) < 1e-5 and np.mean(same < 1e-2) > 0.8 and same.max() < 0.5


def test_variational_refinement_reduces_residual(oracle):
    W, H = 160, 120
    a, b = _pair(W, H, 0.4, -0.25)
    f0 = np.zeros((H, W, 2), np.float32)
    f1 = oracle.variational_refine(a, b)
    assert np.all(np.isfinite(f1))
    c = f1[20:-20, 20:-20]
    assert np.median(c[..., 0]) > 0 and np.median(c[..., 1]) < 0      # moves towards the true displacement
    # refining a flow that is already right keeps it (fixed point of the linearised system up to smoothing)
    truth = np.zeros((H, W, 2), np.float32)
    truth[..., 0], truth[..., 1] = 0.4, -0.25
    f2 = oracle.variational_refine(a, b, truth)[20:-20, 20:-20]
    assert np.abs(np.median(f2[..., 0]) - 0.4) < 0.08 and np.abs(np.median(f2[..., 1]) + 0.25) < 0.08
    same = oracle.variational_refine(a, a)
    assert np.abs(same).max() < 1e-3
    assert f0.shape == f1.shape


def test_calculate_flow_packing(oracle):
    """flow.cpp:37-41: channels (u, v, variance, 0); variance = compare(prev, flowRemap(flow, next))"""
    W, H = 160, 120
    a, b = _pair(W, H, 1.5, 1.0)
    for farneback in (True, False):
        out = oracle.calculate_flow(a, b, farneback)
        assert out.shape == (H, W, 4) and not out[..., 3].any()
        flow = np.ascontiguousarray(out[..., :2])
        ref = oracle.farneback(a, b) if farneback else oracle.variational_refine(a, b)
        np.testing.assert_array_equal(flow, ref)
        np.testing.assert_array_equal(out[..., 2], oracle.compare(a, oracle.flow_remap(flow, b)))


def test_oracle_threads_do_not_change_a_bit(oracle):
    """the row loops of oracle/flow_oracle.c run under OpenMP: every value is computed by one thread in the order written, so 1 thread and
    many give the same bits (what makes the 1080p / 4K parity runs of tests/test_flow_gpu.py affordable)"""
    import ctypes as C
    import ctypes.util
    omp = C.CDLL(ctypes.util.find_library("gomp") or "libgomp.so.1")
    W, H = 200, 150
    a, b = _pair(W, H, 1.5, -1.0)
    res = {}
    for n in (1, 5):
        omp.omp_set_num_threads(n)
        res[n] = [oracle.calculate_flow(a, b, fb) for fb in (True, False)]
    omp.omp_set_num_threads(max(1, __import__("os").cpu_count() or 1))
    for x, y in zip(res[1], res[5]):
        np.testing.assert_array_equal(x, y)


def test_the_bench_flow_pair_carries_a_flow_the_algorithm_returns(oracle):
    """bench.py's flow block times mvs_flow on flow_pair(): Farneback must return at least 70 % of the pair's known shift at EVERY size the
    block reports (round 5 timed a pair on which it returned 0.1 %: the warp path ran on zero flow and the sanity field went unread).  The
    oracle is the stand-in here; tests/test_bench_gpu.py checks the library's own figures in the bench line."""
    import bench
    for (W, H) in bench.FLOW_BLOCK_SIZES:
        a, b, shift = bench.flow_pair(np, W, H, True)
        r = bench.flow_recovery(np, oracle.farneback(a, b), shift)
        assert r["recovered_fraction"] >= 0.70, (W, H, r)
        assert r["epe_vs_known_shift"] <= 0.35 * np.hypot(*shift), (W, H, r)
    # the variational step is a refinement from zero: it must move TOWARDS the sub-pixel shift (reported, not gated, in the bench line)
    a, b, shift = bench.flow_pair(np, 640, 480, False)
    r = bench.flow_recovery(np, oracle.variational_refine(a, b), shift)
    assert 0.0 < r["recovered_fraction"] < 1.0
