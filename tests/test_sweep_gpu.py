"""GPU parity tests of the plane sweep: HIP kernels (through the C ABI) vs the CPU oracle, for both samplers
("fixed": contract v2, the library default; "exact": contract v1) -- each against the oracle's restatement of the same contract.

Tolerances (north_star: depth RMSE < 1e-4 vs the reference path):
  * packed volume cells (count<<24 | sum resp. count<<16 | sum): integer work -> bit-exact expected; the only
    admitted deviation is the Newton reciprocal's documented miss (significand all ones, probability 2^-23
    per sample), budgeted as <= 1e-6 of the cells;
  * depth maps: RMSE < 1e-4 in NDC z and identical arg-min indices except at those cells.
"""
import numpy as np
import pytest

import mvs_amd
from mvs_amd import synth

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-4
CELL_BUDGET = 1e-6


@pytest.fixture(params=["fixed", "exact"])
def sampler(request):
    return request.param


def _rot_cam(W, H, center, yaw, pitch, **kw):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return synth.camera_at(center, W, H, rot=Rx @ Ry, **kw)


def _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, flags, z=(-1.0, 1.0), argmin=False, volume=True, sampler="exact"):
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D, z[0], z[1])
        ctx.sweep_run(0, len(sides), flags)
        if argmin:
            ctx.sweep_argmin()
        return ctx.sweep_fetch(want_volume=volume)


def _check(gpu, ref, D):
    depth, cost, idx, vol = gpu
    d_ref, c_ref, i_ref, v_ref = ref
    if vol is not None:
        bad = np.count_nonzero(vol != v_ref)
        assert bad <= max(1, int(vol.size * CELL_BUDGET)), "%d of %d volume cells differ" % (bad, vol.size)
    rmse = np.sqrt(np.mean((depth.astype(np.float64) - d_ref.astype(np.float64)) ** 2))
    assert rmse < RMSE_TOL, "depth RMSE %g" % rmse
    assert np.count_nonzero(idx != i_ref) <= max(1, int(idx.size * 1e-5))
    both = np.isfinite(c_ref) & np.isfinite(cost)
    assert np.array_equal(np.isfinite(c_ref), np.isfinite(cost))
    assert np.allclose(cost[both], c_ref[both], rtol=1e-6, atol=0) or np.count_nonzero(cost[both] != c_ref[both]) <= 2


def test_rcp_newton_is_correctly_rounded():
    """v_rcp_f32 + one FMA Newton step == IEEE 1/x for every significand, except possibly all-ones"""
    with mvs_amd.Context(64, 64) as ctx:
        for e in (1, 64, 100, 120, 126, 127, 128, 129, 135, 150, 200, 252):
            mism, allones = ctx.test_rcp(e)
            assert mism == allones and mism <= 1, "exponent %d: %d mismatches (%d at all-ones)" % (e, mism, allones)


def test_view_matrices_bit_exact(oracle):
    W, H = 640, 480
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, 3)
    side_cams = np.concatenate([side_cams, _rot_cam(W, H, [0.3, -0.2, 0.1], 0.1, -0.05)[None]])
    sides = sides + [sides[0]]
    with mvs_amd.Context(W, H) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, 4)
        q = ctx.sweep_view_matrices()
    for v in range(4):
        np.testing.assert_array_equal(q[v], oracle.view_matrix(main_cam, side_cams[v], W, H))


CASES = [
    # W, H, D, V, radius, rotated
    (64, 16, 16, 1, 0.3, False),      # exactly one tile, one chunk
    (200, 90, 20, 3, 0.3, False),     # ragged tiles, ragged last chunk
    (320, 240, 32, 4, 0.3, False),
    (333, 77, 7, 2, 0.5, True),       # odd sizes (P % 4 != 0), rotated side views
    (640, 480, 32, 4, 0.15, True),    # BASELINE config c1 shape
]


@pytest.mark.parametrize("W,H,D,V,radius,rot", CASES)
@pytest.mark.parametrize("kernel", ["generic", "tiled"])
def test_sweep_matches_oracle(oracle, W, H, D, V, radius, rot, kernel, sampler):
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=radius, freq_scale=max(W / 1920.0, 0.25))
    if rot:
        side_cams = side_cams.copy()
        side_cams[0] = _rot_cam(W, H, [radius, 0.02, 0.05], 0.05, -0.04)
        if V > 1:
            side_cams[1] = _rot_cam(W, H, [-0.9, 0.6, -0.3], -0.45, 0.3)  # partly out of frame
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler=sampler)
    flags = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    if kernel == "generic":
        flags |= mvs_amd.MVS_SWEEP_FORCE_GENERIC
    _check(_gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, flags, sampler=sampler), ref, D)


def test_argmin_kernel_equals_fused_and_oracle(oracle, sampler):
    W, H, D, V = 320, 240, 24, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler=sampler)
    fused = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_FUSED_ARGMIN, volume=False, sampler=sampler)
    sep = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_VOLUME, argmin=True, sampler=sampler)
    for a, b in zip(fused[:3], sep[:3]):
        np.testing.assert_array_equal(a, b)
    _check(sep, ref, D)


def test_no_views_and_behind_camera(oracle, sampler):
    W, H, D = 128, 64, 8
    main_cam, main_img, _, _, _ = synth.make_views(W, H, 0)
    depth, cost, idx, vol = _gpu_sweep(W, H, main_cam, main_img, np.zeros((0, 4, 4), np.float32), [], D,
                                       mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN, sampler=sampler)
    assert np.all(depth == np.float32(1.0)) and np.all(idx == -1) and np.all(np.isinf(cost)) and np.all(vol == 0)
    # a side camera looking the other way: every sample has w <= 0
    back = synth.camera_at([0, 0, -6.0], W, H, rot=np.diag([-1.0, 1.0, -1.0]))
    side = np.full((H, W), 77, np.uint8)
    ref = oracle.sweep(main_cam, main_img, back[None], [side], D, want_volume=True, sampler=sampler)
    for kernel_flag in (0, mvs_amd.MVS_SWEEP_FORCE_GENERIC):
        got = _gpu_sweep(W, H, main_cam, main_img, back[None], [side], D,
                         mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN | kernel_flag, sampler=sampler)
        _check(got, ref, D)


def test_one_call_form_and_float_volume(oracle, sampler):
    W, H, D, V = 256, 128, 16, 2
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    d_ref, c_ref, i_ref, v_ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, sampler=sampler)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        depth, cost, vol = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True, want_volume=True)
    np.testing.assert_array_equal(depth, d_ref)
    if sampler == "exact":
        cnt = (v_ref >> 16).astype(np.float32)
        s = (v_ref & 0xffff).astype(np.float32)
    else:   # sums in 1/255 grey levels: cost = sum / (255 count), one f32 division of two exactly representable integers
        cnt = (255 * (v_ref >> 24)).astype(np.float32)
        s = (v_ref & 0xffffff).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        expect = np.where(cnt > 0, s / cnt, np.inf).astype(np.float32)
    np.testing.assert_array_equal(vol, expect)
    np.testing.assert_array_equal(cost, c_ref)


def test_yaml_track_cameras(oracle, sampler):
    """BASELINE config c1: cameras of tracks/koberec.yaml (640x480), 32 planes, 4 side views, synthetic frames"""
    import tracks_yaml
    t = tracks_yaml.load("koberec.yaml")
    W, H = t["width"], t["height"]
    cams = t["cameras"]
    main = cams[0]
    ids = [5, 10, 15, 20]
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (H // 8 + 2, W // 8 + 2)).astype(np.float64)
    big = np.kron(base, np.ones((8, 8)))[:H, :W]
    main_img = big.astype(np.uint8)
    sides = [np.roll(big, (i, 2 * i), (0, 1)).astype(np.uint8) for i in range(4)]
    side_cams = np.stack([cams[i] for i in ids])
    ref = oracle.sweep(main, main_img, side_cams, sides, 32, want_volume=True, nthreads=8, sampler=sampler)
    for kernel_flag in (0, mvs_amd.MVS_SWEEP_FORCE_GENERIC):
        got = _gpu_sweep(W, H, main, main_img, side_cams, sides, 32,
                         mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN | kernel_flag, sampler=sampler)
        _check(got, ref, 32)


def test_view_shards_sum_to_full_volume(sampler):
    """the all-reduce invariant on the device: shard volumes add (as integers) to the full volume"""
    W, H, D, V = 320, 160, 16, 4
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME)
        full = ctx.sweep_fetch(want_volume=True)[3].copy()
        acc = np.zeros_like(full)
        for r in range(2):
            ctx.sweep_run(2 * r, 2, mvs_amd.MVS_SWEEP_VOLUME)
            acc += ctx.sweep_fetch(want_volume=True)[3]
    np.testing.assert_array_equal(acc, full)


def test_c2_full_size_tiled_vs_oracle(oracle, sampler):
    """BASELINE config c2 at full size (1280x720, 64 planes, 8 views): tiled kernel vs oracle"""
    W, H, D, V = 1280, 720, 64, 8
    main_cam, main_img, side_cams, sides, gt = synth.make_views(W, H, V)
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler=sampler)
    got = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_VOLUME, argmin=True, sampler=sampler)
    _check(got, ref, D)
    # the sweep recovers the analytic surface it was rendered from
    err = np.abs(got[0] - gt)[16:-16, 16:-16]
    assert np.median(err) <= 2.0 / D


def test_c3_full_size_properties(sampler):
    """BASELINE config c3 (1920x1080, 128 planes, 16 views): size-independent properties"""
    W, H, D, V = 1920, 1080, 128, 16
    main_cam, main_img, side_cams, sides, gt = synth.make_views(W, H, V)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME)
        ctx.sweep_argmin()
        d_t, c_t, i_t, _ = ctx.sweep_fetch()
        # (1) fused selection == separate selection
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        d_f, c_f, i_f, _ = ctx.sweep_fetch()
        np.testing.assert_array_equal(i_t, i_f)
        np.testing.assert_array_equal(d_t, d_f)
        assert ctx.plan_shape() == (5 if sampler == "exact" else 4)   # the ring is rectified: sweep_exact_rect / sweep_fx_rect served the runs above
        # (2) tiled == generic kernel, index for index
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN | mvs_amd.MVS_SWEEP_FORCE_GENERIC)
        d_g, c_g, i_g, _ = ctx.sweep_fetch()
        np.testing.assert_array_equal(i_t, i_g)
        np.testing.assert_array_equal(c_t, c_g)
        # (3) exact sampler: the other thread shape of the tiled kernel (the planner picks 2 x 32 here), index for index;
        #     fixed sampler: the general-camera code path (per-sample reciprocal) on this plane-independent-w geometry
        #     (the ring is rectified: the fixed sampler's runs above took sweep_fx_rect, plan shape 4; here the general tiled kernel,
        #     with and without its plane-independent-w shortcut)
        for extra in ((8 << 8, mvs_amd.MVS_SWEEP_NO_RECT) if sampler == "exact" else (mvs_amd.MVS_SWEEP_NO_RECT, mvs_amd.MVS_SWEEP_NO_RECT | (4 << 8))):
            ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN | extra)
            d_s, c_s, i_s, _ = ctx.sweep_fetch()
            np.testing.assert_array_equal(i_t, i_s)
            np.testing.assert_array_equal(c_t, c_s)
        # (4) linearity over views, as a checksum of the packed cells: shards [0,8) and [8,16) add up to the full volume
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME)
        full = int(ctx.sweep_fetch(want_volume=True)[3].sum(dtype=np.uint64))
        parts = 0
        for v0 in (0, 8):
            ctx.sweep_run(v0, 8, mvs_amd.MVS_SWEEP_VOLUME)
            parts += int(ctx.sweep_fetch(want_volume=True)[3].sum(dtype=np.uint64))
        assert parts == full
    err = np.abs(d_t - gt)[16:-16, 16:-16]
    assert np.median(err) <= 2.0 / D
    assert np.mean(err <= 3.0 / D) > 0.97


def test_c3_full_size_vs_oracle(oracle, sampler):
    """BASELINE config c3 at full size against the oracle, every cell of the 1 GB volume and every depth (the oracle takes
    a couple of seconds on the GPU box's host threads -- it is bench.py's cpu_baseline on the same workload)"""
    import os
    W, H, D, V = 1920, 1080, 128, 16
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V)
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=min(256, os.cpu_count() or 8), sampler=sampler)
    got = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN, sampler=sampler)
    _check(got, ref, D)


def test_c3_full_size_general_cameras_vs_oracle(oracle, sampler):
    """the same size with cameras that are NOT in the main camera's focal plane (every view rotated a little, two of them strongly and
    partly out of frame): the reciprocal-per-sample path that the bundled tracks take, BORDER and SKIP regions in every chunk; depth,
    cost and index of every pixel and every cell of the volume"""
    import os
    W, H, D, V = 1920, 1080, 128, 16
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V)
    side_cams = side_cams.copy()
    rng = np.random.default_rng(11)
    for v in range(V):
        a = 2.0 * np.pi * v / V
        side_cams[v] = _rot_cam(W, H, [0.15 * np.cos(a), 0.15 * np.sin(a), float(rng.uniform(-0.05, 0.05))], float(rng.uniform(-0.03, 0.03)),
                                float(rng.uniform(-0.03, 0.03)))
    side_cams[5] = _rot_cam(W, H, [-0.9, 0.6, -0.3], -0.45, 0.3)
    side_cams[11] = _rot_cam(W, H, [0.5, -0.2, 0.1], 0.25, -0.1)
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=min(256, os.cpu_count() or 8), sampler=sampler)
    got = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN, sampler=sampler)
    _check(got, ref, D)
    shift = 24 if sampler == "fixed" else 16
    counts = got[3] >> shift
    assert counts.max() >= V - 2 and counts.mean() > V / 4 and (counts < counts.max()).mean() > 0.05, "views must overlap, and some cells must miss some"


def test_c4_full_size_depth_vs_oracle(oracle, sampler):
    """BASELINE config c4 (3840x2160, 256 planes, 32 views) at full size on i.i.d. noise frames (SURVEY 8d's adversarial
    input): depth, cost and index of every pixel against the oracle; the 8.5 GB volume is not materialised"""
    import os
    W, H, D, V = 3840, 2160, 256, 32
    main_cam, main_img, side_cams, sides = synth.noise_views(W, H, V)
    d_ref, c_ref, i_ref, _ = oracle.sweep(main_cam, main_img, side_cams, sides, D, nthreads=min(256, os.cpu_count() or 8), sampler=sampler)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        depth, cost, idx, _ = ctx.sweep_fetch()
    np.testing.assert_array_equal(idx, i_ref)
    np.testing.assert_array_equal(depth, d_ref)
    np.testing.assert_array_equal(cost, c_ref)


def _random_camera(rng, W, H, spread, max_angle):
    c = rng.uniform(-spread, spread, 3) * np.array([1.0, 1.0, 0.3])
    yaw, pitch, roll = rng.uniform(-max_angle, max_angle, 3)
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    near = rng.uniform(0.8, 2.0)
    return synth.camera_at(c, W, H, near=near, far=near * rng.uniform(3.0, 6.0), rot=Rz @ Rx @ Ry,
                           fovx=rng.uniform(0.6, 1.4))


@pytest.mark.parametrize("seed", range(8))
def test_random_geometry_stress(oracle, seed, sampler):
    """random sizes, plane counts and wildly different side cameras (wide baselines, rotations up to 40 degrees, other
    intrinsics): exercises the planner's FAST / BORDER / SKIP / GENERIC modes and ragged tiles; tiled == generic ==
    oracle, cell for cell"""
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.integers(2, 400))
    H = int(rng.integers(2, 200))
    D = int(rng.integers(1, 40))
    V = int(rng.integers(1, 6))
    main_cam = _random_camera(rng, W, H, 0.05, 0.05)
    side_cams = np.stack([_random_camera(rng, W, H, rng.choice([0.2, 1.5]), rng.choice([0.05, 0.7])) for _ in range(V)])
    main_img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    sides = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(V)]
    z = (float(rng.uniform(-1.0, -0.2)), float(rng.uniform(0.2, 1.0)))
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, z[0], z[1], want_volume=True, nthreads=8, sampler=sampler)
    flags = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    tiled = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, flags, z=z, sampler=sampler)
    generic = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, flags | mvs_amd.MVS_SWEEP_FORCE_GENERIC, z=z, sampler=sampler)
    np.testing.assert_array_equal(tiled[3], generic[3])
    np.testing.assert_array_equal(tiled[2], generic[2])
    _check(tiled, ref, D)
    _check(generic, ref, D)


def test_argument_errors():
    with mvs_amd.Context(64, 32) as ctx:
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_run(0, 1, mvs_amd.MVS_SWEEP_VOLUME)          # nothing set
        main_cam, main_img, side_cams, sides, _ = synth.make_views(64, 32, 2)
        ctx.sweep_set(main_cam, main_img, side_cams, sides, 4)
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_run(1, 2, mvs_amd.MVS_SWEEP_VOLUME)          # view range past the end
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_run(0, 2, 0)                                 # neither volume nor fused
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_set(main_cam, main_img, side_cams, sides, 0)  # zero planes
    with pytest.raises(mvs_amd.MvsError):
        mvs_amd.Context(1, 1)
    # the widest frame the library accepts: the fixed sampler's 1/256-texel coordinates stop at 16383, the exact sampler takes it
    W, H = 16384, 2
    cam = synth.camera_at([0.0, 0.0, 0.0], W, H)
    img = np.zeros((H, W), np.uint8)
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        ctx.sweep_set(cam, img, cam[None], [img], 2)
        with pytest.raises(mvs_amd.MvsError, match="16383"):
            ctx.sweep_run(0, 1, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        ctx.set_sampler("exact")
        ctx.sweep_run(0, 1, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        assert (ctx.sweep_fetch()[2] == 0).all()                   # identical black frames: every plane costs 0, the first one wins


def test_plane_groups_reproduce_the_full_volume(sampler):
    """mvs_sweep_run_planes over consecutive plane groups == one full run (the pipelined all-reduce path of bench.py)"""
    from mvs_amd import dist as mdist
    W, H, D, V = 320, 160, 40, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME)
        ctx.sweep_argmin()
        d_full, c_full, i_full, v_full = [a.copy() for a in ctx.sweep_fetch(want_volume=True)]
        for kernel_flag in (0, mvs_amd.MVS_SWEEP_FORCE_GENERIC):
            # poison the volume, then rebuild it group by group
            ctx.sweep_run(0, 0, mvs_amd.MVS_SWEEP_VOLUME | kernel_flag)
            for first, count in mdist.plane_groups(D, 3, ctx.plane_granularity()):
                ctx.sweep_run_planes(0, V, first, count, mvs_amd.MVS_SWEEP_VOLUME | kernel_flag)
            ctx.sweep_argmin()
            d, c, i, v = ctx.sweep_fetch(want_volume=True)
            np.testing.assert_array_equal(v, v_full)
            np.testing.assert_array_equal(i, i_full)
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_run_planes(0, V, 8, 16, mvs_amd.MVS_SWEEP_VOLUME)   # not on the granularity
    assert mdist.plane_groups(128, 4, 16) == [(0, 32), (32, 32), (64, 32), (96, 32)]
    assert mdist.plane_groups(40, 3, 16) == [(0, 16), (16, 16), (32, 8)]
    assert mdist.plane_groups(7, 4, 16) == [(0, 7)]


def test_partial_selection_and_merge_equal_the_one_step_argmin(sampler):
    """mvs_sweep_argmin_partial over plane slices + mvs_sweep_combine_partials == mvs_sweep_argmin (the reduce-scatter form of the
    view-sharded pipeline: a rank selects over the planes it owns, the 8-byte partials are merged in ascending plane order);
    ties included (flat frames: every plane ties, the lowest must win), parts of unequal size"""
    import torch
    W, H, D, V = 320, 160, 40, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    flat = np.full((H, W), 90, np.uint8)
    P = W * H
    for img, sd in ((main_img, sides), (flat, [flat] * V)):
        with mvs_amd.Context(W, H, sampler=sampler) as ctx:
            # torch fills / frees on ITS stream, the library works on its own: allocate without a fill and synchronise before
            # handing the buffers over (bench.py shares one explicit stream instead)
            vol_t = torch.empty(D * P, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ctx.sweep_use_volume(vol_t.data_ptr(), vol_t.numel() * 4)
            ctx.sweep_set(main_cam, img, side_cams, sd, D)
            ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME)
            ctx.sweep_argmin()
            d0, c0, i0, _ = [a.copy() if a is not None else None for a in ctx.sweep_fetch()]
            for nparts in (1, 2, 3, 8):
                bounds = [round(k * D / nparts) for k in range(nparts + 1)]
                parts_t = torch.empty(nparts * P, dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                ctx.sweep_run(0, 0, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)            # poison depth / cost / index
                for k in range(nparts):
                    a, b = bounds[k], bounds[k + 1]
                    ctx.sweep_argmin_partial(vol_t.data_ptr() + 4 * a * P, a, b - a, parts_t.data_ptr() + 8 * k * P)
                ctx.sweep_combine_partials(parts_t.data_ptr(), nparts)
                d, c, i, _ = ctx.sweep_fetch()                                  # synchronises the library's stream
                np.testing.assert_array_equal(i, i0)
                np.testing.assert_array_equal(d, d0)
                np.testing.assert_array_equal(c, c0)
            with pytest.raises(mvs_amd.MvsError):
                ctx.sweep_argmin_partial(vol_t.data_ptr(), 30, 20, parts_t.data_ptr())   # planes 30..50 of 40
            ctx.sweep_use_volume(0, 0)


def test_row_bands_reproduce_the_full_result(sampler):
    """mvs_sweep_run_rows over the bands of mdist.row_bands == one full run, for the volume and the fused depth selection,
    tiled and generic kernels; rows outside a band are not touched (bench.py --shard rows)"""
    from mvs_amd import dist as mdist
    W, H, D, V = 320, 200, 24, 3            # 200 rows = 12 full tile rows + a ragged one
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        g = ctx.row_granularity()
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, both)
        d_full, c_full, i_full, v_full = [a.copy() for a in ctx.sweep_fetch(want_volume=True)]
        for kernel_flag in (0, mvs_amd.MVS_SWEEP_FORCE_GENERIC):
            for world in (2, 3, 5):
                ctx.sweep_run(0, 0, both | kernel_flag)          # poison: zero views -> empty cells, depth 1.0
                assert (ctx.sweep_fetch()[0] == 1.0).all()
                bands = mdist.row_bands(H, world, g)
                for k, (first, count) in enumerate(bands):
                    ctx.sweep_run_rows(first, count, 0, V, both | kernel_flag)
                    d, c, i, v = ctx.sweep_fetch(want_volume=True)
                    done = first + count
                    np.testing.assert_array_equal(d[:done], d_full[:done])
                    assert (d[done:] == 1.0).all() and (v[:, done:] == 0).all()      # later bands still untouched
                np.testing.assert_array_equal(v, v_full)
                np.testing.assert_array_equal(i, i_full)
                np.testing.assert_array_equal(c, c_full)
        ctx.sweep_run_rows(0, 0, 0, V, both)                     # an empty band is legal (rank beyond the units)
        for bad in ((g // 2, g), (0, g + g // 2), (192, 16), (-g, g)):   # boundaries off the granularity (8 fixed, 16 exact), past H, negative
            with pytest.raises(mvs_amd.MvsError):
                ctx.sweep_run_rows(bad[0], bad[1], 0, V, both)
        ctx.sweep_run_rows(192, 8, 0, V, both)                   # ragged last band ends at H


def test_plane_split_launches_are_bit_identical(sampler):
    """launches with too few tiles to fill the chip hand each workgroup a subset of a tile's plane chunks and merge the
    partial bests (combine_best); any split count -- forced through the timing-experiment bits -- must give the
    unsplit result, ties included (noise frames produce many equal-cost planes)"""
    W, H, D, V = 200, 100, 70, 3            # 5 plane chunks, ragged last one
    rng = np.random.default_rng(11)
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    flat = np.full((H, W), 90, np.uint8)    # constant frames: every in-frame plane ties -> lowest d must win
    noise_side = rng.integers(0, 4, (H, W), dtype=np.uint8)           # 4 grey levels: many equal-cost planes
    cases = [(main_img, sides), (flat, [flat] * V), (rng.integers(0, 256, (H, W), dtype=np.uint8), [noise_side] * V)]
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        for img, sd in cases:
            ctx.sweep_set(main_cam, img, side_cams, sd, D)
            ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME)
            ctx.sweep_argmin()                                   # reference: separate pass over the volume
            d0, c0, i0, v0 = [a.copy() for a in ctx.sweep_fetch(want_volume=True)]
            for forced in (1, 2, 3, 5, 9):
                # (the ring is rectified: with the fixed sampler `both` takes sweep_fx_rect; NO_RECT keeps the general tiled kernel covered)
                for flags in (both, mvs_amd.MVS_SWEEP_FUSED_ARGMIN, both | mvs_amd.MVS_SWEEP_NO_RECT, mvs_amd.MVS_SWEEP_FUSED_ARGMIN | mvs_amd.MVS_SWEEP_NO_RECT):
                    ctx.sweep_run(0, 0, both)
                    ctx.sweep_run(0, V, flags | (forced << 16))
                    d, c, i, v = ctx.sweep_fetch(want_volume=bool(flags & mvs_amd.MVS_SWEEP_VOLUME))
                    np.testing.assert_array_equal(i, i0)
                    np.testing.assert_array_equal(d, d0)
                    np.testing.assert_array_equal(c, c0)
                    if v is not None:
                        np.testing.assert_array_equal(v, v0)


def test_both_thread_shapes_are_bit_identical(oracle):
    """the tiled kernel exists as 2 px x 32 planes and 4 px x 16 planes per thread; the planner picks per (views, planes).
    Both must agree with each other and the oracle cell for cell: dense planes (32-plane footprints fit LDS -> 2 x 32)
    and coarse planes (they do not -> 4 x 16), ragged sizes, general cameras"""
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    seen = set()
    for (W, H, D, V, radius) in [(384, 200, 96, 3, 0.15), (330, 130, 70, 2, 0.15), (320, 160, 16, 3, 0.5), (384, 200, 200, 2, 0.05)]:
        main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=radius)
        side_cams = side_cams.copy()
        side_cams[0] = _rot_cam(W, H, [0.0, radius, 0.03], 0.01, 0.02)        # one general camera among the ring
        ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8)
        with mvs_amd.Context(W, H, sampler="exact") as ctx:
            ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
            assert ctx.plan_shape() == 0
            ctx.sweep_run(0, V, both)
            auto = [a.copy() for a in ctx.sweep_fetch(want_volume=True)]
            shape = ctx.plan_shape()
            seen.add(shape)
            ctx.sweep_run(0, 0, both)
            ctx.sweep_run(0, V, both | (8 << 8))                              # force 4 x 16
            assert ctx.plan_shape() == 2
            tall = ctx.sweep_fetch(want_volume=True)
            for a, b in zip(auto, tall):
                np.testing.assert_array_equal(a, b)
            ctx.sweep_run(0, V, both)                                        # back to the planner's choice
            assert ctx.plan_shape() == shape
        _check(auto, ref, D)
    assert seen == {1, 2}, "the cases above are meant to exercise both shapes, got %r" % seen


def test_plane_independent_w_path_is_bit_identical(oracle, sampler):
    """ring cameras (parallel axes, centres in the main focal plane) have Q[2][2] == 0 and take the hoisted-reciprocal
    path; it must equal the general path (debug bit) and the oracle cell for cell; a rotated camera must not take it"""
    W, H, D, V = 384, 200, 32, 4
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    for v in range(V):
        assert oracle.view_matrix(main_cam, side_cams[v], W, H)[2, 2] == 0.0
    rot = side_cams.copy()
    rot[1] = _rot_cam(W, H, [0.0, 0.3, 0.05], 0.01, 0.02)  # pitched and displaced along the optical axis
    assert oracle.view_matrix(main_cam, rot[1], W, H)[2, 2] != 0.0
    for cams in (side_cams, rot):
        ref = oracle.sweep(main_cam, main_img, cams, sides, D, want_volume=True, nthreads=8, sampler=sampler)
        fast = _gpu_sweep(W, H, main_cam, main_img, cams, sides, D, mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN, sampler=sampler)
        general = _gpu_sweep(W, H, main_cam, main_img, cams, sides, D,
                             mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN | mvs_amd.MVS_SWEEP_NO_RECT | (4 << 8), sampler=sampler)
        np.testing.assert_array_equal(fast[3], general[3])
        # (fixed sampler, ring cameras: `fast` above is the rectified kernel; the general kernel's hoisted-reciprocal path on its own)
        shortcut = _gpu_sweep(W, H, main_cam, main_img, cams, sides, D,
                              mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN | mvs_amd.MVS_SWEEP_NO_RECT, sampler=sampler)
        np.testing.assert_array_equal(shortcut[3], general[3])
        _check(fast, ref, D)


def test_fixed_sampler_region_prefetch_is_bit_identical(oracle):
    """the fixed sampler's region look-ahead (the next view's region is copied into the other half of the LDS rows while the current
    one is being sampled; debug bit 3 switches it off) must not change a cell; general cameras, border tiles, ragged sizes, and a
    wide-baseline case whose regions are too wide for a half row (falls back to one region at a time)"""
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    for (W, H, D, V, radius) in [(384, 200, 48, 5, 0.15), (330, 130, 70, 3, 0.4), (640, 96, 16, 4, 0.9)]:
        main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=radius)
        side_cams = side_cams.copy()
        side_cams[0] = _rot_cam(W, H, [0.0, radius, 0.03], 0.01, 0.02)
        ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler="fixed")
        ahead = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, both, sampler="fixed")
        plain = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, both | (8 << 8), sampler="fixed")
        for a, b in zip(plain, ahead):
            np.testing.assert_array_equal(a, b)
        _check(ahead, ref, D)


def test_many_views_use_batched_tables(oracle, sampler):
    """70 views: more than the fixed sampler's 64-entry constant tables, so they are reloaded in batches within a chunk (no look-ahead
    across a batch boundary), and more than fit one per-workgroup descriptor table; 16 views x 3 chunks in ONE workgroup (forced
    split count 1) is the opposite case, the table loaded once and the look-ahead running across chunk boundaries"""
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    for (W, H, D, V, splits) in [(128, 40, 40, 70, 0), (128, 40, 40, 70, 1), (192, 48, 44, 16, 1), (192, 48, 44, 16, 3)]:
        main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.2, freq_scale=0.25)
        side_cams = side_cams.copy()
        side_cams[V // 2] = _rot_cam(W, H, [-0.9, 0.6, -0.3], -0.45, 0.3)   # partly out of frame: SKIP / BORDER regions in the sequence
        ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler=sampler)
        _check(_gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, both | (splits << 16), sampler=sampler), ref, D)


def test_one_context_through_sampler_and_view_changes(oracle):
    """state carried by a context: the region plan, the quad images of both samplers (the exact sampler's is built on demand and must be
    rebuilt after new frames), plane tables -- a sequence of sampler switches, new views, new planes and a new main view, each sweep
    compared with the oracle"""
    W, H = 320, 136
    rng = np.random.default_rng(3)
    main_cam, main_img, cams_a, imgs_a, _ = synth.make_views(W, H, 5, radius=0.25, freq_scale=0.3)
    _, main_img_b, cams_b, imgs_b, _ = synth.make_views(W, H, 3, radius=0.4, freq_scale=0.5)
    imgs_c = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(5)]
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        def sweep_and_check(smp, main, cams, imgs, D, z=(-1.0, 1.0)):
            ref = oracle.sweep(main_cam, main, cams, imgs, D, z[0], z[1], want_volume=True, nthreads=8, sampler=smp)
            ctx.sweep_run(0, len(imgs), both)
            _check(ctx.sweep_fetch(want_volume=True), ref, D)
        ctx.sweep_set(main_cam, main_img, cams_a, imgs_a, 24)
        sweep_and_check("fixed", main_img, cams_a, imgs_a, 24)
        ctx.set_sampler("exact")                                   # same inputs: new plan, f16 quad image built now
        sweep_and_check("exact", main_img, cams_a, imgs_a, 24)
        ctx.sweep_set_views(cams_a, imgs_c)                        # new frames under the exact sampler: its quad image is stale
        sweep_and_check("exact", main_img, cams_a, imgs_c, 24)
        ctx.set_sampler("fixed")
        sweep_and_check("fixed", main_img, cams_a, imgs_c, 24)
        ctx.sweep_set_planes(40, -0.8, 0.9)                        # new planes: new plan, same images
        sweep_and_check("fixed", main_img, cams_a, imgs_c, 40, (-0.8, 0.9))
        ctx.sweep_set(main_cam, main_img_b, cams_b, imgs_b, 40, -0.8, 0.9)   # fewer views, another main frame
        sweep_and_check("fixed", main_img_b, cams_b, imgs_b, 40, (-0.8, 0.9))
        ctx.set_sampler("exact")
        sweep_and_check("exact", main_img_b, cams_b, imgs_b, 40, (-0.8, 0.9))


def test_255_views_at_maximal_cost(oracle, sampler):
    """the fixed sampler's cell limit: 255 views, a black main frame against nearly white side frames -- sums of 1.6e7 (24 bits) and
    cross products above 2^31 in the depth selection (they are compared unsigned: HIP's __umul24 returns int)"""
    W, H, D, V = 64, 16, 16, 255
    rng = np.random.default_rng(77)
    main_cam, _, side_cams, _, _ = synth.make_views(W, H, V, radius=0.05, freq_scale=0.25)
    main_img = np.zeros((H, W), np.uint8)
    sides = [rng.integers(250, 256, (H, W), dtype=np.uint8) for _ in range(V)]
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler=sampler)
    gpu = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, both, sampler=sampler)
    _check(gpu, ref, D)
    shift = 24 if sampler == "fixed" else 16
    full = (gpu[3] >> shift) == V
    assert full.any()
    if sampler == "fixed":
        s = (gpu[3][full] & 0xffffff).astype(np.uint64)
        assert (s * V).max() > 2**31, "the case must reach products the signed comparison would have ordered wrongly"
    sep = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_VOLUME, argmin=True, sampler=sampler)
    np.testing.assert_array_equal(sep[2], gpu[2])   # the separate depth-selection kernel agrees with the fused one


def test_sweep_matches_the_committed_golden_vectors(sampler):
    """HIP through the C ABI against tests/golden/sweep_small*.npz (written by the oracle via tests/golden/make_golden.py and
    committed): no oracle runs here, so the HIP path and the oracle cannot drift together unnoticed"""
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(golden, "sweep_small.npz"))
    f = g if sampler == "exact" else np.load(os.path.join(golden, "sweep_small_fx.npz"))
    H, W = g["main_img"].shape
    for kernel_flag in (0, mvs_amd.MVS_SWEEP_FORCE_GENERIC):
        depth, cost, idx, vol = _gpu_sweep(W, H, g["main_cam"], g["main_img"], g["side_cams"], list(g["side_imgs"]), int(f["D"]),
                                           mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN | kernel_flag, sampler=sampler)
        np.testing.assert_array_equal(vol, f["vol"])
        np.testing.assert_array_equal(idx, f["idx"])
        np.testing.assert_array_equal(depth, f["depth"])
        np.testing.assert_array_equal(cost, f["cost"])


def test_rectified_kernels_match_the_committed_golden_vectors(sampler):
    """sweep_fx_rect / sweep_exact_rect through the C ABI against tests/golden/sweep_rect_small.npz (the oracle's output, committed): no
    oracle runs here; the plan shape says the rectified kernel served the run, and the general kernels must give the same bytes"""
    import os
    import zlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sweep_rect_small.npz"))
    H, W = g["main_img"].shape
    V, D = len(g["side_imgs"]), int(g["D"])
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(g["main_cam"], g["main_img"], g["side_cams"], list(g["side_imgs"]), D)
        for flags, shape in ((both, 4 if sampler == "fixed" else 5), (both | mvs_amd.MVS_SWEEP_NO_RECT, None)):
            ctx.sweep_run(0, V, flags)
            if shape is not None:
                assert ctx.plan_shape() == shape
            depth, cost, idx, vol = ctx.sweep_fetch(want_volume=True)
            np.testing.assert_array_equal(depth, g["depth_" + sampler])
            np.testing.assert_array_equal(cost, g["cost_" + sampler])
            np.testing.assert_array_equal(idx, g["idx_" + sampler])
            np.testing.assert_array_equal(vol[::2, ::2, ::2], g["vol_probe_" + sampler])
            assert np.uint32(zlib.crc32(np.ascontiguousarray(vol).tobytes())) == g["vol_crc_" + sampler]


def _roll_cam(W, H, center, roll, yaw=0.0, pitch=0.0):
    cr, sr = np.cos(roll), np.sin(roll)
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return synth.camera_at(center, W, H, rot=Rz @ Rx @ Ry)


BAND_CASES = {
    # name: (W, H, side cameras as a function of the ring's, can the pipeline serve them?)
    "ring": (640, 512, None, True),
    "ragged height": (648, 500, None, True),
    "small": (320, 136, None, True),
    "odd width": (333, 301, None, True),
    "general": (640, 512, lambda W, H, c: [_rot_cam(W, H, [0.15 * np.cos(v), 0.15 * np.sin(v), 0.02 * v], 0.02 * (v - 2), -0.015 * v) for v in range(len(c))], True),
    "upside down": (640, 512, lambda W, H, c: [_roll_cam(W, H, [0.1, 0.05 * v, 0.0], np.pi + 0.1 * v) for v in range(len(c))], True),
    "on its side and far off": (640, 512, lambda W, H, c: [_roll_cam(W, H, [0.1, 0.1, 0.0], 0.5 * np.pi), _roll_cam(W, H, [-0.9, 0.6, -0.3], 0.3, -0.45, 0.3)] +
                                [_roll_cam(W, H, [0.0, -0.12, 0.0], -0.7, 0.0, 0.2)] * (len(c) - 2), True),
    "one camera behind": (640, 512, lambda W, H, c: list(c[:-1]) + [synth.camera_at([0.0, 0.0, -6.0], W, H, rot=np.diag([-1.0, 1.0, -1.0]))], False),
}


@pytest.mark.parametrize("case", list(BAND_CASES))
def test_one_call_band_pipeline_is_bit_identical(case, monkeypatch):
    """mvs_sweep with the fixed sampler sends the side frames over in row bands and sweeps band b while the rows of band b + 1 cross the bus
    (sweep.hip: BandPipeline; by default two bands, for view sets of 16 MB and more -- here forced through the test hook).  Depth and cost
    equal the resident form's (set + run + fetch) and the unbanded one-call's bit for bit, whatever the cameras do to the order of the side
    rows; afterwards the context holds the COMPLETE views (a later mvs_sweep_run reads all rows); and a camera that sees part of the sweep
    from behind (no bound on the rows a band samples) takes the unbanded path."""
    W, H, cams_of, pipelined = BAND_CASES[case]
    D, V = 32, 5
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.2)
    if cams_of is not None:
        side_cams = np.stack([np.asarray(c, np.float32) for c in cams_of(W, H, side_cams)])
    rng = np.random.default_rng(5)
    sides = [np.ascontiguousarray(np.clip(s.astype(np.int32) + rng.integers(-9, 10, s.shape), 0, 255).astype(np.uint8)) for s in sides]
    want = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN, sampler="fixed")
    assert np.isfinite(want[1]).mean() > 0.3, "the views must overlap the main view"
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:   # no hook: a view set this small goes up in one piece
        depth, cost = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
        assert ctx.onecall_bands() == 0
        np.testing.assert_array_equal(depth, want[0])
        np.testing.assert_array_equal(cost, want[1])
    for forced in (2, 3, 4, 8):
        monkeypatch.setenv("MVS_ONECALL_BANDS", str(forced))
        with mvs_amd.Context(W, H, sampler="fixed") as ctx:
            for _ in range(2):   # the second call finds the plan in the cache: band 0 is staged outside the planner's hook
                depth, cost = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
                assert ctx.onecall_bands() == (min(forced, H // 64) if pipelined else 0)
                np.testing.assert_array_equal(depth, want[0])
                np.testing.assert_array_equal(cost, want[1])
            if forced == 2:
                ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
                again = ctx.sweep_fetch(want_volume=True)
                for a, b in zip(again, want):
                    np.testing.assert_array_equal(a, b)
                # a volume request goes through the unbanded path
                d2, vol = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True)
                assert ctx.onecall_bands() == 0
                np.testing.assert_array_equal(d2, want[0])


def test_one_call_at_c3_is_pipelined_by_default():
    """BASELINE c3's frames (16 views of 1920 x 1080: 33 MB) through mvs_sweep: two row bands without any hook, the resident form's bits"""
    W, H, D, V = 1920, 1080, 16, 16
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V)
    want = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, mvs_amd.MVS_SWEEP_FUSED_ARGMIN, volume=False, sampler="fixed")
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        depth, cost = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
        assert ctx.onecall_bands() == 2
        np.testing.assert_array_equal(depth, want[0])
        np.testing.assert_array_equal(cost, want[1])
    with mvs_amd.Context(W, H, sampler="exact") as ctx:
        ctx.sweep(main_cam, main_img, side_cams, sides, D)
        assert ctx.onecall_bands() == 0


@pytest.mark.parametrize("W,H,D,V", [(640, 360, 48, 5), (333, 203, 21, 3), (1280, 720, 64, 8)])
def test_separable_path_of_the_general_kernel_is_bit_identical(oracle, W, H, D, V, monkeypatch):
    """Side cameras with the main camera's orientation but off its focal plane (moved along the optical axis too: not sweep_fx_rect's case)
    take the general kernel's separable path -- reciprocal per (view, plane) and the row part of the LDS addresses per (view, row, plane)
    from tables, the column part shared by a thread's rows (sweep_fx.hip: plan_sep_tables / sample_view_sep).  Every cell, depth, cost and
    index equals the oracle's and the same kernel's with the path switched off; a rotated view among them takes the general form."""
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.2)
    side_cams = side_cams.copy()
    for v in range(V):
        a = 2.0 * np.pi * v / V
        side_cams[v] = synth.camera_at([0.2 * np.cos(a), 0.2 * np.sin(a), 0.11 * (v - 1)], W, H)       # v = 1 stays in the focal plane
    if V >= 5:
        side_cams[3] = _rot_cam(W, H, [0.1, -0.15, 0.05], 0.02, -0.015)                                  # not separable
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler="fixed")
    both = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN
    got = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, both, sampler="fixed")
    _check(got, ref, D)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    assert (got[3] >> 24).max() >= V - 1
    monkeypatch.setenv("MVS_NO_SEP", "1")
    plain = _gpu_sweep(W, H, main_cam, main_img, side_cams, sides, D, both, sampler="fixed")
    for a, b in zip(got, plain):
        np.testing.assert_array_equal(a, b)
    # fused-only launches, row bands and view subsets go through the same tables
    monkeypatch.delenv("MVS_NO_SEP")
    with mvs_amd.Context(W, H, sampler="fixed") as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        d, c, i, _ = ctx.sweep_fetch()
        np.testing.assert_array_equal(d, ref[0])
        np.testing.assert_array_equal(i, ref[2])
        ctx.sweep_run(1, V - 1, mvs_amd.MVS_SWEEP_VOLUME)
        vol = ctx.sweep_fetch(want_volume=True)[3]
    sub = oracle.sweep(main_cam, main_img, side_cams[1:], sides[1:], D, want_volume=True, nthreads=8, sampler="fixed")
    np.testing.assert_array_equal(vol, sub[3])
