"""Frames that are already on the device: mvs_sweep_set_main_device / mvs_sweep_set_views_device (raw u8
frames in HBM -> quad images in one pass, no PCIe) and mvs_sweep_handles (one main view over the frame store: nothing uploaded, copied or
re-prepared) must give exactly what the host-pointer entries give -- depth, best cost, index and every cell of the volume -- and what the
oracle gives; the quad images written straight from the raw frames are checked at a ragged size (byte path) and at aligned ones (dword
path); the padded u8 frames rebuilt from the quad images serve the un-tiled kernels and the exact sampler."""
import numpy as np
import pytest
import torch

import mvs_amd
import tracks_yaml
from mvs_amd import synth

pytestmark = pytest.mark.gpu

BOTH = mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN


def _rotated(side_cams, W, H, V, radius):
    cams = []
    for vi in range(V):
        ang = 2.0 * np.pi * vi / max(V, 1)
        yaw, pitch = 0.012 * np.cos(ang), 0.012 * np.sin(ang)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        rot = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        cams.append(synth.camera_at([radius * np.cos(ang), radius * np.sin(ang), 0.0], W, H, rot=rot))
    return np.stack(cams)


@pytest.mark.parametrize("size", [(322, 241, 32, 5), (640, 360, 48, 6), (66, 10, 16, 2)])
@pytest.mark.parametrize("sampler", ["fixed", "exact"])
def test_device_frames_equal_host_frames_and_the_oracle(oracle, size, sampler):
    W, H, D, V = size
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.25)
    ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, nthreads=8, sampler=sampler)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, BOTH)
        host = ctx.sweep_fetch(want_volume=True)
    for a, b in zip(host, ref):
        np.testing.assert_array_equal(a, b)
    main_t = torch.as_tensor(main_img, device="cuda").contiguous()
    side_t = [torch.as_tensor(s, device="cuda").contiguous() for s in sides]
    torch.cuda.synchronize()
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        ctx.sweep_set_planes(D)
        ctx.sweep_set_main_device(main_cam, main_t.data_ptr())
        ctx.sweep_set_views_device(side_cams, [t.data_ptr() for t in side_t])
        for flags in (BOTH, mvs_amd.MVS_SWEEP_VOLUME | mvs_amd.MVS_SWEEP_FUSED_ARGMIN | mvs_amd.MVS_SWEEP_FORCE_GENERIC):   # tiled kernels; un-tiled (padded frames from the quads)
            ctx.sweep_run(0, V, flags)
            dev = ctx.sweep_fetch(want_volume=True)
            for a, b in zip(dev, host):
                np.testing.assert_array_equal(a, b)
        # a new set of views on the same context (fewer, other order), then back to host frames
        ctx.sweep_set_views_device(side_cams[::-1][:2], [t.data_ptr() for t in side_t[::-1][:2]])
        ctx.sweep_run(0, 2, BOTH)
        d2 = ctx.sweep_fetch()[0]
        ctx.sweep_set_views(side_cams[::-1][:2], sides[::-1][:2])
        ctx.sweep_run(0, 2, BOTH)
        np.testing.assert_array_equal(ctx.sweep_fetch()[0], d2)
    np.testing.assert_array_equal(d2, oracle.sweep(main_cam, main_img, side_cams[::-1][:2], sides[::-1][:2], D, nthreads=8, sampler=sampler)[0])


def _handles_case(W, H, D, V, main_cam, main_img, side_cams, sides, expect_shape=None):
    with mvs_amd.Context(W, H) as ctx:
        d1, c1 = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
    cap = V + 5
    order = [int(k) for k in np.random.default_rng(V).permutation(cap)[:V + 1]]     # slots in no particular order; slot != view index
    with mvs_amd.Context(W, H) as ctx:
        ctx.frame_store(cap)
        ctx.frame_upload(order[0], main_img)
        for v in range(V):
            ctx.frame_upload(order[1 + v], sides[v])
        d, c = ctx.sweep_handles(order[0], main_cam, order[1:], side_cams, D, want_cost=True)
        np.testing.assert_array_equal(d, d1)
        np.testing.assert_array_equal(c, c1)
        if expect_shape is not None:
            assert ctx.plan_shape() == expect_shape
        # the staged state refers to the store now: another run gives the same, with the volume too
        ctx.sweep_run(0, V, BOTH)
        d_again = ctx.sweep_fetch()[0]
        np.testing.assert_array_equal(d_again, d1)
        # a second main view over the same store (views swapped: the old main frame becomes a side view)
        d_b = ctx.sweep_handles(order[1], side_cams[0], [order[0]] + order[2:], np.concatenate([main_cam[None], side_cams[1:]]), D)
        with mvs_amd.Context(W, H) as ctx2:
            d_b1 = ctx2.sweep(side_cams[0], sides[0], np.concatenate([main_cam[None], side_cams[1:]]), [main_img] + list(sides[1:]), D)
        np.testing.assert_array_equal(d_b, d_b1)
    return d1


def test_handles_equal_mvs_sweep_at_c2_on_the_ring_and_with_rotated_cameras(oracle):
    W, H, D, V = 1280, 720, 64, 8
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.15)
    d = _handles_case(W, H, D, V, main_cam, main_img, side_cams, sides, expect_shape=4)          # rectified kernel
    np.testing.assert_array_equal(d, oracle.sweep(main_cam, main_img, side_cams, sides, D, nthreads=8, sampler="fixed")[0])
    _handles_case(W, H, D, V, main_cam, main_img, _rotated(side_cams, W, H, V, 0.15), sides, expect_shape=3)   # general kernel


def test_handles_equal_mvs_sweep_at_c3():
    W, H, D, V = 1920, 1080, 128, 16
    main_cam, main_img, side_cams, sides = synth.noise_views(W, H, V)
    _handles_case(W, H, D, V, main_cam, main_img, side_cams, sides, expect_shape=4)


@pytest.mark.parametrize("name", ["koberec.yaml", "zatisi.yaml", "koule-tr.yaml"])
def test_handles_on_track_cameras(oracle, name):
    t = tracks_yaml.load(name)
    W, H, cams = t["width"], t["height"], t["cameras"]
    n = len(cams)
    m = n // 2
    step = max(1, n // 12)
    ids = [m - 2 * step, m - step, m + step, m + 2 * step]
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(5)]
    side_cams = np.stack([cams[i] for i in ids])
    d = _handles_case(W, H, 128, 4, cams[m], frames[0], side_cams, frames[1:], expect_shape=3)
    np.testing.assert_array_equal(d, oracle.sweep(cams[m], frames[0], side_cams, frames[1:], 128, nthreads=8, sampler="fixed")[0])


def test_handles_errors_and_state():
    W, H, D, V = 320, 200, 32, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    with mvs_amd.Context(W, H) as ctx:
        with pytest.raises(mvs_amd.MvsError, match="holds no frame"):
            ctx.sweep_handles(0, main_cam, [1, 2, 3], side_cams, D)          # no store yet
        ctx.frame_store(4)
        ctx.frame_upload(0, main_img)
        for v in range(V - 1):
            ctx.frame_upload(1 + v, sides[v])
        with pytest.raises(mvs_amd.MvsError, match="slot 3 holds no frame"):
            ctx.sweep_handles(0, main_cam, [1, 2, 3], side_cams, D)
        ctx.frame_upload(3, sides[2])
        d = ctx.sweep_handles(0, main_cam, [1, 2, 3], side_cams, D)
        d0 = ctx.sweep_handles(0, main_cam, [], side_cams[:0], D)             # no side views: background everywhere
        assert (d0 == 1.0).all()
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_handles(0, main_cam, [1, 2, 3], side_cams, 0)
        ctx.set_sampler("exact")
        with pytest.raises(mvs_amd.MvsError, match="MVS_SAMPLER_FIXED"):
            ctx.sweep_handles(0, main_cam, [1, 2, 3], side_cams, D)
        ctx.set_sampler("fixed")
        np.testing.assert_array_equal(ctx.sweep_handles(0, main_cam, [1, 2, 3], side_cams, D), d)
        # the un-tiled kernel and the exact sampler need the padded frames, which store-resident views do not have: refused, not wrong
        with pytest.raises(mvs_amd.MvsError, match="frame-store"):
            ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN | mvs_amd.MVS_SWEEP_FORCE_GENERIC)
        # re-sizing the store drops the views that lived in it: a run must be refused, not read freed memory
        ctx.frame_store(16)
        with pytest.raises(mvs_amd.MvsError, match="set main view"):
            ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        # and the host-pointer path works on the same context afterwards
        np.testing.assert_array_equal(ctx.sweep(main_cam, main_img, side_cams, sides, D), d)


def test_a_fixed_rigs_next_frames_reuse_the_plan_and_new_cameras_do_not(oracle, monkeypatch):
    """the plan depends on cameras and planes, not on frames: new frames under the cameras of the last set must give what a fresh context gives
    (plan reused), and so must new cameras, new planes and a new main camera in between (plan remade) -- each against the oracle"""
    W, H, D, V = 320, 200, 32, 4
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.2)
    rng = np.random.default_rng(5)
    frames2 = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(V + 1)]
    rot = _rotated(side_cams, W, H, V, 0.2)

    def ref(mc, mi, sc, si, planes):
        return oracle.sweep(mc, mi, sc, si, planes, nthreads=8, sampler="fixed")[0]
    with mvs_amd.Context(W, H) as ctx:
        np.testing.assert_array_equal(ctx.sweep(main_cam, main_img, side_cams, sides, D), ref(main_cam, main_img, side_cams, sides, D))
        np.testing.assert_array_equal(ctx.sweep(main_cam, frames2[0], side_cams, frames2[1:], D), ref(main_cam, frames2[0], side_cams, frames2[1:], D))   # same rig, new frames
        np.testing.assert_array_equal(ctx.sweep(main_cam, frames2[0], rot, frames2[1:], D), ref(main_cam, frames2[0], rot, frames2[1:], D))               # new cameras: general kernel
        np.testing.assert_array_equal(ctx.sweep(main_cam, main_img, rot, sides, D), ref(main_cam, main_img, rot, sides, D))                                 # same (general) rig, new frames
        np.testing.assert_array_equal(ctx.sweep(main_cam, main_img, side_cams, sides, D + 7), ref(main_cam, main_img, side_cams, sides, D + 7))           # new planes
        np.testing.assert_array_equal(ctx.sweep(side_cams[0], sides[0], side_cams[1:], sides[1:], D + 7), ref(side_cams[0], sides[0], side_cams[1:], sides[1:], D + 7))   # new main camera
        # staged form: views set twice with the same cameras (the second time the plan is reused), then with the reuse switched off
        ctx.sweep_set(main_cam, main_img, side_cams, sides, D)
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        ctx.sweep_set_views(side_cams, frames2[1:])
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        a = ctx.sweep_fetch()[0]
        ctx.set_plan_cache(False)
        ctx.sweep_set_views(side_cams, frames2[1:])
        ctx.sweep_run(0, V, mvs_amd.MVS_SWEEP_FUSED_ARGMIN)
        np.testing.assert_array_equal(ctx.sweep_fetch()[0], a)
    np.testing.assert_array_equal(a, ref(main_cam, main_img, side_cams, frames2[1:], D))


def test_no_state_of_a_long_lived_context_leaks_into_a_result():
    """tests/perf/fuzz_api.py in short: random sequences of view sets (rectified, rotated, mixed), planes, sampler switches, MVS_SWEEP_NO_RECT, the
    one-call entry, frame-store handles and device-resident setters on ONE context -- every result equal to a fresh context's for the same inputs"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import fuzz_api
    assert sum(fuzz_api.run(seed, 30) for seed in (11, 12, 13)) == 0
