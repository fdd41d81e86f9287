"""mvs_sweep_sharded with n = 2, 4, 8 ranks on the ONE GPU of a test box.  RCCL refuses two ranks per device, so
the seven collective entry points mvs_comm resolves with dlopen come from tests/loopback_rccl (a thread barrier + kernels between
same-device buffers, selected through the MVS_RCCL_LIBRARY hook) and MVS_COMM_ALLOW_SAME_DEVICE=1 lifts the one-rank-per-GPU check.
What this executes for the first time with more than one rank: the host barrier and the shared error flag, the plane-group all-reduce
pipeline on the second stream, reduce-scatter slicing + partial selection + all-gather + merge, row bands, view shards that are empty
(more ranks than views), and the abort path.  It measures nothing: no scaling curve exists until a multi-GPU node runs the real library
(tests/test_comm_gpu.py keeps those cases, skipping here)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import mvs_amd
from mvs_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOPBACK = os.path.join(ROOT, "tests", "loopback_rccl", "_build", "libloopback_rccl.so")


@pytest.fixture
def loopback(monkeypatch):
    assert os.path.exists(LOOPBACK), "build it: make -C tests/loopback_rccl (or __graft_entry__.build())"
    monkeypatch.setenv("MVS_RCCL_LIBRARY", LOOPBACK)
    monkeypatch.setenv("MVS_COMM_ALLOW_SAME_DEVICE", "1")


_SCENES = {}


def _scene(W, H, D, V):
    key = (W, H, D, V)
    if key not in _SCENES:
        main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.2)
        with mvs_amd.Context(W, H) as ctx:
            d1, c1 = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
        _SCENES[key] = (main_cam, main_img, side_cams, sides, d1, c1)
    return _SCENES[key]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n", [2, 4, 8])
def test_every_mode_with_n_ranks_equals_the_single_gpu_sweep_at_c2_size(loopback, n):
    """c2's size (1280 x 720, 64 planes, 8 views): depth and best cost bit-identical to mvs_sweep in rows / views x plane groups {1, 4} /
    views_scatter"""
    W, H, D, V = 1280, 720, 64, 8
    main_cam, main_img, side_cams, sides, d1, c1 = _scene(W, H, D, V)
    with mvs_amd.Comm([0] * n, W, H) as comm:
        assert comm.size() == n
        for mode, groups in (("rows", None), ("views", 1), ("views", 4), ("views_scatter", None)):
            comm.set_mode(mode, groups)
            d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
            np.testing.assert_array_equal(d, d1, err_msg="%s n=%d" % (mode, n))
            np.testing.assert_array_equal(c, c1, err_msg="%s n=%d" % (mode, n))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sampler", ["fixed", "exact"])
def test_ragged_shards(loopback, sampler):
    """more ranks than views (empty view shards), a plane count that is no multiple of the ranks (scatter falls back to the all-reduce
    pipeline), a height that leaves the last row band short and some ranks without rows, both samplers"""
    W, H, D, V = 320, 72, 40, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        d1, c1 = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
        d0, c0 = ctx.sweep(main_cam, main_img, side_cams[:0], [], D, want_cost=True)
    for n in (2, 4, 8):
        with mvs_amd.Comm([0] * n, W, H, sampler=sampler) as comm:
            for mode, groups in (("rows", None), ("views", 2), ("views", 64), ("views_scatter", None)):
                comm.set_mode(mode, groups)
                d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
                np.testing.assert_array_equal(d, d1, err_msg="%s n=%d" % (mode, n))
                np.testing.assert_array_equal(c, c1, err_msg="%s n=%d" % (mode, n))
                d, c = comm.sweep(main_cam, main_img, side_cams[:0], [], D)   # no views at all
                np.testing.assert_array_equal(d, d0)


def _worker(n, mode, groups, env_extra, timeout=300):
    env = dict(os.environ, MVS_RCCL_LIBRARY=LOOPBACK, MVS_COMM_ALLOW_SAME_DEVICE="1", **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_loopback_worker.py"), str(n), mode, str(groups)], env=env, capture_output=True, text=True,
                         timeout=timeout)   # a hang is a TimeoutExpired here: the failure this test exists to catch
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,groups", [("rows", 1), ("views", 4), ("views_scatter", 1)])
def test_a_rank_that_fails_locally_makes_every_rank_return(mode, groups):
    """one rank gives up in its local phase (as a failed device allocation would): nobody enters a collective, the call returns that rank's
    error within the timeout, and the communicator is still usable afterwards"""
    assert os.path.exists(LOOPBACK)
    r = _worker(4, mode, groups, {"MVS_COMM_TEST_FAIL_RANK": "2"})
    assert "rank 2" in r["first"] and "MVS_COMM_TEST_FAIL_RANK" in r["first"], r
    assert r["first_s"] < 60.0, r
    assert "rank 2" in r["second"], r   # the hook is still set: the same error again, not a hang and not MVS_ESTATE


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,groups,fail", [("views", 4, "allreduce:1:2"), ("views", 1, "allreduce:0:1"), ("views_scatter", 1, "reducescatter:3:1"),
                                              ("views_scatter", 1, "allgather:2:1")])
def test_a_failing_collective_aborts_every_rank(mode, groups, fail):
    """a collective returns an error on one rank while the others are waiting inside theirs: the abort path unblocks them, the call
    returns an error within the timeout, and the communicator refuses further work (MVS_ESTATE: create a new one)"""
    assert os.path.exists(LOOPBACK)
    r = _worker(4, mode, groups, {"LOOPBACK_RCCL_FAIL": fail})
    assert r["first"] != "ok" and "nccl" in r["first"], r
    assert r["first_s"] < 60.0, r
    assert "aborted" in r["second"], r


def test_rows_mode_needs_no_collective_library(monkeypatch):
    """the default mode exchanges nothing, so a communicator comes up and sweeps without any RCCL communicator being created
    (ncclCommInitAll is deferred to the first sweep that exchanges something): two ranks on one device -- which RCCL itself would refuse at
    init -- are fine in rows mode"""
    W, H, D, V = 320, 200, 32, 4
    main_cam, main_img, side_cams, sides, d1, c1 = _scene(W, H, D, V)
    monkeypatch.setenv("MVS_COMM_ALLOW_SAME_DEVICE", "1")
    monkeypatch.delenv("MVS_RCCL_LIBRARY", raising=False)
    with mvs_amd.Comm([0, 0], W, H) as comm:   # real RCCL loaded (torch has it), never initialised: two ranks on one device are fine in rows mode
        d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
        np.testing.assert_array_equal(d, d1)
        np.testing.assert_array_equal(c, c1)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n", [2, 8])
def test_the_resident_form_runs_every_mode_on_what_was_uploaded_once(loopback, n):
    """mvs_comm_set_planes / _set_main / _set_views once, then mvs_comm_run in every mode (persistent rank threads, nothing uploaded or
    planned per call) and mvs_comm_fetch: depth and best cost of the whole view, from rank 0's GPU, bit-identical to mvs_sweep -- rows mode's
    bands arrive there by peer copies.  Repeated runs, mode switches without a new upload, a new main view, a new view set."""
    W, H, D, V = 640, 360, 48, 6
    main_cam, main_img, side_cams, sides, d1, c1 = _scene(W, H, D, V)
    with mvs_amd.Comm([0] * n, W, H) as comm:
        with pytest.raises(mvs_amd.MvsError, match="set planes, main view and side views first"):
            comm.run()
        comm.set(main_cam, main_img, side_cams, sides, D)
        for mode, groups, flags in (("rows", None, 0), ("views", 3, 0), ("rows", None, mvs_amd.MVS_SWEEP_VOLUME), ("views_scatter", None, 0), ("views", 1, 0), ("rows", None, 0)):
            comm.set_mode(mode, groups)
            for _ in range(3):
                comm.run(flags)
            d, c = comm.fetch()
            np.testing.assert_array_equal(d, d1, err_msg="%s n=%d" % (mode, n))
            np.testing.assert_array_equal(c, c1, err_msg="%s n=%d" % (mode, n))
        # a new main view invalidates the side views (their matrices depend on the main camera), like a context
        lib = comm.lib
        cam = np.ascontiguousarray(main_cam, np.float32)
        img = np.ascontiguousarray(sides[0], np.uint8)
        assert lib.mvs_comm_set_main(comm.h, cam.ctypes.data_as(mvs_amd._fp), img.ctypes.data_as(mvs_amd._u8p)) == 0
        with pytest.raises(mvs_amd.MvsError, match="side views first"):
            comm.run()
        # another view set (the first side view as main frame, fewer views), swept resident, against a fresh single-GPU sweep
        comm.set(side_cams[0], sides[0], side_cams[1:4], sides[1:4], 32)
        with mvs_amd.Context(W, H) as ctx:
            d2, c2 = ctx.sweep(side_cams[0], sides[0], side_cams[1:4], sides[1:4], 32, want_cost=True)
        for mode in ("views_scatter", "rows"):
            comm.set_mode(mode)
            comm.run()
            d, c = comm.fetch()
            np.testing.assert_array_equal(d, d2)
            np.testing.assert_array_equal(c, c2)


def test_the_one_call_form_leaves_its_upload_resident_for_the_same_mode(loopback):
    """mvs_sweep_sharded in a views mode uploads only each rank's own views: mvs_comm_run afterwards works in that split and says what is
    missing in rows mode (which needs every view on every rank)"""
    W, H, D, V = 320, 200, 32, 4
    main_cam, main_img, side_cams, sides, d1, c1 = _scene(W, H, D, V)
    with mvs_amd.Comm([0] * 4, W, H) as comm:
        comm.set_mode("views", 2)
        d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
        np.testing.assert_array_equal(d, d1)
        comm.run()
        np.testing.assert_array_equal(comm.fetch()[0], d1)
        comm.set_mode("rows")
        with pytest.raises(mvs_amd.MvsError, match="mvs_comm_set_views"):
            comm.run()
        comm.sweep(main_cam, main_img, side_cams, sides, D)   # rows: all views everywhere
        comm.set_mode("views_scatter")
        comm.run()
        np.testing.assert_array_equal(comm.fetch()[0], d1)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("n", [2, 8])
def test_every_mode_at_c4_size(loopback, n):
    """BASELINE config 4's size -- 3840 x 2160, 256 planes, 32 views: an 8.49 GB packed volume per rank, 2.12 G cells, plane slices and
    plane groups at offsets past 4 GiB -- through all three modes, resident and one-call, bit-identical to mvs_sweep (the
    slice arithmetic of csrc/comm.cpp had only ever run at c2's size).  Noise frames: every cell matters, no depth meaning."""
    W, H, D, V = 3840, 2160, 256, 32
    main_cam, main_img, side_cams, sides = synth.noise_views(W, H, V)
    with mvs_amd.Context(W, H) as ctx:
        d1, c1 = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
    with mvs_amd.Comm([0] * n, W, H) as comm:
        comm.set(main_cam, main_img, side_cams, sides, D)
        for mode, groups in (("rows", None), ("views", 4), ("views_scatter", None)):
            comm.set_mode(mode, groups)
            comm.run(mvs_amd.MVS_SWEEP_VOLUME)
            d, c = comm.fetch()
            np.testing.assert_array_equal(d, d1, err_msg="%s n=%d" % (mode, n))
            np.testing.assert_array_equal(c, c1, err_msg="%s n=%d" % (mode, n))
        comm.set_mode("views", 3)   # plane groups of 96, 96, 64 planes: the last group starts past 6 GiB
        d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
        np.testing.assert_array_equal(d, d1)
        np.testing.assert_array_equal(c, c1)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n", [1, 2, 8])
def test_two_calls_in_flight_keep_their_results_apart(loopback, n, monkeypatch):
    """mvs_comm_run_async / mvs_comm_wait (rows mode): view A queued, view B uploaded and queued while A may still be travelling, then the
    waits publish A and B in order -- each equal to the single-GPU sweep of its own view (two result pairs on rank 0, a staging copy per rank
    and slot).  Under the poison hook every fresh allocation is 0xFF-filled on its stream: a fill that could land after a band would show."""
    monkeypatch.setenv("MVS_POISON_ALLOC", "1")
    W, H, D, V = 640, 360, 48, 6
    main_cam, main_img, side_cams, sides, dA, cA = _scene(W, H, D, V)
    with mvs_amd.Context(W, H) as ctx:
        dB, cB = ctx.sweep(side_cams[0], sides[0], side_cams[1:5], sides[1:5], D, want_cost=True)
    with mvs_amd.Comm([0] * n, W, H) as comm:
        assert comm.peer_access() == [1] * n and comm.devices() == [0] * n and comm.note() == "no error"
        comm.set(main_cam, main_img, side_cams, sides, D)
        with pytest.raises(mvs_amd.MvsError, match="no call in flight"):
            comm.wait()
        comm.run_async()
        comm.run_async()
        assert comm.pending() == 2
        with pytest.raises(mvs_amd.MvsError, match="two calls are in flight"):
            comm.run_async()
        with pytest.raises(mvs_amd.MvsError, match="in flight"):
            comm.run()
        with pytest.raises(mvs_amd.MvsError, match="in flight"):
            comm.set(main_cam, main_img, side_cams, sides, D)
        comm.wait()
        d, c = comm.fetch()
        np.testing.assert_array_equal(d, dA)
        np.testing.assert_array_equal(c, cA)
        comm.wait()
        assert comm.pending() == 0
        # A in flight; B uploaded (the upload waits for A's sweeps, not for A's bands), B queued; results published in order
        comm.run_async()
        comm.wait()
        comm.run_async()                                                      # A again, in flight while ...
        comm.wait()
        comm.set(side_cams[0], sides[0], side_cams[1:5], sides[1:5], D)       # ... new inputs: the old maps are no longer "the result"
        with pytest.raises(mvs_amd.MvsError, match="no result yet"):
            comm.fetch()
        comm.run_async()
        comm.run_async()
        comm.wait()
        d, c = comm.fetch()
        np.testing.assert_array_equal(d, dB)
        np.testing.assert_array_equal(c, cB)
        comm.wait()
        np.testing.assert_array_equal(comm.fetch()[0], dB)
        # many calls back to back, two in flight throughout
        comm.run_async()
        for _ in range(20):
            comm.run_async()
            comm.wait()
        comm.wait()
        np.testing.assert_array_equal(comm.fetch()[0], dB)
        # the synchronous call still works afterwards and publishes rank 0's context maps
        comm.run()
        np.testing.assert_array_equal(comm.fetch()[0], dB)
        # a view-sharded mode completes inside run_async: one in flight, wait publishes it
        if n > 1:
            comm.set_mode("views", 2)
            comm.run_async()
            with pytest.raises(mvs_amd.MvsError, match="only MVS_SHARD_ROWS"):
                comm.run_async()
            comm.wait()
            d, c = comm.fetch()
            np.testing.assert_array_equal(d, dB)
            np.testing.assert_array_equal(c, cB)


def test_a_sync_resident_run_under_the_poison_hook(loopback, monkeypatch):
    """rank 0's maps are allocated -- and, under MVS_POISON_ALLOC, filled -- before the ranks meet, so no whole-map fill can land
    on rank 0's stream behind another rank's band copy on the first resident run"""
    monkeypatch.setenv("MVS_POISON_ALLOC", "1")
    W, H, D, V = 640, 360, 48, 6
    main_cam, main_img, side_cams, sides, dA, cA = _scene(W, H, D, V)
    for _ in range(3):
        with mvs_amd.Comm([0] * 8, W, H) as comm:
            comm.set(main_cam, main_img, side_cams, sides, D)
            comm.run()
            d, c = comm.fetch()
            np.testing.assert_array_equal(d, dA)
            np.testing.assert_array_equal(c, cA)


def test_two_calls_in_flight_with_ranks_that_have_no_rows(loopback):
    """72 rows on 8 ranks: the row bands are two tile rows each, ranks 5-7 sweep nothing and send nothing -- the queue, the waits and the published
    maps must not depend on every rank taking part; both samplers' row granularities (8 and 16 rows)"""
    W, H, D, V = 320, 72, 40, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    for sampler in ("fixed", "exact"):
        with mvs_amd.Context(W, H, sampler=sampler) as ctx:
            d1, c1 = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
        with mvs_amd.Comm([0] * 8, W, H, sampler=sampler) as comm:
            comm.set(main_cam, main_img, side_cams, sides, D)
            comm.run_async()
            comm.run_async()
            for _ in range(2):
                comm.wait()
                d, c = comm.fetch()
                np.testing.assert_array_equal(d, d1, err_msg=sampler)
                np.testing.assert_array_equal(c, c1, err_msg=sampler)
            for _ in range(5):
                comm.run_async()
                comm.wait()
            np.testing.assert_array_equal(comm.fetch()[0], d1)
