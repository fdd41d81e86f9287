"""mvs_comm / mvs_sweep_sharded (the RCCL collective behind the C ABI): the single-device path on a one-GPU box -- RCCL with one
rank, reduce-scatter + partial selection + all-gather + merge, and the all-reduce fallback -- must equal mvs_sweep bit for bit;
with two or more visible GPUs the same is checked across devices (skipped otherwise: RCCL refuses two ranks per device)."""
import os

import numpy as np
import pytest

import mvs_amd
from mvs_amd import synth

pytestmark = pytest.mark.gpu


def _scene(W=320, H=200, D=32, V=5):
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    return W, H, D, V, main_cam, main_img, side_cams, sides


@pytest.mark.parametrize("sampler", ["fixed", "exact"])
def test_single_device_comm_equals_one_call_sweep(oracle, sampler, monkeypatch):
    W, H, D, V, main_cam, main_img, side_cams, sides = _scene()
    with mvs_amd.Context(W, H, sampler=sampler) as ctx:
        d1, c1 = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
    d_ref, c_ref, _, _ = oracle.sweep(main_cam, main_img, side_cams, sides, D, nthreads=8, sampler=sampler)
    np.testing.assert_array_equal(d1, d_ref)
    with mvs_amd.Comm([0], W, H, sampler=sampler) as comm:
        assert comm.size() == 1 and comm.mode() == mvs_amd.MVS_SHARD_ROWS
        for mode, groups in (("rows", None), ("views", 1), ("views", 2), ("views", 64), ("views_scatter", None)):
            comm.set_mode(mode, groups)   # row bands / all-reduce pipeline by plane groups / reduce-scatter + partial selection + all-gather + merge
            d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
            np.testing.assert_array_equal(d, d1)
            np.testing.assert_array_equal(c, c1)
        with pytest.raises(mvs_amd.MvsError):
            comm.set_mode(7)
        with pytest.raises(mvs_amd.MvsError):
            comm.set_mode("views", 0)
        comm.set_mode("views_scatter")
        d, c = comm.sweep(main_cam, main_img, side_cams[:2], sides[:2], 7)   # other sizes on the same communicator
        with mvs_amd.Context(W, H, sampler=sampler) as ctx:
            d7, c7 = ctx.sweep(main_cam, main_img, side_cams[:2], sides[:2], 7, want_cost=True)
        np.testing.assert_array_equal(d, d7)
        np.testing.assert_array_equal(c, c7)
        d, _ = comm.sweep(main_cam, main_img, side_cams[:0], [], D)           # no views at all
        assert (d == 1.0).all()
        monkeypatch.setenv("MVS_COMM_ALLREDUCE", "1")                         # the fallback for plane counts not divisible by the ranks
        d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
        np.testing.assert_array_equal(d, d1)
        np.testing.assert_array_equal(c, c1)
        with pytest.raises(mvs_amd.MvsError):
            comm.sweep(main_cam, main_img, side_cams, sides, 0)
        # more views than a summed cell can count: refused before any rank enters a collective (a late refusal would hang the others)
        too_many = 256 if sampler == "fixed" else 257
        with pytest.raises(mvs_amd.MvsError, match="views"):
            comm.sweep(main_cam, main_img, np.repeat(side_cams[:1], too_many, 0), [sides[0]] * too_many, D)


def test_comm_create_errors_on_the_gpu_box():
    with pytest.raises(mvs_amd.MvsError):
        mvs_amd.Comm([99], 64, 48)            # no such device
    with pytest.raises(mvs_amd.MvsError):
        mvs_amd.Comm([0], 1, 1)               # bad size


def _gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two visible GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_multi_device_comm_equals_single_gpu(ndev):
    if _gpu_count() < ndev:
        pytest.skip("%d GPUs visible" % _gpu_count())
    W, H, D, V, main_cam, main_img, side_cams, sides = _scene(640, 480, 64, 16)
    with mvs_amd.Context(W, H) as ctx:
        d1, c1 = ctx.sweep(main_cam, main_img, side_cams, sides, D, want_cost=True)
    with mvs_amd.Comm(list(range(ndev)), W, H) as comm:
        for mode, groups in (("rows", None), ("views", 4), ("views", 1), ("views_scatter", None)):
            comm.set_mode(mode, groups)
            d, c = comm.sweep(main_cam, main_img, side_cams, sides, D)
            np.testing.assert_array_equal(d, d1)
            np.testing.assert_array_equal(c, c1)
        # the resident rows mode between REAL devices: peer access enabled both ways at creation, the bands travel GPU to GPU; two calls in flight
        assert comm.devices() == list(range(ndev)) and len(comm.peer_access()) == ndev and comm.peer_access()[0] == 1
        assert (comm.note() == "no error") == all(comm.peer_access()), comm.note()
        comm.set_mode("rows")
        comm.set(main_cam, main_img, side_cams, sides, D)
        comm.run()
        np.testing.assert_array_equal(comm.fetch()[0], d1)
        comm.run_async()
        comm.run_async()
        for _ in range(2):
            comm.wait()
            d, c = comm.fetch()
            np.testing.assert_array_equal(d, d1)
            np.testing.assert_array_equal(c, c1)
        comm.set_mode("views_scatter")
        d, c = comm.sweep(main_cam, main_img, side_cams, sides, D - 1)        # (scatter mode) all-reduce fallback
    with mvs_amd.Context(W, H) as ctx:
        d2, c2 = ctx.sweep(main_cam, main_img, side_cams, sides, D - 1, want_cost=True)
    np.testing.assert_array_equal(d, d2)
    np.testing.assert_array_equal(c, c2)


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two visible GPUs")
def test_bench_over_rccl_matches_single_gpu():
    """bench.py under torch.distributed.run with the nccl backend (= RCCL): every sharding reproduces the single-GPU depth CRC"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = min(2, _gpu_count())
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", str(n), "--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["depth_crc32"] == line["depth_crc32_single_gpu"]
    chk = line["multi_gpu_check"]   # N ranks on N DISTINCT devices, as RCCL itself saw the group
    assert chk["rccl_ranks"] == n and chk["backend"] == "nccl" and chk["distinct_devices"] == n
    assert line["c4_rows"]["depth_crc32"] == line["c4_rows"]["depth_crc32_single_gpu"]
    for alt in line["config"]["alternatives"].values():
        assert alt["depth_crc32"] == line["depth_crc32_single_gpu"]
