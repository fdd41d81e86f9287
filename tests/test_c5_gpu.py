"""BASELINE config 5 on the GPU, against the oracle, frame by frame (tracks/zatisi.yaml, 128 planes, 4 neighbours per main
frame): every 4th main frame (30 of 120) through mvs_sweep AND mvs_process_frame, single process vs the oracle (whole arrays),
then the same frames sharded over two ranks with the host gather of the point blocks in frame order (recon.cpp:65-117) --
checksums must equal the single-process run.  Out of scope and absent: video decoding, CGAL meshing."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import c5_common
import mvs_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c5_sequence_frame_by_frame_and_two_rank_gather(oracle):
    seq = c5_common.Sequence()
    W, H = seq.W, seq.H
    assert len(seq.mains) >= 30 and seq.n == 120
    nthreads = min(64, os.cpu_count() or 8)
    soup = oracle.load_mesh(seq.verts, seq.faces)
    single = []
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(seq.verts, seq.faces)
        for f in seq.mains:
            ids = seq.sides(f)
            cams = np.stack([seq.cams[j] for j in ids])
            frames = [seq.frame(j) for j in ids]
            main_img = seq.frame(f)
            depth, cost, pts = c5_common.process_main_frame(ctx, seq, f)
            # (a) the D-plane sweep: depth, cost against the oracle's restatement of the same (fixed) sampler; RMSE tolerance of
            #     north_star is 1e-4 -- measured 0
            d_ref, c_ref, i_ref, _ = oracle.sweep(seq.cams[f], main_img, cams, frames, c5_common.PLANES, nthreads=nthreads, sampler="fixed")
            np.testing.assert_array_equal(depth, d_ref)
            np.testing.assert_array_equal(cost, c_ref)
            assert (depth != 1.0).mean() > 0.5, "most pixels of a zatisi main frame see at least one neighbour"
            # (b) the reference's per-frame stage on the proxy mesh
            d = oracle.depth(soup, seq.cams[f], W, H)
            flows = []
            for cam, img in zip(cams, frames):
                mixed, d = oracle.mix_background(oracle.projected(soup, seq.cams[f], img, cam), main_img, d)
                flows.append(oracle.calculate_flow(main_img, mixed, False))
            ref = oracle.triangulate_pixels(flows, seq.cams[f], cams, d)
            assert pts.shape == ref.shape
            np.testing.assert_array_equal(pts[:, :4], ref[:, :4])
            np.testing.assert_allclose(pts[:, 4:], ref[:, 4:], rtol=1e-5, atol=1e-6)
            single.append({"main": f, "depth_crc": c5_common.crc(depth), "points": int(pts.shape[0]), "points_crc": c5_common.crc(pts[:, :4])})
    assert sum(s["points"] for s in single) > 10000

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "c5_worker.py")]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    two = json.loads(lines[0])
    assert two["frames"] == single, "frame-sharded run differs from the single-process run"
    assert two["cloud_points"] == sum(s["points"] for s in single)


def test_c5_batched_resident_entry_equals_one_call_path_on_all_120_frames():
    """the frame store + mvs_sweep_batch (frames uploaded once, several main frames per launch) against the one-call mvs_sweep, every
    one of the 120 zatisi main frames, depth and best cost bit for bit; batches of 1, 7 and 16 main frames"""
    seq = c5_common.Sequence()
    W, H, n = seq.W, seq.H, seq.n
    frames = [seq.frame(f) for f in range(n)]
    with mvs_amd.Context(W, H) as ctx:
        ref = {}
        for f in range(n):
            ids = seq.sides(f)
            d, c = ctx.sweep(seq.cams[f], frames[f], np.stack([seq.cams[j] for j in ids]), [frames[j] for j in ids], c5_common.PLANES, want_cost=True)
            ref[f] = (c5_common.crc(d), c5_common.crc(c))
    with mvs_amd.Context(W, H) as ctx:
        ctx.frame_store(n)
        for f in range(n):
            ctx.frame_upload(f, frames[f])
        done = 0
        for batch in (1, 7, 16):
            while done < n and (batch == 16 or done < {1: 3, 7: 3 + 28}[batch]):
                mains = list(range(done, min(n, done + batch)))
                side_slots = np.array([seq.sides(f) for f in mains], np.int32)
                depth, cost = ctx.sweep_batch(mains, np.stack([seq.cams[f] for f in mains]), side_slots,
                                              np.stack([np.stack([seq.cams[j] for j in seq.sides(f)]) for f in mains]), c5_common.PLANES, want_cost=True)
                for k, f in enumerate(mains):
                    assert (c5_common.crc(depth[k]), c5_common.crc(cost[k])) == ref[f], "main frame %d (batch of %d)" % (f, len(mains))
                done += len(mains)
        assert done == n
        # the asynchronous form: three batches queued back to back (two in flight, the third call waits for the oldest), each into
        # its own page-locked buffer, complete after mvs_sweep_batch_wait; then a synchronous call on the same context
        bufs = [(mvs_amd.pinned_array((8, H, W), np.float32), mvs_amd.pinned_array((8, H, W), np.float32)) for _ in range(3)]
        for rnd in range(2):
            groups = [list(range(rnd * 24 + 8 * k, rnd * 24 + 8 * k + 8)) for k in range(3)]
            for (dbuf, cbuf), mains in zip(bufs, groups):
                ctx.sweep_batch_async(mains, np.stack([seq.cams[f] for f in mains]), np.array([seq.sides(f) for f in mains], np.int32),
                                      np.stack([np.stack([seq.cams[j] for j in seq.sides(f)]) for f in mains]), c5_common.PLANES, out=dbuf, cost_out=cbuf)
            ctx.sweep_batch_wait()
            for (dbuf, cbuf), mains in zip(bufs, groups):
                for k, f in enumerate(mains):
                    assert (c5_common.crc(dbuf[k]), c5_common.crc(cbuf[k])) == ref[f], "main frame %d (asynchronous batch)" % f
        ctx.sweep_batch_wait()   # nothing in flight: returns at once
        d = ctx.sweep_batch([5], seq.cams[5:6], np.array([seq.sides(5)], np.int32), np.stack([np.stack([seq.cams[j] for j in seq.sides(5)])]), c5_common.PLANES)
        assert c5_common.crc(d[0]) == ref[5][0]
        # argument checks: an empty slot, a slot outside the store
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_batch([n + 5], seq.cams[:1], np.array([[0, 1, 2, 3]], np.int32), np.stack([seq.cams[:4]]), c5_common.PLANES)
        ctx.frame_store(4)   # re-sizing empties the store
        with pytest.raises(mvs_amd.MvsError):
            ctx.sweep_batch([0], seq.cams[:1], np.array([[0, 1, 2, 3]], np.int32), np.stack([seq.cams[:4]]), c5_common.PLANES)
