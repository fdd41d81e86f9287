"""world_size-2 gloo tests of the multi-GPU shardings (CPU): the oracle stands in for each rank's GPU sweep, the
collective and the bookkeeping are the product's (mvs_amd.dist)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, tmp):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    from mvs_amd import dist as mdist
    from mvs_amd import synth
    o = orc.load()
    W, H, D, V = 64, 40, 12, 5
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3, freq_scale=0.3)
    if mode == "views":
        v0, vn = mdist.view_shard(V, rank, world)
        _, _, _, part = o.sweep(main_cam, main_img, side_cams[v0:v0 + vn], sides[v0:v0 + vn], D, want_volume=True)
        t = torch.from_numpy(part.astype(np.int64))  # gloo has no uint32; RCCL path uses int32 on the same bits
        mdist.allreduce_volume(dist, t)
        vol = t.numpy().astype(np.uint32)
        depth, cost, idx = o.argmin(vol, o.plane_table(D, -1.0, 1.0))
        np.savez(os.path.join(tmp, "views_%d.npz" % rank), vol=vol, depth=depth, idx=idx)
    elif mode == "rows":
        # each rank owns one band of rows of the SAME main view (the oracle's rows stand in for mvs_sweep_run_rows)
        bands = mdist.row_bands(H, world, 16)
        r0, rn = bands[rank]
        full = o.sweep(main_cam, main_img, side_cams, sides, D)[0]
        local = torch.from_numpy(np.ascontiguousarray(full[r0:r0 + rn]))
        depth = mdist.gather_rows(dist, torch, local, bands, W).numpy()
        np.savez(os.path.join(tmp, "rows_%d.npz" % rank), depth=depth, r0=r0, rn=rn)
    else:
        frames = 5
        mine = mdist.frame_shard(frames, rank, world)
        local = []
        for f in mine:
            mc, mi, sc, si, _ = synth.make_views(W, H, 2, radius=0.3, freq_scale=0.3, seed=synth.SEED_SCENE + f)
            local.append(o.sweep(mc, mi, sc, si, D)[0])
        allres = mdist.gather_frames(dist, local, frames, rank, world)
        np.savez(os.path.join(tmp, "frames_%d.npz" % rank), depths=np.stack(allres))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["views", "rows", "frames"])
def test_two_rank_sharding(oracle, tmp_path, mode):
    from mvs_amd import synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    W, H, D, V = 64, 40, 12, 5
    if mode == "views":
        main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3, freq_scale=0.3)
        d_ref, _, i_ref, v_ref = oracle.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True)
        for r in range(world):
            g = np.load(tmp_path / ("views_%d.npz" % r))
            np.testing.assert_array_equal(g["vol"], v_ref)      # the all-reduced shards == single-process volume
            np.testing.assert_array_equal(g["depth"], d_ref)
            np.testing.assert_array_equal(g["idx"], i_ref)
    elif mode == "rows":
        main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3, freq_scale=0.3)
        d_ref = oracle.sweep(main_cam, main_img, side_cams, sides, D)[0]
        heights = []
        for r in range(world):
            g = np.load(tmp_path / ("rows_%d.npz" % r))
            np.testing.assert_array_equal(g["depth"], d_ref)    # unequal bands (32 + 8 rows) reassemble exactly
            heights.append(int(g["rn"]))
        assert heights == [32, 8]
    else:
        ref = []
        for f in range(5):
            mc, mi, sc, si, _ = synth.make_views(W, H, 2, radius=0.3, freq_scale=0.3, seed=synth.SEED_SCENE + f)
            ref.append(oracle.sweep(mc, mi, sc, si, D)[0])
        for r in range(world):
            np.testing.assert_array_equal(np.load(tmp_path / ("frames_%d.npz" % r))["depths"], np.stack(ref))


def test_shard_bookkeeping():
    from mvs_amd import dist as mdist
    for V in (0, 1, 5, 16, 17):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                a, n = mdist.view_shard(V, r, world)
                cover += list(range(a, a + n))
            assert cover == list(range(V))
    assert mdist.frame_shard(5, 1, 2) == [1, 3] and mdist.frame_shard(2, 3, 8) == []
    for H in (1, 16, 40, 1080, 2160):
        for world in (1, 2, 3, 8, 100):
            bands = mdist.row_bands(H, world, 16)
            assert len(bands) == world and bands[0][0] == 0
            assert all(a % 16 == 0 or n == 0 for a, n in bands)
            cover = [r for a, n in bands for r in range(a, a + n)]
            assert cover == list(range(H))
            sizes = [n for _, n in bands if n]
            assert max(sizes) - min(sizes) <= 16 + 15      # balanced to one unit (the last band may be ragged)
    assert mdist.row_bands(1080, 8, 16) == [(0, 144), (144, 144), (288, 144), (432, 144), (576, 128), (704, 128), (832, 128), (960, 120)]


def test_equal_row_bands_are_contiguous_at_a_fixed_stride():
    """bench.py's rows sharding gathers the bands at a fixed stride and reads the depth map from the gathered buffer: band r must
    start at r * tallest, only the last non-empty band may be shorter, and the tallest band is no taller than row_bands' tallest"""
    from mvs_amd import dist as mdist
    for H in (1, 16, 40, 480, 1080, 2160):
        for world in (1, 2, 3, 4, 8, 100):
            bands = mdist.equal_row_bands(H, world, 16)
            tallest = max(n for _, n in bands)
            assert len(bands) == world and (tallest % 16 == 0 or tallest == H)
            assert [r for a, n in bands for r in range(a, a + n)] == list(range(H))
            for r, (a, n) in enumerate(bands):
                assert a == min(H, r * tallest) and (n == tallest or a + n == H)
            assert tallest <= max(n for _, n in mdist.row_bands(H, world, 16)) + 15   # same number of units; only a ragged last band differs
    assert mdist.equal_row_bands(1080, 8, 16) == [(0, 144)] + [(144 * r, 144) for r in range(1, 7)] + [(1008, 72)]
