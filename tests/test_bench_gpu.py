"""bench.py's multi-rank step logic on a one-GPU box: two ranks share GPU 0 and talk over gloo (MVS_BENCH_SAME_DEVICE
test hook; RCCL refuses two ranks on one device).  The strong-scaling shardings must reproduce the single-GPU depth map
bit for bit: `views` through the integer all-reduce of the packed volume, `rows` through the all-gather of depth bands."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, world):
    env = dict(os.environ)
    cmd = [sys.executable]
    if world > 1:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env["MVS_BENCH_SAME_DEVICE"] = "1"
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", "c1", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline"] + (["--no-extras"] if "--with-extras" not in extra else []) + [e for e in extra if e != "--with-extras"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_and_two_rank_shardings():
    single = _bench([], 1)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in single, key
    assert single["n_gpus"] == 1 and single["vs_baseline"] is None and single["depth_check"] is True
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in single["roofline"], key
    # the SURVEY 8d ring is rectified: the fixed sampler's plan is served by sweep_fx_rect; N = 1 lines carry no scaling / shard
    assert single["config"]["sampler"] == "fixed" and single["roofline"]["kernel"] == "sweep_fx_rect"
    assert single["scaling"] is None and single["config"]["shard"] is None
    assert "one_call_mvs_sweep" not in single   # --no-extras
    # the default N > 1 line: strong scaling of one main view by row bands, the north_star's view shardings timed beside it
    dflt = _bench([], 2)
    assert dflt["n_gpus"] == 2 and dflt["scaling"] == "strong" and dflt["config"]["shard"] == "rows"
    assert dflt["depth_crc32"] == single["depth_crc32"] == dflt["depth_crc32_single_gpu"]
    # the self-verifying block of an N > 1 line: the group's size as a collective saw it, one identity per rank (here: the SAME device twice -- the hook)
    chk = dflt["multi_gpu_check"]
    assert chk["rccl_ranks"] == 2 and chk["backend"] == "gloo" and len(chk["devices"]) == 2 and chk["distinct_devices"] == 1
    assert chk["peer_access_to_rank0"] == [True, True] and "c4_rows" not in dflt   # (--no-extras)
    rs = _bench(["--shard", "views", "--collective", "reduce_scatter"], 2)   # partial selection per plane slice + merge
    assert rs["depth_crc32"] == single["depth_crc32"] and rs["config"]["collective"] == "reduce_scatter"
    for shard, scaling in (("views", "strong"), ("rows", "strong"), ("frames", "weak")):
        two = _bench(["--shard", shard], 2)
        assert two["n_gpus"] == 2 and two["scaling"] == scaling and two["depth_check"] is True
        assert "TEST HOOK" in two["data"]
        assert two["depth_crc32"] == single["depth_crc32"], shard   # rank 0's frame is the single-GPU frame in every mode
    full = _bench(["--with-extras"], 2)   # default N = 2 with the alternatives block
    alt = full["config"]["alternatives"]
    assert set(alt) == {"views_allreduce", "views_reduce_scatter", "frames_weak"}
    assert alt["frames_weak"]["scaling"] == "weak" and alt["frames_weak"]["samples_per_s"] > 0 and alt["frames_weak"]["collective_bytes_per_rank_per_step"] == 0
    for name, a in alt.items():
        if name == "frames_weak":
            continue
        assert a["depth_crc32"] == single["depth_crc32"] and a["collective_bytes_per_rank_per_step"] > 0 and a["views_per_rank"] == 2
    # ... and the same GPUs once more through the product's own multi-GPU entry (rank 0's child process: mvs_comm_set_* + mvs_comm_run)
    vc = full["via_comm"]
    assert "error" not in vc, vc
    assert vc["depth_crc32"] == single["depth_crc32"] and vc["ms_per_step"] > 0 and "mvs_comm_run" in vc["entry"] and "TEST HOOK" in vc["data"]
    assert set(vc["modes"]) == {"rows", "rows_async", "views", "views_scatter"} and all(m["depth_crc32"] == single["depth_crc32"] for m in vc["modes"].values())
    assert vc["peer_access"] == [1, 1] and vc["devices"] == [0, 0] and vc["modes"]["rows_async"]["in_flight"] == 2
    # BASELINE's config 4 in rows mode beside the headline: same step, its own single-GPU reference, depth map reproduced
    c4 = full["c4_rows"]
    assert "error" not in c4, c4
    assert c4["workload"].startswith("c4: 3840x2160, 256 planes, 32") and c4["depth_crc32"] == c4["depth_crc32_single_gpu"]
    assert c4["ms_per_step"] > 0 and c4["single_gpu_ms_per_step"] > 0 and c4["speedup_vs_single_gpu_in_process"] > 0 and sum(c4["rows_per_rank"]) == 2160
    for s in ("exact",):
        ex = _bench(["--sampler", s], 1)
        assert ex["roofline"]["kernel"] in ("sweep_tiled", "sweep_exact_rect") and ex["depth_check"] is True   # (the ring is rectified: sweep_exact_rect where its boxes fit the LDS slots)
    # the N = 1 line with its extras: the exact sampler, what the two samplers disagree on, general cameras, a >= 2 s sustained run
    ext = _bench(["--with-extras"], 1)
    assert ext["exact_sampler"]["sampler"] == "exact" and ext["exact_sampler"]["ms_per_step"] > 0
    dis = ext["sampler_disagreement"]
    # (c1: 32 planes, 4 views -- coarser than the c2 / c3 cases tests/test_sampler_tolerance_gpu.py holds to the stated tolerance)
    assert dis["plane_flip_rate"] <= 0.06 and dis["plane_flip_rate_more_than_one_plane"] <= 0.003 and dis["mean_abs_best_cost_difference_grey_levels"] <= 0.2
    assert ext["general_camera_path"]["kernel"] == "sweep_fx_tiled" and ext["general_camera_path"]["ms_per_step"] > 0
    assert ext["sustained"]["seconds"] >= 1.9 and ext["sustained"]["steps"] >= 300
    # a new (main, views) set from raw frames resident in HBM: prepared, planned and swept inside the timed step, same depth map
    cold = ext["cold_step"]
    assert cold["ms"] >= ext["ms_per_step"] * 0.9 and cold["depth_crc32"] == ext["depth_crc32_single_gpu"] and 0.0 < cold["frac"] < 1.0
    assert 0.0 < cold["same_cameras_ms"] <= cold["ms"] * 1.2   # (the plan reused: not slower than planning, up to noise)
    assert "combine_best" in ext["roofline"]["ms_per_launch_covers"]
    assert "roofline_traffic" in ext["general_camera_path"]
    # side cameras that are translations of the main one (any direction): the same kernel's separable path, faster than its general form
    assert 0 < ext["general_camera_path"]["translation_only_cameras_ms_per_step"] < ext["general_camera_path"]["ms_per_step"]
    # the reference's own per-frame stage: one context's latency and four contexts' throughput on one GPU, every thread's points equal the single context's
    for alg in ("variational", "farneback"):
        rs = ext["reference_stage"][alg]
        assert rs["thread_results_equal_single_context"] is True and rs["points_per_main_frame"] > 50000
        assert 0 < rs["ms_per_main_frame_4_contexts_on_threads"] < rs["ms_per_main_frame_one_context"]
        # with the sequence's frames in the frame store (mvs_process_frame_slots; CRC-checked against the host-frame calls inside bench.py)
        assert 0 < rs["ms_per_main_frame_one_context_frame_store"] < 1.1 * rs["ms_per_main_frame_one_context"]
        assert 0 < rs["ms_per_main_frame_4_contexts_frame_store"] < rs["ms_per_main_frame_one_context"]
    # the flow block: Farneback returns the pair's known shift at every size it reports (round 5 timed a pair it returned nothing on)
    for name, f in ext["flow"].items():
        if name.startswith("farneback"):
            assert f["recovered_fraction"] >= 0.70 and f["epe_vs_known_shift"] < 0.35 * (f["known_shift_px"][0] ** 2 + f["known_shift_px"][1] ** 2) ** 0.5, (name, f)
            assert 0 < f["fused_floor_bytes"] < f["algorithmic_bytes"] and 0 < f["frac_fused_floor"] < f["frac"] < 1
        elif name.startswith("variational"):
            assert 0 < f["recovered_fraction"] < 1 and f["device_ms"] > 0


def test_via_comm_line():
    """bench.py --via-comm: ONE process drives N GPUs through mvs_comm_set_* / mvs_comm_run (here: 3 ranks on the one GPU, loopback
    collectives, labelled as the test hook it is); the line keeps the contract and every mode reproduces the single-GPU depth map"""
    env = dict(os.environ, MVS_BENCH_SAME_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--via-comm", "--gpus", "3", "--config", "c1", "--steps", "3", "--warmup", "1"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["n_gpus"] == 3 and rec["scaling"] == "strong" and "TEST HOOK" in rec["data"] and rec["depth_check"] is True
    assert rec["depth_crc32"] == rec["depth_crc32_single_gpu"]
    modes = rec["config"]["modes"]
    assert set(modes) == {"rows", "rows_async", "views", "views_scatter"} and all(m["depth_crc32"] == rec["depth_crc32"] for m in modes.values())
    assert rec["config"]["peer_access"] == [1, 1, 1] and rec["config"]["devices"] == [0, 0, 0] and "no error" in rec["config"]["peer_access_note"]
