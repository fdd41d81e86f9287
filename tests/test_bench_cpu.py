"""bench.py helpers that need no GPU."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_traffic_picks_the_newest_profile_numerically(tmp_path, monkeypatch):
    """roofline.traffic comes from the committed PMC summaries; 'v12' must win over 'v9' (a plain string sort says otherwise)"""
    b = _bench()
    d = tmp_path / "profiles" / "r01"
    d.mkdir(parents=True)
    for tag, val in (("v9_split", 9.0), ("v12_final", 12.0), ("v3_linear", 3.0)):
        (d / ("pmc_c3_%s.json" % tag)).write_text(json.dumps({"kernels": {"sweep_tiled<2, 32, true, true>": {"hbm_bytes_per_launch": val}}}))
    d2 = tmp_path / "profiles" / "r02"
    d2.mkdir()
    (d2 / "pmc_c3_v1_first.json").write_text(json.dumps({"kernels": {"sweep_tiled<2, 32, true, true>": {"hbm_bytes_per_launch": 101.0}}}))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    got = b.pmc_traffic("sweep_tiled", "c3")
    assert got["bytes"] == 101.0 and got["source"].endswith("pmc_c3_v1_first.json")   # a later round beats any version of an earlier one
    (d2 / "pmc_c3_v1_first.json").unlink()
    assert b.pmc_traffic("sweep_tiled", "c3")["bytes"] == 12.0
    assert b.pmc_traffic("sweep_tiled", "c9") is None


def test_committed_profiles_resolve():
    """every sweep kernel resolves to the newest committed summary that names it -- "newest" is read off the directory listing, not
    pinned to a round number (a literal here broke HEAD at the end of round 3)"""
    import glob
    import re
    b = _bench()

    def key(path):
        m = re.search(r"profiles[/\\]r(\d+)[/\\]pmc_[^_]+_v(\d+)", path)
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    for kernel in ("sweep_tiled", "sweep_fx_tiled", "sweep_fx_rect"):       # exact sampler, fixed sampler general / rectified
        got = b.pmc_traffic(kernel, "c3")
        assert got is not None and got["bytes"] > 1.0e9 and "pmc_c3_v" in got["source"], kernel
        naming = []
        for path in glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_c3_*.json")):
            rec = json.load(open(path))
            if any(n.startswith(kernel) and "hbm_bytes_per_launch" in c for n, c in rec.get("kernels", {}).items()):
                naming.append(path)
        assert os.path.join(ROOT, got["source"]) == max(naming, key=key), kernel
        assert 0.3 < got["valu_utilisation"] < 1.05


def test_gpus_without_a_launcher_starts_the_launcher_as_a_child():
    """`python bench.py --gpus 2` without torch.distributed.run: one process drives one GPU, so bench.py starts the launcher itself --
    as a child process, before torch is imported or a GPU touched -- and relays its output and exit code (the first 8-GPU
    node must yield a curve, not a usage error).  Here there is no GPU: both ranks come up under the launcher, say so, and the launcher's
    non-zero exit code is what this process returns."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c1", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert "--nproc-per-node 2" in out.stderr and "starting" in out.stderr            # the launcher was started by bench.py itself ...
    assert out.stderr.count("bench.py needs a GPU") >= 1 or "ProcessGroupNCCL" in out.stderr or "NCCL" in out.stderr   # ... its ranks ran bench.py (and found no GPU)
    assert out.returncode != 0                                                         # ... and its exit code came back
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_via_comm_refuses_a_launcher():
    """--via-comm is ONE process driving N GPUs through mvs_comm_*: under torch.distributed.run it must refuse before touching anything"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--via-comm"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "ONE process" in out.stderr
