"""The C++ host mirror of recon.hpp (Configuration, Heuristic, util) -- CPU part, via host_selftest."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "mesh-reconstruction_amd", "bin", "host_selftest")
TRACKS = os.path.join(ROOT, "tests", "data", "tracks")


def test_host_selftest_cpu():
    """cv::theRNG known answers, the four bundled YAML-tracks files, camera centres, skipFrames, getopt, filterPoints"""
    assert os.path.exists(SELFTEST), "run __graft_entry__.build() first"
    r = subprocess.run([SELFTEST, "cpu", TRACKS], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu selftest: 0 failures" in r.stdout


def test_host_library_links_only_the_c_abi():
    lib = os.path.join(ROOT, "mesh-reconstruction_amd", "lib", "libmvs_host.so")
    out = subprocess.check_output(["readelf", "-d", lib]).decode()
    assert "libmvs_hip.so" in out and "oracle" not in out
    syms = subprocess.check_output(["nm", "-DC", lib]).decode()
    for name in ["spawnRender(Heuristic)", "calculateFlow(", "mixBackground(", "Heuristic::chooseCameras", "Configuration::Configuration"]:
        assert name in syms, name
