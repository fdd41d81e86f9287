"""Configuration's clip input (configuration.cpp:169-245) from a YUV4MPEG2 stream: frame fi * skip of the stream becomes tracked frame
fi, a frame of another size is resized (cv::resize INTER_LINEAR: orc_resize_linear_u8's arithmetic; on the GPU), colour becomes grey by
BGR2GRAY.  Checked against the stated formulas in numpy (BT.601 limited range, 16.16; cvtColor's 14-bit weights)."""
import os
import re
import subprocess

import numpy as np

import tracks_yaml as tracks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "mesh-reconstruction_amd", os.environ.get("MVS_BUILD_VARIANT", ""), "bin", "host_selftest")
TRACKS = os.path.join(ROOT, "tests", "data", "tracks")


def run(tmp_path, oracle, scale):
    text = open(os.path.join(TRACKS, "zatisi.yaml")).read()
    yaml_path = tmp_path / "zatisi.yaml"
    yaml_path.write_text(text)
    t = tracks.load("zatisi.yaml")
    W, H, n = t["width"], t["height"], len(t["cameras"])
    clip = re.search(r"path:\s*\"?([^\s\"]+)", text).group(1)
    skip = 3
    nf = 3 * 7 + 1                                             # enough stream frames for 8 tracked frames at skip 3
    rng = np.random.default_rng(3)
    w, h = W // scale, H // scale
    stream = []
    with open(tmp_path / (clip + ".y4m"), "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C420jpeg\n" % (w, h))
        for i in range(nf):
            yy, xx = np.mgrid[0:h, 0:w]
            Y = (16 + 219 * (0.5 + 0.5 * np.sin((xx + 5 * i) / 19.0) * np.cos(yy / 23.0))).astype(np.uint8)
            U = rng.integers(90, 170, (h // 2, w // 2), dtype=np.uint8)
            V = rng.integers(90, 170, (h // 2, w // 2), dtype=np.uint8)
            f.write(b"FRAME\n" + Y.tobytes() + U.tobytes() + V.tobytes())
            stream.append((Y, U, V))
    r = subprocess.run([SELFTEST, "frames", str(yaml_path), str(tmp_path), str(skip)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    tracked = (n + skip - 1) // skip
    have = min(tracked, (nf + skip - 1) // skip)
    assert "frames selftest: %d of %d frames, %d x %d" % (have, tracked, W, H) in r.stdout

    def expect(Y, U, V):
        c = 298 * (Y.astype(np.int64) - 16)
        d = np.repeat(np.repeat(U.astype(np.int64) - 128, 2, 0), 2, 1)
        e = np.repeat(np.repeat(V.astype(np.int64) - 128, 2, 0), 2, 1)
        R = np.clip((c + 409 * e + 128) >> 8, 0, 255)
        G = np.clip((c - 100 * d - 208 * e + 128) >> 8, 0, 255)
        B = np.clip((c + 516 * d + 128) >> 8, 0, 255)
        bgr = np.stack([B, G, R], -1).astype(np.uint8)
        if scale != 1:
            bgr = np.stack([oracle.resize_linear(np.ascontiguousarray(bgr[:, :, k]), W, H) for k in range(3)], -1)
        big = bgr.astype(np.int64)
        return ((big[:, :, 0] * 1868 + big[:, :, 1] * 9617 + big[:, :, 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)

    for fi in (0, 1, 5, 7):
        got = np.fromfile(tmp_path / ("gray_%03d.u8" % fi), np.uint8).reshape(H, W)
        np.testing.assert_array_equal(got, expect(*stream[fi * skip]))
    assert not os.path.exists(tmp_path / ("gray_%03d.u8" % have))   # the stream ended: later tracked frames were not supplied
