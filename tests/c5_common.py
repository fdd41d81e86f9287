"""BASELINE config 5 as far as it is in scope (tracks/zatisi.yaml: 120 calibrated 640x480 frames, 128 planes, every main frame
against its 4 neighbours at -10, -5, +5, +10 frames; the reference's main-camera loop recon.cpp:65-117 shards by main frame):
inputs shared by tests/test_c5_gpu.py and its two-rank worker.  The clip is missing from the reference checkout
(.MISSING_LARGE_BLOBS), so the frames are synthetic and deterministic."""
import zlib

import numpy as np

import scenes
import tracks_yaml

PLANES = 128
OFFSETS = (-10, -5, 5, 10)
STRIDE = 4          # every 4th main frame: 30 of the 120


class Sequence:
    def __init__(self):
        t = tracks_yaml.load("zatisi.yaml")
        self.W, self.H, self.cams = t["width"], t["height"], t["cameras"]
        self.n = len(self.cams)
        self.bundles = t["bundles"]
        self.mains = list(range(0, self.n, STRIDE))
        rng = np.random.default_rng(5)
        base = rng.integers(0, 256, (self.H // 4 + 2, self.W // 4 + 70), dtype=np.uint8).astype(np.float32)
        self._big = np.kron(base, np.ones((4, 4), np.float32))
        # one proxy mesh for the whole sequence, as the reference renders one mesh per outer iteration (recon.cpp:42-64)
        self.verts, self.faces = scenes.proxy_plane(self.bundles, self.cams[self.n // 2], scale=0.6)

    def frame(self, f):
        """a textured strip scrolling 2 px per frame plus a frame-dependent ripple: no two frames are equal"""
        yy, xx = np.mgrid[0:self.H, 0:self.W]
        img = self._big[:self.H, 2 * (f % 100):2 * (f % 100) + self.W] * 0.8 + 25.0 * np.sin((xx + 3 * f) / 17.0) * np.cos((yy - 2 * f) / 21.0) + 25.0
        return img.clip(0, 255).astype(np.uint8)

    def sides(self, f):
        return [min(self.n - 1, max(0, f + o)) for o in OFFSETS]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def process_main_frame(ctx, seq, f):
    """what one rank does for main frame f: the D-plane sweep through the one-call entry and the reference's own per-frame
    stage; returns (depth map, best cost, point block)"""
    ids = seq.sides(f)
    cams = np.stack([seq.cams[j] for j in ids])
    frames = [seq.frame(j) for j in ids]
    depth, cost = ctx.sweep(seq.cams[f], seq.frame(f), cams, frames, PLANES, want_cost=True)
    pts = ctx.process_frame(seq.cams[f], seq.frame(f), cams, frames, False)
    return depth, cost, pts
