"""Deterministic inputs for dump_goldens (numpy only): python3 make_inputs.py <dir>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests"))
from data import glx_scene  # noqa: E402  the reference's own self-test inputs (render_glx.cpp:407-410), held as data


def frames(W, H):
    """two related frames: a smooth pattern and the same pattern moved by a few pixels with a brightness ripple"""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    a = 127.5 + 60 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 40 * np.sin((xx + 2 * yy) / 41.0)
    b = 127.5 + 60 * np.sin((xx - 3.0) / 23.0) * np.cos((yy + 2.0) / 17.0) + 40 * np.sin((xx - 3.0 + 2 * (yy + 2.0)) / 41.0)
    rng = np.random.Generator(np.random.PCG64(0x5EED0008))
    a += rng.normal(0, 2.0, a.shape)
    b += rng.normal(0, 2.0, b.shape)
    return a.clip(0, 255).astype(np.uint8), b.clip(0, 255).astype(np.uint8)


def main(out):
    os.makedirs(out, exist_ok=True)
    W, H = glx_scene.W, glx_scene.H
    fa, fb = frames(W, H)
    for name, arr in (("points", np.asarray(glx_scene.POINTS, np.float32)), ("faces", np.asarray(glx_scene.FACES, np.int32)),
                      ("mvp", np.asarray(glx_scene.MVP, np.float32)), ("side_mvp", np.asarray(glx_scene.SIDE_MVP, np.float32)),
                      ("frame_a", fa), ("frame_b", fb)):
        np.save(os.path.join(out, "in_%s.npy" % name), np.ascontiguousarray(arr))
    print("wrote inputs to", out, "(%d x %d)" % (W, H))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "io"))
