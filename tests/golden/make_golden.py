"""Regenerates the small golden vectors from the CPU oracle (run in the dev container):
    python tests/golden/make_golden.py
The reference holds no golden outputs for this path, so these pin the oracle
against silent drift, not against the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import orc  # noqa: E402
from mvs_amd import synth  # noqa: E402


def main():
    o = orc.load()
    W, H, D, V = 48, 32, 12, 3
    main_cam, main_img, side_cams, sides, gt = synth.make_views(W, H, V, radius=0.3, freq_scale=0.3)
    depth, cost, idx, vol = o.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True)
    np.savez_compressed(os.path.join(HERE, "sweep_small.npz"), main_cam=main_cam, main_img=main_img,
                        side_cams=side_cams, side_imgs=np.stack(sides), D=D, depth=depth, cost=cost, idx=idx, vol=vol,
                        gt=gt)
    print("wrote sweep_small.npz", depth.shape, vol.shape)


if __name__ == "__main__":
    main()
