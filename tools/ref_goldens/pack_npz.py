"""Packs dump_goldens' .npy outputs (plus the inputs they were made from) into tests/golden/ref_*.npz:
    python3 pack_npz.py <io dir> <tests/golden>"""
import glob
import os
import sys

import numpy as np

io, out = sys.argv[1], sys.argv[2]
arrs = {os.path.basename(p)[:-4]: np.load(p) for p in glob.glob(os.path.join(io, "*.npy"))}
render = {k: v for k, v in arrs.items() if k.startswith("in_") or k in ("depth", "depth_side", "projected", "mixed", "depth_after_mix")}
flow = {k: v for k, v in arrs.items() if k in ("in_frame_a", "in_frame_b", "compare", "flow_remap", "flow_farneback", "flow_variational")}
for name, d in (("ref_render.npz", render), ("ref_flow.npz", flow)):
    missing = [k for k in d if d[k] is None]
    np.savez_compressed(os.path.join(out, name), **d)
    print("wrote", os.path.join(out, name), sorted(d), missing)
