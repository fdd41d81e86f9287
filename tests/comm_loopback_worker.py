"""Worker of tests/test_comm_loopback_gpu.py: one failure-injection case of mvs_sweep_sharded per process, so that a hang (the thing under
test) is a subprocess timeout, not a frozen suite.  Usage: comm_loopback_worker.py <n> <mode> <groups>; the environment carries
MVS_RCCL_LIBRARY / MVS_COMM_ALLOW_SAME_DEVICE and one of LOOPBACK_RCCL_FAIL / MVS_COMM_TEST_FAIL_RANK.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (first: libmvs_hip.so binds to the HIP runtime torch brings along)

import mvs_amd  # noqa: E402
from mvs_amd import synth  # noqa: E402


def main():
    n, mode, groups = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    W, H, D, V = 320, 200, 64, 8
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.3)
    out = {"n": n, "mode": mode}
    with mvs_amd.Comm([0] * n, W, H) as comm:
        comm.set_mode(mode, groups if mode == "views" else None)
        t0 = time.perf_counter()
        try:
            comm.sweep(main_cam, main_img, side_cams, sides, D)
            out["first"] = "ok"
        except mvs_amd.MvsError as e:
            out["first"] = str(e)
        out["first_s"] = time.perf_counter() - t0
        try:   # after a failed exchange the communicator refuses further work; after a failed local phase it is still usable
            d, _ = comm.sweep(main_cam, main_img, side_cams, sides, D)
            out["second"] = "ok"
            with mvs_amd.Context(W, H) as ctx:
                d1 = ctx.sweep(main_cam, main_img, side_cams, sides, D)
            out["second_equal"] = bool(np.array_equal(d, d1))
        except mvs_amd.MvsError as e:
            out["second"] = str(e)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
