"""Two-rank worker of tests/test_c5_gpu.py (launched with torch.distributed.run): main frames are dealt round robin to the ranks
(the reference's independent `fa` loop, recon.cpp:65), every rank runs them through the C ABI on the GPU, the per-frame
results are gathered on the host in frame order (recon.cpp:115-116 appends the point blocks) and rank 0 prints their checksums.
Both ranks share GPU 0 and talk over gloo: a test hook for one-GPU boxes (RCCL refuses two ranks per device)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

import c5_common  # noqa: E402
import mvs_amd  # noqa: E402
from mvs_amd import dist as mdist  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    seq = c5_common.Sequence()
    mine = mdist.frame_shard(len(seq.mains), rank, world)
    local = []
    with mvs_amd.Context(seq.W, seq.H, 0) as ctx:
        ctx.load_mesh(seq.verts, seq.faces)
        for k in mine:
            depth, cost, pts = c5_common.process_main_frame(ctx, seq, seq.mains[k])
            local.append((depth, pts))
    gathered = mdist.gather_frames(dist, local, len(seq.mains), rank, world)   # frame order, like the reference's append
    if rank == 0:
        cloud = np.concatenate([pts for _, pts in gathered])                  # recon.cpp:115-116
        print(json.dumps({"frames": [{"main": seq.mains[k], "depth_crc": c5_common.crc(d), "points": int(p.shape[0]), "points_crc": c5_common.crc(p[:, :4])}
                                     for k, (d, p) in enumerate(gathered)], "cloud_points": int(cloud.shape[0]), "cloud_crc": c5_common.crc(cloud[:, :4])}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
