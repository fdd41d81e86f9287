"""Multi-GPU sharding of the sweep (one process per GPU, torch.distributed; backend nccl = RCCL on ROCm).

Three shardings:
  frames: rank r owns main frames r, r+world, ... -- the reference's independent `fa` loop (recon.cpp:65); no collective
  views : the side views of ONE main frame are split across ranks; the packed u32 volume (count<<16 | sum) is summed
          with all_reduce -- exact, because cells are integers -- and every rank selects depth on the full volume.
  rows  : the pixel rows of ONE main frame are split into bands, one per rank (rows are independent); every
          rank holds all side views, sweeps its band with fused depth selection, and the bands of the depth map are
          all-gathered (4 bytes per pixel in total) -- strong scaling of a single main view with no volume exchange.
The functions here are backend-agnostic so the CPU tests run them over gloo with the oracle standing in for the GPU.
"""


def view_shard(V, rank, world):
    """contiguous view range [first, first+count) of `rank`; ranks beyond V get an empty range"""
    per = (V + world - 1) // world
    first = min(rank * per, V)
    return first, max(0, min(per, V - first))


def frame_shard(num_frames, rank, world):
    """main-frame indices processed by `rank` (round robin, like `fa % world == rank`)"""
    return list(range(rank, num_frames, world))


def allreduce_volume(dist, volume_tensor):
    """in-place SUM all-reduce of the packed volume (an int32/int64 torch tensor aliasing the u32 cells)"""
    dist.all_reduce(volume_tensor, op=dist.ReduceOp.SUM)
    return volume_tensor


def gather_frames(dist, local_results, num_frames, rank, world):
    """collect per-frame results (picklable) of the frame sharding on every rank, ordered by frame index"""
    gathered = [None] * world
    dist.all_gather_object(gathered, local_results)
    out = [None] * num_frames
    for r, items in enumerate(gathered):
        for idx, value in zip(frame_shard(num_frames, r, world), items):
            out[idx] = value
    return out


def plane_groups(D, groups, granularity):
    """split planes 0..D into `groups` contiguous ranges whose boundaries are multiples of `granularity`"""
    units = (D + granularity - 1) // granularity
    groups = max(1, min(groups, units))
    out, start = [], 0
    for g in range(groups):
        n = units // groups + (1 if g < units % groups else 0)
        first = start * granularity
        last = min(D, (start + n) * granularity)
        out.append((first, last - first))
        start += n
    return out


def row_bands(H, world, granularity):
    """split rows 0..H into `world` contiguous bands [first, first+count) whose boundaries are multiples of `granularity`
    (the last band ends at H); ranks beyond the number of units get empty bands.  Bands differ by at most one unit."""
    units = (H + granularity - 1) // granularity
    out, start = [], 0
    for r in range(world):
        n = units // world + (1 if r < units % world else 0)
        first = min(H, start * granularity)
        last = min(H, (start + n) * granularity)
        out.append((first, last - first))
        start += n
    return out


def equal_row_bands(H, world, granularity):
    """as row_bands, but every band has the same height `tallest` (a multiple of `granularity`) except the last non-empty one, which
    takes the remainder: band r starts at row r * tallest, so bands gathered at a fixed stride are contiguous rows of the image.
    The tallest band is as tall as row_bands' tallest (the step time is the same), the last one is shorter."""
    units = (H + granularity - 1) // granularity
    tallest = ((units + world - 1) // world) * granularity
    out = []
    for r in range(world):
        first = min(H, r * tallest)
        out.append((first, min(H, first + tallest) - first))
    return out


def gather_rows(dist, torch, local_rows, bands, W):
    """all-gather row bands of unequal height: every rank contributes its [count, W] band (padded to the tallest band),
    returns the assembled [H, W] map on every rank"""
    tallest = max(c for _, c in bands)
    pad = torch.zeros((tallest, W), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    parts = [torch.empty_like(pad) for _ in bands]
    dist.all_gather(parts, pad)
    return torch.cat([part[:c] for part, (_, c) in zip(parts, bands)], dim=0)
