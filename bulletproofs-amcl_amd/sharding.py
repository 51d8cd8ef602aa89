"""Index-range sharding of an MSM over the ranks of one node (one process per GPU) and the single exchange step.

MSM is a commutative-monoid sum, so rank g owns the contiguous range [g*n/N, (g+1)*n/N) of points and scalars, runs
the whole bucket pipeline on it, and contributes W window records; one all_gather (RCCL on GPUs, gloo in the CPU
tests) replaces the "reduce" -- point addition is not an RCCL reduction op.  No arithmetic here.
"""


def shard_range(n_total, world, rank):
    """Contiguous [lo, hi) of rank `rank`; sizes differ by at most one; the ranges tile [0, n_total)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_records(mine, world):
    """Gather every rank's record tensor (uint8, same length on all ranks) into one [world * len] tensor, in rank
    order.  Uses all_gather_into_tensor where the backend has it (nccl = RCCL), all_gather otherwise (gloo)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return mine
    if dist.get_backend() == "nccl":
        out = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine)
    else:                                     # gloo: CPU tests and one-device rehearsals; records travel through host memory
        host = mine.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        out = torch.cat(parts).to(mine.device)
    return out


def largest_shard(n_total, world):
    return -(-n_total // world)


def common_window_bits(bp, curve, n_total, world):
    """The window width every rank must fix (Context.set_window_bits) before Context-level two-stage calls when shard sizes may
    differ: the width the library picks for the LARGEST shard.  Record blocks carry their geometry and bp_msm_g1_finish
    refuses sets that disagree, so forgetting this fails loudly instead of mis-weighting windows."""
    return bp.msm_geometry(curve, largest_shard(n_total, world))[0]


# ---- 2-D sharding: index range x window group (round 4) ------------------------------------------------------------------------
# The index-range split leaves every rank with ALL windows of its slice: bucket reduce, sort and host tail do not shrink with the
# slice (0.84 of 1.94 ms at 2^19 points), which caps the strong scaling of BASELINE config 4 at ~5.4x on 8 GPUs.  Splitting the W
# windows over `window_groups` ranks that share a `window_groups` times longer slice keeps the accumulate work per rank and divides
# the rest.  Rank r: index group r // window_groups, window group r % window_groups.


def plan_2d(world, windows, window_groups=None):
    """(index_groups, window_groups) with index_groups * window_groups == world and window_groups | windows.  Default: as many window
    groups as divide both (8 ranks, 16 windows -> 1 x 8: every rank all points, 2 windows)."""
    if window_groups is None:
        window_groups = max(g for g in range(1, world + 1) if world % g == 0 and windows % g == 0)
    if world % window_groups or windows % window_groups:
        raise ValueError("window_groups must divide both the world size and the window count")
    return world // window_groups, window_groups


def shard_2d(n_total, world, rank, windows, window_groups=None):
    """-> (lo, hi, w_first, w_count) of rank `rank`."""
    ig, wg = plan_2d(world, windows, window_groups)
    lo, hi = shard_range(n_total, ig, rank // wg)
    per = windows // wg
    return lo, hi, (rank % wg) * per, per
