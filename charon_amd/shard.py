"""Read sharding for the multi-GPU mode.

Reads are independent units: every rank holds a full index replica and classifies a contiguous
slice of the read stream; nothing is exchanged on the data path.  The only collective is the sum of the
per-category summary counters (ResultSummary, include/result.hpp:18-25 of the reference), done once at the end.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """contiguous, disjoint, covering: rank r gets [lo, hi); sizes differ by at most one read"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def summary_counts(call, num_categories):
    """[classified per category..., unclassified] from a uint8 call vector (255 = no call)"""
    call = np.asarray(call, dtype=np.uint8)
    out = np.zeros(num_categories + 1, dtype=np.int64)
    for c in range(num_categories):
        out[c] = int((call == c).sum())
    out[num_categories] = int((call == 255).sum())
    return out


def merge_summary(local, dist, device=None):
    """sum-all-reduce of the summary counters (RCCL on GPUs, gloo in the CPU tests)"""
    import torch
    t = torch.as_tensor(np.asarray(local, dtype=np.int64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
