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


# ---- sparse row-sharded exchange (include/charon_hip.h: chn_shardx_*) ---------------------------------------------------------
def row_splits(n_rows, world):
    """[splits[r], splits[r+1]) = rows of rank r: the same contiguous, covering ranges as shard_range"""
    return [shard_range(n_rows, r, world)[0] for r in range(world)] + [int(n_rows)]


def all_to_all_v(dist, send, send_counts, width=1):
    """Variable-size all-to-all of a 1-D tensor made of `world` consecutive groups (group r goes to rank r; send_counts[r] items of
    `width` elements each).  Returns (recv, recv_counts): the groups received from rank 0, 1, ... in rank order.
    RCCL (backend nccl): one all_to_all_single with split sizes.  gloo has no all-to-all: emulated with all_gather (CPU tests)."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    counts = torch.tensor([int(c) for c in send_counts], dtype=torch.int64, device=send.device)
    if dist.get_backend() == "nccl":
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts)
        rc = [int(x) for x in recv_counts.tolist()]
        recv = torch.empty(sum(rc) * width, dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, output_split_sizes=[c * width for c in rc], input_split_sizes=[int(c) * width for c in send_counts])
        return recv, rc
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    longest = max(int(c.sum()) for c in all_counts) * width
    padded = torch.zeros(max(longest, 1), dtype=send.dtype, device=send.device)
    padded[:send.numel()] = send
    everyone = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(everyone, padded)
    parts, rc = [], []
    for src in range(world):
        c = [int(x) for x in all_counts[src].tolist()]
        lo = sum(c[:rank]) * width
        parts.append(everyone[src][lo:lo + c[rank] * width])
        rc.append(c[rank])
    return (torch.cat(parts) if parts else send[:0]), rc
