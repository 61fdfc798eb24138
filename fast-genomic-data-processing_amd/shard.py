"""Multi-GPU sharding of the two hot paths: one process per GPU, NO collective on the data path.

PairHMM test cases are independent and sort/mark-duplicate shards are coordinate ranges routed by
the host, so a rank only ever needs to know which slice is its own.
``torch.distributed`` is used for exactly two things: the barrier around the timed region and the
max-over-ranks of its duration (the bench contract)."""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items work units for this rank."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_pairs(d, rank, world):
    """Slice of a packed PairHMM batch: the rank's share of the test-case list (reads and
    haplotypes stay shared and are only referenced by index)."""
    lo, hi = shard_bounds(len(d["pair_read"]), rank, world)
    out = dict(d)
    out["pair_read"] = np.ascontiguousarray(d["pair_read"][lo:hi])
    out["pair_hap"] = np.ascontiguousarray(d["pair_hap"][lo:hi])
    return out, (lo, hi)


def coordinate_shards(coord, L, world):
    """Host routing step of the sort/mark-duplicate path: rank k owns unified coordinates
    [k*ceil(L/world), (k+1)*ceil(L/world)) -- what the reference's 100 range partitions do
    (sortmardup/tbb/range_partitioner.h:98-100).  Returns the owning rank of every key."""
    width = (L + world) // world
    return np.minimum(coord // np.uint64(width), np.uint64(world - 1)).astype(np.int64)


def max_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
