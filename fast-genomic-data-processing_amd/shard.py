"""Multi-GPU sharding of the two hot paths: one process per GPU, NO collective on the data path.

PairHMM test cases are independent, so rank r owns ``shard_bounds(total, r, world)`` of ONE stream (its resident
batch and the range it gives its host work queue, ``PairHMMQueue.run(lo, hi)``); sort / mark-duplicate shards are
coordinate ranges cut by the host router (``sortdedup.Routed``, mgx_sortdedup_route) with ``only_shard = rank``
(SURVEY.md section 8e).
``torch.distributed`` is used for exactly two things: the barrier around the timed region and the
max-over-ranks of its duration (the bench contract)."""
import os



def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items work units for this rank."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
