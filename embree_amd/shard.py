"""Multi-GPU partitioning of a ray batch: rays are independent and the BVH is read-only,
so a batch is split into contiguous per-rank ranges and the BVH is replicated (every rank builds it from
the same inputs).  There is no data-path collective; ranks only meet at host-side barriers.
The reference has no multi-process code at all (single-process library) -- this is new design."""


def shard_range(total, rank, world):
    """Contiguous range [g*M/G, (g+1)*M/G) of rank g; ranges tile [0, total) exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return (total * rank) // world, (total * (rank + 1)) // world


def aggregate_throughput(units_per_rank, elapsed_max_s):
    """Whole-job rate = all units processed by all ranks / slowest rank's time (driver contract)."""
    return sum(units_per_rank) / elapsed_max_s


def max_over_ranks(value, dist=None):
    """MAX-reduce a host scalar over the process group (gloo); identity without a group."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_counts(count, dist=None):
    """All ranks' unit counts (host all_gather), used to compute the aggregate."""
    if dist is None or not dist.is_initialized():
        return [int(count)]
    import torch
    mine = torch.tensor([int(count)], dtype=torch.int64)
    out = [torch.zeros(1, dtype=torch.int64) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(o[0]) for o in out]
