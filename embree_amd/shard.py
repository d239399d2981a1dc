"""Multi-GPU partitioning of a ray batch: rays are independent and the BVH is read-only,
so a batch is split into contiguous per-rank ranges and the BVH is replicated (every rank builds it from
the same inputs: the build is deterministic, the replicas are bit-identical, nothing is broadcast).
Tracing needs no exchange between the ranks.  What the consumer of the results needs afterwards is the one
real exchange step of the path: the fields a query WRITES are packed on the GPU (mi355_pack_hits /
mi355_pack_occluded, embree_amd/csrc/shard.hip) and gathered over RCCL / xGMI -- `Communicator` below:
ncclGather to rank 0 for closest hits, ncclAllGather for occlusion results (north star of BASELINE.json).
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


# ---- results gathered on the GPU over RCCL / xGMI (embree_amd/csrc/shard.hip; SURVEY.md 8(e), BASELINE.json north_star) ----------------------------
class Communicator:
    """One RCCL communicator per process (= per GPU).  Rank 0 makes the 128-byte id (ncclGetUniqueId); it reaches the other ranks through the
    torch.distributed group the launcher already formed (gloo broadcast of 128 bytes) -- torch is the rendezvous, RCCL moves the data.
    world == 1 works too (RCCL runs a one-rank communicator), which is how the GPU test exercises this path on a 1-GPU box."""

    def __init__(self, gpu, rank, world, dist=None):
        import ctypes as C
        from . import api
        self.L, self.rank, self.world, self.gpu = api.load(), rank, world, gpu
        ident = (C.c_char * 128)()
        if rank == 0:
            if self.L.mi355_comm_unique_id(ident) != 0:
                raise RuntimeError("mi355_comm_unique_id: " + self.L.mi355_last_error().decode())
        if world > 1:
            if dist is None or not dist.is_initialized():
                raise RuntimeError("a torch.distributed group is needed to pass the RCCL id to the other ranks")
            import torch
            t = torch.tensor(list(ident.raw), dtype=torch.uint8)
            dist.broadcast(t, src=0)
            ident = (C.c_char * 128)(*bytes(t.tolist()))
        h = C.c_void_p()
        if self.L.mi355_comm_init(gpu, ident, world, rank, C.byref(h)) != 0:
            raise RuntimeError("mi355_comm_init: " + self.L.mi355_last_error().decode())
        self.h = h

    def allgather(self, d_send, d_recv, bytes_per_rank, stream=None):
        if self.L.mi355_comm_allgather(self.h, d_send, d_recv, bytes_per_rank, stream) != 0:
            raise RuntimeError("mi355_comm_allgather: " + self.L.mi355_last_error().decode())

    def gather(self, d_send, d_recv, bytes_per_rank, root=0, stream=None):
        if self.L.mi355_comm_gather(self.h, d_send, d_recv, bytes_per_rank, root, stream) != 0:
            raise RuntimeError("mi355_comm_gather: " + self.L.mi355_last_error().decode())

    def close(self):
        if self.h:
            self.L.mi355_comm_destroy(self.h)
            self.h = None


def gather_host(array, dist=None):
    """Host-side stand-in for the device gather (tests on CPU, 2 ranks on one GPU where RCCL refuses two ranks per device):
    every rank contributes a 1-D numpy array of the SAME length, every rank gets the concatenation in rank order."""
    import numpy as np
    if dist is None or not dist.is_initialized():
        return np.array(array, copy=True)
    import torch
    mine = torch.from_numpy(np.ascontiguousarray(array).view(np.uint8).copy())
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return np.concatenate([o.numpy() for o in out]).view(array.dtype)
