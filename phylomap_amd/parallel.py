"""Multi-GPU plumbing: one process per GPU, replicas (independent chains / sites) sharded across ranks.

The sampling itself needs no exchange: a replica's Philox streams are keyed by its GLOBAL
replica id, so a rank's results do not depend on how many ranks there are.  The only collective is the
sum-reduction of the per-iteration sufficient statistics (RCCL over xGMI: backend "nccl"; "gloo" in CPU tests).
"""
from __future__ import annotations

import os


def env_rank():
    """(rank, world_size, local_rank) from the torch.distributed.run environment (single process: 0, 1, 0)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def weak_shard(per_rank: int, rank: int):
    """Weak scaling: every rank runs `per_rank` replicas; returns (global offset, count)."""
    return rank * per_rank, per_rank


def split_replicas(total: int, world: int, rank: int):
    """Strong scaling: split `total` replicas as evenly as possible; returns (global offset, count)."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def init_process_group(backend: str | None = None, device_index: int | None = None):
    """Initialise torch.distributed from the environment (MASTER_ADDR must be 127.0.0.1 on one node)."""
    import torch
    import torch.distributed as dist
    rank, world, local_rank = env_rank()
    if world == 1:
        return rank, world, local_rank
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
    dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def allreduce_stats(stats):
    """In-place sum over ranks of an (iterations x columns) float64 tensor of sufficient statistics."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats
