"""Multi-GPU plumbing: instances shard by rank (no data-path collective); the only exchange is
one all-reduce of the KPI vector Main.m prints (ABO/Main.m:203-263).  Backend "nccl" is RCCL on
ROCm; the same code runs over gloo on CPU in the tests."""
from __future__ import annotations

import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(rank: int, world: int, per_rank: int):
    """Weak scaling: rank r owns instances [r*per_rank, (r+1)*per_rank)."""
    return rank * per_rank, (rank + 1) * per_rank


def reduce_kpis(kpi, world: int):
    """Sum-reduce a small KPI tensor over all ranks (in place); no-op for a single rank."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(kpi, op=dist.ReduceOp.SUM)
    return kpi


def max_over_ranks(value: float, world: int, device=None) -> float:
    if world <= 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
