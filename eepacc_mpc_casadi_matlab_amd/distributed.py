"""Multi-GPU plumbing: instances shard by rank (no data-path collective); the only exchange is the reduction of
the key figures Main.m prints (ABO/Main.m:203-263): sums (bad exits, distance, energy, travel time at the cut-off
distance, sum a^2, sum j^2, samples) with one all-reduce(SUM), extremes of a and j with one all-reduce(MIN) and one
all-reduce(MAX).  Backend "nccl" is RCCL on ROCm; the same code runs over gloo on CPU in the tests."""
from __future__ import annotations

import os

SUM_FIELDS = ("bad_exits", "distance_m", "energy_J", "travel_time_at_cutoff_s", "reached_cutoff", "sum_a2", "sum_j2",
              "samples", "instances")
EXT_FIELDS = ("a", "j")


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(rank: int, world: int, per_rank: int):
    """Weak scaling: rank r owns instances [r*per_rank, (r+1)*per_rank)."""
    return rank * per_rank, (rank + 1) * per_rank


def local_kpis(traj, status, E, Ts: float, cutoff_dist: float, OUT):
    """Key figures of this rank's instances from a closed-loop window.  traj [n][OUT_N][B], status [n][B],
    E [n][B] (cumulative energy of the window, eepacc_postprocess) -- torch tensors on any device.
    Returns (sums [len(SUM_FIELDS)], mins [2], maxs [2]) as float64 tensors on that device."""
    import torch
    n, _, B = traj.shape
    s = traj[:, OUT["s"]]; a = traj[:, OUT["a"]]
    j = (a[1:] - a[:-1]) / Ts if n > 1 else torch.zeros((1, B), dtype=traj.dtype, device=traj.device)   # j_opt = diff(a_opt)/Ts
    # travel time at the cut-off distance (Main.m:150-161, 236-238): first sample at or beyond it, if reached in the window
    beyond = s >= cutoff_dist
    reached = beyond.any(dim=0)
    first = torch.argmax(beyond.to(torch.int8), dim=0).to(torch.float64)
    t_cut = torch.where(reached, first * Ts, torch.zeros_like(first))
    sums = torch.stack([
        (status != 0).sum().to(torch.float64),
        (s[-1] - s[0]).sum(),
        E[-1].sum(),
        t_cut.sum(),
        reached.sum().to(torch.float64),
        (a * a).sum(),
        (j * j).sum(),
        torch.tensor(float(n * B), dtype=torch.float64, device=traj.device),
        torch.tensor(float(B), dtype=torch.float64, device=traj.device),
    ]).to(torch.float64)
    mins = torch.stack([a.min(), j.min()]).to(torch.float64)
    maxs = torch.stack([a.max(), j.max()]).to(torch.float64)
    return sums, mins, maxs


def reduce_kpis(sums, mins, maxs, world: int):
    """All-reduce the three small KPI tensors over all ranks (in place); no-op for a single rank."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(mins, op=dist.ReduceOp.MIN)
        dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
    return sums, mins, maxs


def kpi_dict(sums, mins, maxs):
    """The reduced key figures as the numbers Main.m:203-263 prints (whole job)."""
    import math
    v = dict(zip(SUM_FIELDS, [float(x) for x in sums.tolist()]))
    out = {
        "bad_exits": int(round(v["bad_exits"])),
        "distance_km": v["distance_m"] / 1e3,
        "energy_kWh": v["energy_J"] / 3.6e6,
        "instances_reaching_cutoff": int(round(v["reached_cutoff"])),
        "mean_travel_time_at_cutoff_s": (v["travel_time_at_cutoff_s"] / v["reached_cutoff"]) if v["reached_cutoff"] > 0 else None,
        "a_rms": math.sqrt(v["sum_a2"] / max(v["samples"], 1.0)),
        "j_rms": math.sqrt(v["sum_j2"] / max(v["samples"] - v["instances"], 1.0)),
        "a_min": float(mins[0]), "a_max": float(maxs[0]), "j_min": float(mins[1]), "j_max": float(maxs[1]),
        "samples": int(round(v["samples"])), "instances": int(round(v["instances"])),
    }
    return out


def max_over_ranks(value: float, world: int, device=None) -> float:
    if world <= 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
