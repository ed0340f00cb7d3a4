"""Multi-GPU plumbing: one process per GPU, clips sharded by batch, NO data-path collective (clips never interact in
the forward: SURVEY section 8e) and exactly one collective per run -- an ``all_gather`` of per-rank metric scalars
(RCCL over xGMI with backend "nccl"; "gloo" on CPU in the tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of ``n_items`` clips for ``rank`` (first ``n_items % world`` ranks get one more)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_metrics(values: Sequence[float], device=None) -> torch.Tensor:
    """all_gather a short vector of float64 metrics; returns [world, len(values)] on the CPU (on every rank)."""
    mine = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():        # also at world size 1: the collective then runs over the real backend
        out: List[torch.Tensor] = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(out, mine)
        return torch.stack(out).cpu()
    return mine.cpu().unsqueeze(0)


def whole_job_rate(units_per_rank: Sequence[int], seconds_per_rank: Sequence[float]) -> float:
    """Aggregate throughput: all units processed divided by the slowest rank's time."""
    return float(sum(units_per_rank)) / max(seconds_per_rank)
