"""Multi-GPU plumbing for the hot path.  Inference shards by independent stereo pairs -- one
process per GPU, no data-path collective; the only collective is the timing MAX."""
from __future__ import annotations

import torch


def shard_pairs(n_pairs: int, rank: int, world: int) -> list[int]:
    """Strided split of a list of pairs (what DistributedSampler(shuffle=False) does,
    stereo/datasets/__init__.py:64-65)."""
    return list(range(rank, n_pairs, world))


def reduce_step_time(local_seconds: float, device: torch.device) -> float:
    """Max over ranks of the timed region (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(local_seconds)
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(pairs_per_rank_per_step: int, steps: int, world: int, seconds: float) -> float:
    return world * pairs_per_rank_per_step * steps / seconds
