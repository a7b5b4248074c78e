"""Env-batch sharding across the GPUs of one node (one process per GPU, ``torch.distributed``; backend ``nccl`` = RCCL).

Envs never interact, so rank r simply owns the global envs ``[r * B, (r + 1) * B)``: seeds are the global
env index, there is NO collective on the step path, and the only exchange is one small metrics reduction per episode.
The reference has no multi-device path; this is the MI355X-native addition the north star asks for.
"""
from typing import Optional, Tuple

import torch


def shard_range(rank: int, envs_per_rank: int) -> Tuple[int, int]:
    """Global env indices owned by ``rank``."""
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def shard_seeds(rank: int, envs_per_rank: int, base: int = 0, device=None) -> torch.Tensor:
    """Seeds of the rank's envs = ``base`` + global env index: a sharded run reproduces the unsharded one env for env."""
    start, stop = shard_range(rank, envs_per_rank)
    return torch.arange(start, stop, dtype=torch.int32, device=device) + base


def episode_metrics(cumulative_rewards: torch.Tensor, finished: torch.Tensor, env_steps: int) -> torch.Tensor:
    """Per-rank metrics vector: (sum of cumulative reward per agent ..., env-steps, finished envs), float64."""
    A = cumulative_rewards.shape[0]
    out = torch.zeros(A + 2, dtype=torch.float64, device=cumulative_rewards.device)
    out[:A] = cumulative_rewards.sum(dim=1, dtype=torch.float64)
    out[A] = float(env_steps)
    out[A + 1] = finished.sum()
    return out


def globalize_totals(totals: torch.Tensor, group=None) -> torch.Tensor:
    """Sum a shard's batch totals (int32 / int64 ``[A + 3]``: lit fires, fires each agent can attack, envs not terminated, envs not
    truncated) over the ranks, in place: the OPTIONAL per-step exchange of a sharded job that wants the reference's two batch-global step
    semantics — "every env is finished" (utils/env.py:211-213) and "agent a has no task in any env" (wildfire.py:434-435) — evaluated over
    the whole job instead of per shard.  With the ``nccl`` backend (RCCL) the tensor stays on the device; ``gloo`` needs it
    on the host (tests).  A no-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(totals, op=dist.ReduceOp.SUM, group=group)
    return totals


def reduce_metrics(metrics: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the metrics vector over all ranks (the job's only collective; a no-op without an initialised process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(metrics, group=group)
    return metrics
