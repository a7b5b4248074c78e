"""Action spaces of the cybersecurity environment (reference: envs/cybersecurity/env/spaces/actions.py:11-99).

Attackers: ``OneOf([attack node] * n + [noop -1])``; defenders: ``OneOf([move to node] * n + [noop -1, patch -2, monitor -3])`` with patch
left out at the home node unless bad actions are shown, and only the noop for an agent that is absent (n = 0)."""
import functools

import torch

from free_range_zoo_amd.utils.spaces import BatchedOneOfSpace, Space


def build_action_space(agent_type: str, show_bad_actions: bool, environment_task_counts: torch.Tensor, current_location: torch.Tensor = None,
                       sampler=None, epoch=None) -> BatchedOneOfSpace:
    if agent_type == 'attacker':
        return BatchedOneOfSpace(environment_task_counts, tail=[-1], sampler=sampler, epoch=epoch)
    if agent_type != 'defender':
        raise ValueError(f'Invalid agent type: {agent_type}')
    counts, location = environment_task_counts, current_location

    def tail_mask() -> torch.Tensor:  # which of (noop, patch, monitor) exist per env; resolved by code that inspects the members
        has_tasks = counts > 0
        can_patch = has_tasks & (torch.full_like(has_tasks, bool(show_bad_actions)) | (location != -1))
        return torch.stack([torch.ones_like(has_tasks), can_patch, has_tasks], dim=1)

    return BatchedOneOfSpace(counts, tail=[-1, -2, -3], tail_mask=tail_mask, sampler=sampler, epoch=epoch)


@functools.lru_cache(maxsize=100)
def build_single_defender_action_space(num_tasks_in_environment: int, current_location: int, show_bad_actions: bool):
    if num_tasks_in_environment == 0:
        return Space.OneOf([Space.Discrete(1, start=-1)])
    tail = [-1, -3] if (not show_bad_actions and current_location == -1) else [-1, -2, -3]
    return Space.OneOf([Space.Discrete(1, start=0) for _ in range(num_tasks_in_environment)] + [Space.Discrete(1, start=v) for v in tail])


@functools.lru_cache(maxsize=100)
def build_single_attacker_action_space(num_tasks_in_environment: int):
    return Space.OneOf([Space.Discrete(1, start=0) for _ in range(num_tasks_in_environment)] + [Space.Discrete(1, start=-1)])
