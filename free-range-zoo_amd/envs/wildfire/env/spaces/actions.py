"""Action spaces of the wildfire environment (reference: envs/wildfire/env/spaces/actions.py:10-41).

Per env ``OneOf([Discrete(1, start=0)] * n + [Discrete(1, start=-1)])`` (fight task i / noop); the batch-level builder returns the
count-based ``BatchedOneOfSpace`` whose members are materialised only on inspection and which samples on the device."""
import functools

from free_range_zoo_amd.utils.spaces import BatchedOneOfSpace, Space


def build_action_space(environment_task_counts, sampler=None) -> BatchedOneOfSpace:
    return BatchedOneOfSpace(environment_task_counts, tail=[-1], sampler=sampler)


@functools.lru_cache(maxsize=100)
def build_single_action_space(num_tasks_in_environment: int):
    return Space.OneOf([Space.Discrete(1, start=0) for _ in range(num_tasks_in_environment)] + [Space.Discrete(1, start=-1)])
