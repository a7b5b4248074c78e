"""Observation spaces of the rideshare environment (reference: envs/rideshare/env/spaces/observations.py:9-88).

Per env ``Dict{self: Box(y, x, #accepted, #riding), others: Tuple[Box] * (agents - 1), tasks: Tuple[Box] * n}`` with one
(y, x, y_dest, x_dest, accepted_by, riding_by, fare, entered_step) box per task the agent sees; count-based at the batch level."""
import functools
from typing import Tuple

from free_range_zoo_amd.utils.spaces import BatchedSpace, Space


def build_observation_space(environment_task_counts, num_agents: int, agent_high: Tuple[int], passenger_high: Tuple[int]) -> BatchedSpace:
    agent_high, passenger_high = tuple(agent_high), tuple(passenger_high)
    return BatchedSpace(environment_task_counts, lambda n: build_single_observation_space(agent_high, passenger_high, n, num_agents))


@functools.lru_cache(maxsize=100)
def build_single_observation_space(agent_high: Tuple[int], passenger_high: Tuple[int], num_tasks: int, num_agents: int):
    return Space.Dict({
        'self': build_single_agent_observation_space(agent_high),
        'others': Space.Tuple([build_single_agent_observation_space(agent_high) for _ in range(num_agents - 1)]),
        'tasks': build_single_passenger_observation_space(passenger_high, num_tasks),
    })


@functools.lru_cache(maxsize=100)
def build_single_agent_observation_space(high: Tuple[int]):
    return Space.Box(low=[0] * len(high), high=high)


@functools.lru_cache(maxsize=100)
def build_single_passenger_observation_space(high: Tuple[int], num_tasks: int):
    return Space.Tuple([Space.Box([0] * len(high), high=high) for _ in range(num_tasks)])
