"""Observation spaces of the cybersecurity environment (reference: envs/cybersecurity/env/spaces/observations.py:11-165).

The space does not depend on the state: every env gets the same ``Dict{self, others, tasks}`` object, cached; `others` keeps the columns
``observe_other_power / presence (/ location)`` select, an empty selection being ``Discrete(0)`` as in the reference."""
import functools
from typing import List, Tuple

from free_range_zoo_amd.utils.spaces import Space


@functools.lru_cache(maxsize=100)
def build_observation_space(agent_type: str, num_nodes: int, parallel_envs: int, num_attackers: int, num_defenders: int, attacker_high: Tuple[int],
                            defender_high: Tuple[int], network_high: Tuple[int], include_power: bool, include_presence: bool,
                            include_location: bool) -> List:
    if agent_type == 'defender':
        space = build_single_defender_observation_space(defender_high=defender_high, network_high=network_high, num_tasks=num_nodes,
                                                        num_agents=num_defenders, include_power=include_power, include_presence=include_presence,
                                                        include_location=include_location)
    elif agent_type == 'attacker':
        space = build_single_attacker_observation_space(attacker_high=attacker_high, network_high=network_high, num_tasks=num_nodes,
                                                        num_agents=num_attackers, include_power=include_power, include_presence=include_presence)
    else:
        raise ValueError(f'Invalid agent type: {agent_type}')
    return [space] * parallel_envs


def _others(high: Tuple[int], keep: Tuple[bool, ...]) -> Tuple[int, ...]:
    return tuple(bound for bound, kept in zip(high, keep) if kept)


@functools.lru_cache(maxsize=100)
def build_single_attacker_observation_space(attacker_high: Tuple[int], network_high: Tuple[int], num_tasks: int, num_agents: int,
                                            include_power: bool = True, include_presence: bool = True):
    other_high = _others(attacker_high, (include_power, include_presence))
    return Space.Dict({
        'self': build_single_agent_observation_space(attacker_high),
        'others': Space.Tuple([build_single_agent_observation_space(other_high) for _ in range(num_agents - 1)]),
        'tasks': build_single_subnetwork_observation_space(network_high, num_tasks),
    })


@functools.lru_cache(maxsize=100)
def build_single_defender_observation_space(defender_high: Tuple[int], network_high: Tuple[int], num_tasks: int, num_agents: int,
                                            include_power: bool = True, include_presence: bool = True, include_location: bool = True):
    other_high = _others(defender_high, (include_power, include_presence, include_location))
    return Space.Dict({
        'self': build_single_agent_observation_space(defender_high),
        'others': Space.Tuple([build_single_agent_observation_space(other_high) for _ in range(num_agents - 1)]),
        'tasks': build_single_subnetwork_observation_space(network_high, num_tasks),
    })


@functools.lru_cache(maxsize=100)
def build_single_agent_observation_space(high: Tuple[int]):
    if len(high) == 0:
        return Space.Discrete(0, start=0)
    return Space.Box(low=[0] * len(high), high=[int(i) for i in high])


@functools.lru_cache(maxsize=None)
def build_single_subnetwork_observation_space(high: Tuple[int], num_tasks: int):
    return Space.Tuple([Space.Box(low=[0] * len(high), high=high) for _ in range(num_tasks)])
