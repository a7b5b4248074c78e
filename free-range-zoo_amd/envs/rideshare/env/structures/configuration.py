"""Rideshare configuration dataclasses.

Mirror of free_range_zoo/envs/rideshare/env/structures/configuration.py: same class names, field names and ValueError
conditions (RewardConfiguration :12-59, PassengerConfiguration :62-80, AgentConfiguration :83-110,
RideshareConfiguration :113-174).  ``to_cstruct`` lowers a configuration to ``frz_rideshare_cfg`` + the schedule array.
"""
from dataclasses import dataclass
import functools

import numpy as np
import torch

from free_range_zoo_amd.utils.configuration import Configuration
from free_range_zoo_amd import _capi


def _require(condition: bool, message: str) -> None:
    if not condition:
        raise ValueError(message)


@dataclass
class RewardConfiguration(Configuration):
    """Costs / rewards: move, drop, noop, accept, pool limit, waiting costs (pick_cost and use_pooling_rewards are carried but,
    as in the reference's step, unused)."""
    pick_cost: float
    move_cost: float
    drop_cost: float
    noop_cost: float
    accept_cost: float
    pool_limit_cost: float
    use_pooling_rewards: bool
    use_variable_move_cost: bool
    use_waiting_costs: bool
    wait_limit: torch.IntTensor
    long_wait_time: int
    general_wait_cost: float
    long_wait_cost: float

    def validate(self):
        _require(len(self.wait_limit) == 3, 'Wait limit should have three elements.')
        _require(bool(self.wait_limit.min() > 0), 'Wait limit elements should all be greater than 0.')
        _require(self.long_wait_time > 0, 'Long wait time should be greater than 0.')
        return True


@dataclass
class PassengerConfiguration(Configuration):
    """schedule: int tensor [tasks, (timestep, env or -1 for every env, y, x, y_dest, x_dest, fare)]."""
    schedule: torch.IntTensor

    def validate(self):
        _require(len(self.schedule.shape) == 2, 'Schedule should be a 2D tensor')
        _require(self.schedule.shape[-1] == 7, 'Schedule should have 7 elements in the last dimesion.')
        return True


@dataclass
class AgentConfiguration(Configuration):
    """start_positions [A, 2], pool_limit, travel model switches."""
    start_positions: torch.IntTensor
    pool_limit: int
    use_diagonal_travel: bool
    use_fast_travel: bool

    @functools.cached_property
    def num_agents(self) -> int:
        return self.start_positions.shape[0]

    def validate(self) -> bool:
        _require(self.pool_limit > 0, 'Pool limit must be greater than 0')
        return True


@dataclass
class RideshareConfiguration(Configuration):
    """Top-level rideshare configuration."""
    grid_height: int
    grid_width: int
    agent_config: AgentConfiguration
    passenger_config: PassengerConfiguration
    reward_config: RewardConfiguration

    @functools.cached_property
    def max_fare(self) -> int:
        return self.passenger_config.schedule[:, 6].max().item()

    def validate(self) -> bool:
        super().validate()
        _require(self.grid_width >= 1, 'grid_width should be greater than 0')
        _require(self.grid_height >= 1, 'grid_height should be greater than 0')
        return True


def default_max_passengers(schedule: np.ndarray, parallel_envs: int) -> int:
    """Upper bound on simultaneously live passengers of one env: everything scheduled for it (wildcards + its own rows)."""
    wild = int((schedule[:, 1] == -1).sum())
    own = schedule[(schedule[:, 1] >= 0) & (schedule[:, 1] < parallel_envs), 1]
    most = int(np.bincount(own).max()) if own.size else 0
    return max(1, wild + most)


def to_cstruct(configuration, parallel_envs: int, max_steps, max_passengers: int = None, track_cumulative_rewards: bool = True,
               first_env_index: int = 0):
    """Lower a (reference-shaped) RideshareConfiguration -> (frz_rideshare_cfg, schedule int32 [S, 7] numpy array)."""
    agent, reward = configuration.agent_config, configuration.reward_config
    schedule = np.ascontiguousarray(configuration.passenger_config.schedule.detach().cpu().numpy().astype(np.int32))
    A = int(agent.start_positions.shape[0])
    if A > _capi.DEFINES['FRZ_MAX_AGENTS']:
        raise ValueError('too many agents for frz_rideshare_cfg')
    if max_passengers is None:
        max_passengers = default_max_passengers(schedule, parallel_envs)
    if max_passengers > _capi.DEFINES['FRZ_MAX_PASSENGERS']:
        raise ValueError(f'more than {_capi.DEFINES["FRZ_MAX_PASSENGERS"]} passenger slots per env are not supported; '
                         f'pass max_passengers=<bound on simultaneously live passengers>')
    c = _capi.frz_rideshare_cfg()
    c.parallel_envs, c.grid_height, c.grid_width, c.num_agents = int(parallel_envs), int(configuration.grid_height), int(
        configuration.grid_width), A
    c.max_steps = -1 if max_steps is None else int(max_steps)
    c.max_passengers = int(max_passengers)
    c.pool_limit = int(agent.pool_limit)
    c.use_fast_travel, c.use_diagonal_travel = int(agent.use_fast_travel), int(agent.use_diagonal_travel)
    c.use_variable_move_cost, c.use_waiting_costs = int(reward.use_variable_move_cost), int(reward.use_waiting_costs)
    c.track_cumulative_rewards = int(track_cumulative_rewards)
    limits = reward.wait_limit.detach().cpu().tolist()
    for i in range(3):
        c.wait_limit[i] = int(limits[i])
    c.long_wait_time = int(reward.long_wait_time)
    c.move_cost, c.drop_cost, c.noop_cost, c.accept_cost = reward.move_cost, reward.drop_cost, reward.noop_cost, reward.accept_cost
    c.pool_limit_cost, c.general_wait_cost, c.long_wait_cost = reward.pool_limit_cost, reward.general_wait_cost, reward.long_wait_cost
    positions = agent.start_positions.detach().cpu().to(torch.int64)
    for a in range(A):
        c.start_y[a], c.start_x[a] = int(positions[a, 0]), int(positions[a, 1])
    c.schedule_rows = int(schedule.shape[0])
    c.first_env_index = int(first_env_index)
    return c, schedule
