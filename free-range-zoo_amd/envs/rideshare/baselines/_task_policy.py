"""Shared body of the greedy / FIFO rideshare baselines: one launch of ``frz_rideshare_task_policy`` per observation."""
from typing import Any, Dict, Optional, Tuple

import torch

from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.agent import Agent
from free_range_zoo_amd.utils.env import stream_ptr

KINDS = ('greedy_focus', 'greedy_global', 'fifo_focus', 'fifo_global')


def jagged_parts(nested: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(values, offsets int64 [B + 1], lengths int64 [B]) of a jagged nested tensor (with or without explicit lengths)."""
    offsets = nested.offsets().to(torch.int64)
    lengths = nested.lengths()
    lengths = (offsets[1:] - offsets[:-1]) if lengths is None else lengths.to(torch.int64)
    return nested.values(), offsets.contiguous(), lengths.contiguous()


class TaskPolicyBaseline(Agent):
    """Pick the passenger with the smallest key (total distance to completion / arrival step) among the tasks the reference's
    agents look at, optionally sticking to the passenger in progress, and answer accept / pick / drop by its state.

    Consumes what the action-task mapping wrapper hands out: ``observe((observation, {'agent_action_mapping': mapping}))``.
    Ties are broken uniformly from a Philox stream ``(seed, decision counter, env)`` (the reference draws them from torch's
    global generator); ``observe(..., tie_draws=int64 [B])`` replays given draws instead.  Everything else is the reference's
    behaviour as written (csrc/rideshare_baselines.hip).
    """
    kind = 'greedy_focus'

    def __init__(self, *args, agent_configuration=None, seed: int = 0, first_env_index: int = 0, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        if self.kind.startswith('greedy') and agent_configuration is None:
            raise ValueError('the greedy agents need agent_configuration (use_diagonal_travel decides the distance)')
        self.use_diagonal_travel = bool(getattr(agent_configuration, 'use_diagonal_travel', False))
        self.seed = seed
        self.first_env_index = first_env_index
        self.decisions = 0
        self.actions = None

    def act(self, action_space) -> torch.Tensor:
        return self.actions

    def observe(self, observation: Tuple[Dict[str, Any], Dict[str, torch.Tensor]], tie_draws: Optional[torch.Tensor] = None) -> None:
        self.observation, mapping = observation
        self.t_mapping = mapping['agent_action_mapping']
        tasks, obs_self = self.observation['tasks'], self.observation['self']
        device = obs_self.device
        if device.type != 'cuda':
            raise ValueError('the baselines run on the GPU the env lives on (no CPU fallback)')
        task_values, task_offsets, task_lengths = jagged_parts(tasks)
        _, _, map_lengths = jagged_parts(self.t_mapping)
        if self.actions is None or self.actions.device != device:
            self.actions = torch.zeros((self.parallel_envs, 2), dtype=torch.int32, device=device)
        task_values = task_values.to(torch.int32).contiguous()
        if task_values.numel() == 0:
            task_values = torch.zeros((1, 8), dtype=torch.int32, device=device)
        obs_self = obs_self.to(torch.int32).contiguous()
        draws = None if tie_draws is None else tie_draws.to(device=device, dtype=torch.int64).contiguous()
        self._keepalive = (task_values, task_offsets, task_lengths, map_lengths, obs_self, draws)
        _capi.check(_capi.lib().frz_rideshare_task_policy(task_values.data_ptr(), task_offsets.data_ptr(), task_lengths.data_ptr(),
                                                          map_lengths.data_ptr(), obs_self.data_ptr(), self.parallel_envs,
                                                          KINDS.index(self.kind), int(self.use_diagonal_travel), self.seed, self.decisions,
                                                          self.first_env_index, None if draws is None else draws.data_ptr(),
                                                          self.actions.data_ptr(), stream_ptr(device)), 'frz_rideshare_task_policy')
        self.decisions += 1
