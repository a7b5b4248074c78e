"""Shared body of the strongest / weakest fire baselines: one launch of ``frz_wildfire_extreme_fire_policy`` per observation."""
from typing import Any, Dict, Tuple

import torch

from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.agent import Agent
from free_range_zoo_amd.utils.env import stream_ptr


def _jagged_parts(nested: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(values, offsets int64 [B + 1], lengths int64 [B]) of a jagged nested tensor (with or without explicit lengths)."""
    offsets = nested.offsets().to(torch.int64)
    lengths = nested.lengths()
    lengths = (offsets[1:] - offsets[:-1]) if lengths is None else lengths.to(torch.int64)
    return nested.values(), offsets.contiguous(), lengths.contiguous()


class ExtremeFireBaseline(Agent):
    """Always fight the strongest (``weakest=False``) or weakest fire among the tasks the reference's baseline looks at.

    Consumes what the action-task mapping wrapper hands out: ``observe((observation, {'agent_action_mapping': mapping}))``.
    Ties are broken uniformly from a Philox stream ``(seed, decision counter, env)`` (the reference draws them from torch's
    global generator); everything else is the reference's behaviour as written (csrc/wildfire_baselines.hip).
    """
    weakest = False

    def __init__(self, *args, seed: int = 0, first_env_index: int = 0, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.seed = seed
        self.first_env_index = first_env_index
        self.decisions = 0
        self.actions = None

    def act(self, action_space) -> torch.Tensor:
        return self.actions

    def observe(self, observation: Tuple[Dict[str, Any], Dict[str, torch.Tensor]]) -> None:
        self.observation, mapping = observation
        self.t_mapping = mapping['agent_action_mapping']
        tasks, obs_self = self.observation['tasks'], self.observation['self']
        device = obs_self.device
        if device.type != 'cuda':
            raise ValueError('the baselines run on the GPU the env lives on (no CPU fallback)')
        task_values, task_offsets, _ = _jagged_parts(tasks)
        _, map_offsets, map_lengths = _jagged_parts(self.t_mapping)
        if self.actions is None or self.actions.device != device:
            self.actions = torch.zeros((self.parallel_envs, 2), dtype=torch.int32, device=device)
        task_values = task_values.to(torch.int64).contiguous()
        if task_values.numel() == 0:  # no lit cell in any env: an empty tensor has no address to hand over
            task_values = torch.zeros((1, 4), dtype=torch.int64, device=device)
        obs_self = obs_self.to(torch.float32).contiguous()
        self._keepalive = (task_values, task_offsets, map_offsets, map_lengths, obs_self)
        _capi.check(_capi.lib().frz_wildfire_extreme_fire_policy(task_values.data_ptr(), task_offsets.data_ptr(), map_offsets.data_ptr(),
                                                                 map_lengths.data_ptr(), obs_self.data_ptr(), self.parallel_envs,
                                                                 int(self.weakest), self.seed, self.decisions, self.first_env_index,
                                                                 self.actions.data_ptr(), stream_ptr(device)),
                    'frz_wildfire_extreme_fire_policy')
        self.decisions += 1
