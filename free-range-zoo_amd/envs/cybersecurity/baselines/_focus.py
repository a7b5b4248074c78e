"""Shared body of the patched / exploited / camp baselines: one launch of ``frz_cybersecurity_focus_policy`` per observation."""
from typing import Any, Dict, Optional, Tuple

import torch

from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.agent import Agent
from free_range_zoo_amd.utils.env import stream_ptr

KINDS = ('patched_attacker', 'exploited_attacker', 'patched_defender', 'exploited_defender', 'camp_defender')


class FocusPolicyBaseline(Agent):
    """Pick a target from ``observation['tasks'][:, 0]``, act on it for three steps, pick again (the reference's stateful agents).

    Consumes what the action-task mapping wrapper hands out: ``observe((observation, {'agent_action_mapping': mapping}))``.  The
    agent's state — ``target_node``, ``time_focused``, ``actions`` — lives on the env's GPU and is updated by the kernel.  Ties are
    broken uniformly from a Philox stream ``(seed, decision counter, env)`` (torch's global generator in the reference);
    ``observe(..., tie_draws=int64 [B])`` replays given draws.  ``across_nodes=True`` reads feature 0 of every node instead of the
    reference's ``tasks[:, 0]`` (the features of node 0); the default is the reference's behaviour as written
    (csrc/cybersecurity_baselines.hip).
    """
    kind = 'patched_attacker'
    mapping_key = 'agent_action_mapping'

    def __init__(self, *args, seed: int = 0, first_env_index: int = 0, across_nodes: bool = False, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.seed = seed
        self.first_env_index = first_env_index
        self.across_nodes = across_nodes
        self.subnetwork_states = 0
        self.decisions = 0
        self.target_node = self.time_focused = self.actions = None

    def _state_on(self, device: torch.device) -> None:
        if self.actions is None or self.actions.device != device:
            self.target_node = torch.full((self.parallel_envs, ), -1, dtype=torch.int32, device=device)
            self.time_focused = torch.zeros((self.parallel_envs, ), dtype=torch.int32, device=device)
            self.actions = torch.zeros((self.parallel_envs, 2), dtype=torch.int32, device=device)

    def act(self, action_space) -> torch.Tensor:
        return self.actions

    def _camp_target(self, nodes: int) -> int:
        return 0

    def observe(self, observation: Tuple[Dict[str, Any], Dict[str, torch.Tensor]], tie_draws: Optional[torch.Tensor] = None) -> None:
        self.observation, mapping = observation
        self.t_mapping = mapping[self.mapping_key] if self.mapping_key in mapping else mapping['agent_action_mapping']
        tasks, obs_self = self.observation['tasks'], self.observation['self']
        device = obs_self.device
        if device.type != 'cuda':
            raise ValueError('the baselines run on the GPU the env lives on (no CPU fallback)')
        self._state_on(device)
        tasks = tasks.to(torch.int64).contiguous()
        obs_self = obs_self.to(torch.float32).contiguous()
        B, N, F = tasks.shape
        row = (N * F, F, N) if self.across_nodes else (N * F, 1, F)
        draws = None if tie_draws is None else tie_draws.to(device=device, dtype=torch.int64).contiguous()
        self._keepalive = (tasks, obs_self, draws)
        _capi.check(_capi.lib().frz_cybersecurity_focus_policy(tasks.data_ptr(), row[0], row[1], row[2], obs_self.data_ptr(), obs_self.shape[1],
                                                               self.parallel_envs, KINDS.index(self.kind), self.subnetwork_states,
                                                               self._camp_target(N), int(self.t_mapping.numel()), self.seed, self.decisions,
                                                               self.first_env_index, None if draws is None else draws.data_ptr(),
                                                               self.target_node.data_ptr(), self.time_focused.data_ptr(),
                                                               self.actions.data_ptr(), stream_ptr(device)), 'frz_cybersecurity_focus_policy')
        self.decisions += 1
