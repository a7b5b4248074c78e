"""Agent-environment-cycle view of a batched parallel env (the reference's native API: utils/env.py:203-242, BatchedAECEnv).

The reference's envs ARE AEC envs — agents act one after the other, the simulation steps when the last one has acted — and its
parallel API is an adapter on top (utils/conversions.py:59-99).  Here the parallel env is the native object (one kernel launch per
step) and this view supplies the AEC protocol around it for code written against ``<domain>_v0.env(...)``: ``agent_selection``,
``step(actions_of_the_selected_agent)``, ``last()``, ``observe(agent)``, ``agent_iter()``.  Everything else is the parallel env's.
"""
from typing import Any, Dict, Iterator, Optional

import torch


class BatchedAECView:

    def __init__(self, env):
        self.env = env
        self._index = 0
        self._pending: Dict[str, torch.Tensor] = {}
        self._between = False  # True while a cycle is open: rewards read as zero (utils/env.py:215 clears them on every agent step)

    def __getattr__(self, name):
        return getattr(self.env, name)

    # ------------------------------------------------------------------------------------------------- protocol
    @property
    def agent_selection(self) -> str:
        return self.env.agents[self._index]

    @property
    def aec_env(self):
        return self

    def reset(self, seed=None, options: Optional[Dict[str, Any]] = None) -> None:
        self.env.reset(seed=seed, options=options)
        self._index, self._pending, self._between = 0, {}, False

    @property
    def rewards(self) -> Dict[str, torch.Tensor]:
        if self._between:
            return {agent: torch.zeros_like(value) for agent, value in self.env.rewards.items()}
        return self.env.rewards

    @torch.no_grad()
    def step(self, actions: torch.Tensor) -> None:
        """Record the selected agent's ``[parallel_envs, 2]`` actions; the simulation steps once the last agent has acted."""
        env, agent = self.env, self.agent_selection
        # an agent that is terminated or truncated in every env no longer steps, and the selection stays put (utils/env.py:211-213)
        if bool(torch.all(env.terminations[agent])) or bool(torch.all(env.truncations[agent])):
            return
        self._pending[agent] = actions
        self._between = True
        if self._index == len(env.agents) - 1:
            env.step({name: self._pending[name] for name in env.agents})
            self._pending, self._between = {}, False
        self._index = (self._index + 1) % len(env.agents)

    def observe(self, agent: str):
        return self.env.observe(agent)

    def last(self, observe: bool = True):
        """``(observation, cumulative reward, terminations, truncations, info)`` of the selected agent (pettingzoo AECEnv.last)."""
        agent = self.agent_selection
        observation = self.env.observe(agent) if observe else None
        return (observation, self.env._cumulative_rewards[agent], self.env.terminations[agent], self.env.truncations[agent],
                self.env.infos.get(agent, {}))

    def agent_iter(self, max_iter: int = 2**63) -> Iterator[str]:
        """The selected agent, again and again, until every env is finished (or ``max_iter`` selections)."""
        count = 0
        while count < max_iter and not bool(torch.all(self.env.finished)):
            yield self.agent_selection
            count += 1
