"""Generic interface for agents (mirrors free_range_zoo/utils/agent.py:9-42)."""
from abc import ABC
from typing import Any, Dict

import torch


class Agent(ABC):
    """Generic interface for agents: ``observe(observation)`` then ``act(action_space)`` -> actions ``[parallel_envs, 2]``."""

    def __init__(self, agent_name: str, parallel_envs: int) -> None:
        self.agent_name = agent_name
        self.parallel_envs = parallel_envs

    def act(self, action_space) -> torch.Tensor:
        """Return the actions, one ``[index, action]`` pair per parallel environment."""

    def observe(self, observation: Dict[str, Any]) -> None:
        """Observe the environment."""
