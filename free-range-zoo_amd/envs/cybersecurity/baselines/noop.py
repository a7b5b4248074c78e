"""Agent that always performs a no-op action (free_range_zoo/envs/cybersecurity/baselines/noop.py:8-28)."""
import torch

from free_range_zoo_amd.utils.agent import Agent


class NoopBaseline(Agent):
    """``[0, -1]`` for every parallel environment."""

    def act(self, action_space) -> torch.Tensor:
        device = getattr(action_space, 'task_counts', torch.zeros(0)).device
        actions = torch.zeros((self.parallel_envs, 2), dtype=torch.int32, device=device)
        actions[:, 1] = -1
        return actions
