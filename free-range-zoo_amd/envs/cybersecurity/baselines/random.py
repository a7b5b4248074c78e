"""Agent that samples the actions available to it uniformly (free_range_zoo/envs/cybersecurity/baselines/random.py)."""
import torch

from free_range_zoo_amd.utils.agent import Agent


class RandomBaseline(Agent):
    """``action_space.sample_nested()``: a device-side uniform draw per env from the count-based space (utils/spaces.py)."""

    def act(self, action_space) -> torch.Tensor:
        return action_space.sample_nested().to(torch.int32)
