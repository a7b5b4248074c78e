"""Wildfire state (mirrors free_range_zoo/envs/wildfire/env/structures/state.py:10-68)."""
from dataclasses import dataclass

import torch

from free_range_zoo_amd.utils.state import State


@dataclass
class WildfireState(State):
    """
    fires:        int32 [B, H, W]  >0 lit (required power), <0 unlit / put out / burnt (-type), 0 no fire possible
    intensity:    int32 [B, H, W]  0 .. num_fire_states-1
    fuel:         int32 [B, H, W]  remaining ignitions
    agents:       int32 [A, 2]     (y, x), shared by all envs
    suppressants: float32 [B, A]
    capacity:     float32 [B, A]
    equipment:    int32 [B, A]
    """
    fires: torch.Tensor
    intensity: torch.Tensor
    fuel: torch.Tensor
    agents: torch.Tensor
    suppressants: torch.Tensor
    capacity: torch.Tensor
    equipment: torch.Tensor

    def __post_init__(self):
        super().__post_init__()
        self.metadata = {'shared': ('agents', )}

    def __getitem__(self, indices):
        return WildfireState(fires=self.fires[indices], intensity=self.intensity[indices], fuel=self.fuel[indices], agents=self.agents,
                             suppressants=self.suppressants[indices], capacity=self.capacity[indices],
                             equipment=self.equipment[indices])

    def __hash__(self) -> int:
        parts = (self.fires, self.intensity, self.fuel, self.agents, self.suppressants, self.capacity, self.equipment)
        return hash(tuple(tuple(t.detach().cpu().reshape(-1).tolist()) for t in parts))
