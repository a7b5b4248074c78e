"""Cybersecurity state (mirrors free_range_zoo/envs/cybersecurity/env/structures/state.py:12-53)."""
from dataclasses import dataclass

import torch

from free_range_zoo_amd.utils.state import State


@dataclass
class CybersecurityState(State):
    """
    network_state: int32 [B, N]   0 = best .. num_states-1
    location:      int32 [B, D]   node of each defender, -1 = home
    presence:      bool  [B, A]   attackers first, then defenders
    """
    network_state: torch.Tensor
    location: torch.Tensor
    presence: torch.Tensor

    def __getitem__(self, indices):
        return CybersecurityState(network_state=self.network_state[indices], location=self.location[indices],
                                  presence=self.presence[indices])

    def __hash__(self) -> int:
        parts = (self.network_state, self.location, self.presence)
        return hash(tuple(tuple(t.detach().cpu().reshape(-1).tolist()) for t in parts))
