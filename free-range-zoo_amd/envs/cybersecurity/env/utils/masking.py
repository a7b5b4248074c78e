"""Observation column masks (mirrors free_range_zoo/envs/cybersecurity/env/utils/masking.py:6-28)."""
from functools import lru_cache

import torch


@lru_cache(maxsize=100)
def mask_observation(agent_name: str, observe_other_power: bool, observe_other_presence: bool, observe_other_location: bool):
    """Columns of the ``others`` observation an agent keeps: defenders (power, presence, location), attackers (power, presence)."""
    kind = agent_name.split('_')[0]
    if kind == 'defender':
        return torch.tensor([observe_other_power, observe_other_presence, observe_other_location], dtype=torch.bool)
    if kind == 'attacker':
        return torch.tensor([observe_other_power, observe_other_presence], dtype=torch.bool)
    return None
