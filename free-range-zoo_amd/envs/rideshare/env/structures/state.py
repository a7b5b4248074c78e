"""Rideshare state (mirrors free_range_zoo/envs/rideshare/env/structures/state.py:10-66)."""
from dataclasses import dataclass
from typing import Optional

import torch

from free_range_zoo_amd.utils.state import State


class _RowsByEnv:
    """Iterates ``table[table[:, 0] == env]`` for env = 0 .. B-1; evaluated by the consumer (the table may be a pending host copy)."""

    def __init__(self, table: torch.Tensor, parallel_envs: int):
        self.table, self.parallel_envs = table, parallel_envs

    def __iter__(self):
        for env in range(self.parallel_envs):
            yield self.table[self.table[:, 0] == env]


@dataclass
class RideshareState(State):
    """
    agents:     int32 [B, A, 2]  (y, x)
    passengers: int32 [P, 11]    global table sorted by env, entry order inside an env; columns
                (env, y, x, y_dest, x_dest, fare, state{0 unaccepted, 1 accepted, 2 riding}, driver (-1 none),
                 entered_step, accepted_step, picked_step)
    """
    agents: torch.Tensor
    passengers: Optional[torch.Tensor]

    def __len__(self) -> int:
        return self.agents.shape[0]

    def __getitem__(self, indices):
        indices = torch.as_tensor(indices, device=self.agents.device).reshape(-1)
        rows = torch.isin(self.passengers[:, 0], indices.to(self.passengers.dtype))
        return RideshareState(agents=self.agents[indices], passengers=self.passengers[rows])

    def to_dataframe_parts(self, copy=lambda tensor: tensor):
        """The passenger table is logged per env as the rows whose env column matches (structures/state.py:50-66)."""
        return [('agents', copy(self.agents)), ('passengers', _RowsByEnv(copy(self.passengers), self.agents.shape[0]))], []

    def __hash__(self) -> int:
        parts = (self.agents, self.passengers)
        return hash(tuple(tuple(t.detach().cpu().reshape(-1).tolist()) for t in parts))
