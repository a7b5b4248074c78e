"""State base class (mirrors free_range_zoo/utils/state.py:11-239: clone / save_initial / restore_initial /
save_checkpoint / restore_from_checkpoint / load_state / stack / cat / __len__).

State tensors handed out by the envs are batch-major *views* of the struct-of-arrays HBM buffers the HIP kernels
step in place (e.g. ``fires`` is a ``[B, H, W]`` view with strides ``(1, W*B, B)``), so reading ``env.state()`` costs
nothing and writes through the view are seen by the next ``step()``.
"""
from __future__ import annotations

import copy
from abc import ABC
from dataclasses import dataclass
from typing import List, Optional

import torch


@dataclass
class State(ABC):
    """Batched environment state: a dataclass of tensors whose first dimension is the env batch."""

    def __post_init__(self):
        self.metadata = {}
        self.initial_state = None
        self.checkpoint = None

    # -- helpers ---------------------------------------------------------------------------------------------
    def _tensor_fields(self):
        skip = ('metadata', 'initial_state', 'checkpoint')
        return [(k, v) for k, v in vars(self).items() if k not in skip]

    def _shared(self):
        return tuple(self.metadata.get('shared', ())) if isinstance(self.metadata, dict) else ()

    def to(self, device: torch.device = torch.device('cpu')):
        for name, value in self._tensor_fields():
            if hasattr(value, 'to'):
                setattr(self, name, value.to(device))
        return self

    def clone(self):
        fields = {}
        for name, value in self._tensor_fields():
            fields[name] = value.clone() if hasattr(value, 'clone') else copy.deepcopy(value)
        cloned = self.__class__(**fields)
        cloned.initial_state = self.initial_state.clone() if self.initial_state is not None else None
        cloned.checkpoint = self.checkpoint.clone() if self.checkpoint is not None else None
        cloned.metadata = copy.deepcopy(self.metadata)
        return cloned

    def _restore_from(self, source, batch_indices):
        if source is None:
            raise ValueError('State to restore from is not saved')
        for name, value in source._tensor_fields():
            if not hasattr(value, 'clone'):
                setattr(self, name, value)
            elif batch_indices is None or name in self._shared():
                getattr(self, name).copy_(value) if getattr(self, name, None) is not None else setattr(self, name, value.clone())
            else:
                getattr(self, name)[batch_indices] = value[batch_indices]

    def save_initial(self):
        self.initial_state = None
        self.initial_state = self.clone()

    def restore_initial(self, batch_indices: Optional[torch.Tensor] = None) -> None:
        if self.initial_state is None:
            raise ValueError('Initial state is not saved')
        self._restore_from(self.initial_state, batch_indices)

    def save_checkpoint(self):
        self.checkpoint = None
        self.checkpoint = self.clone()

    def restore_from_checkpoint(self, batch_indices: Optional[torch.Tensor] = None) -> None:
        if self.checkpoint is None:
            raise ValueError('Checkpoint is not saved')
        self._restore_from(self.checkpoint, batch_indices)

    def load_state(self, state, batch_indices: Optional[torch.Tensor] = None) -> None:
        """Copy ``state`` in (row i of ``state`` goes to env ``batch_indices[i]``; everything if no indices)."""
        for name, value in state._tensor_fields():
            if not hasattr(value, 'clone'):
                setattr(self, name, value)
            elif batch_indices is None or name in self._shared():
                getattr(self, name).copy_(value)
            else:
                getattr(self, name)[batch_indices] = value.to(getattr(self, name).device)

    @staticmethod
    def stack(states: List['State'], *args, **kwargs):
        first = states[0]
        fields = {}
        for name, value in first._tensor_fields():
            if name in first._shared() or name == 'agents':
                fields[name] = value
            else:
                fields[name] = torch.stack([getattr(s, name) for s in states], *args, **kwargs)
        return first.__class__(**fields)

    @staticmethod
    def cat(states: List['State'], *args, **kwargs):
        first = states[0]
        fields = {}
        for name, value in first._tensor_fields():
            if name in first._shared() or name == 'agents':
                fields[name] = value
            else:
                fields[name] = torch.cat([getattr(s, name) for s in states], *args, **kwargs)
        return first.__class__(**fields)

    # -- logging (state.py:180-191) --------------------------------------------------------------------------
    def to_dataframe_parts(self, copy=lambda tensor: tensor):
        """``(per-env columns, shared columns)`` as ``(name, tensor)`` lists in the reference's column order; ``copy`` is applied to
        every tensor (the logging tap passes a stream-ordered device-to-host copy)."""
        shared = self._shared()
        columns = [(name, copy(value)) for name, value in self._tensor_fields() if name not in shared]
        return columns, [(name, copy(getattr(self, name))) for name in shared]

    def to_dataframe(self):
        """Convert the state into a dataframe: one row per env, cells are ``str(tensor.tolist())``."""
        import pandas as pd
        columns, shared = self.to_dataframe_parts(lambda tensor: tensor.cpu())
        df = pd.DataFrame({name: [str(row.tolist()) for row in value] for name, value in columns})
        for name, value in shared:
            df[name] = str(value.tolist())
        return df

    def unwrap(self) -> List['State']:
        """Unwrap a set of batched states into one state per env (state.py:193-206)."""
        return [self[index] for index in range(len(self))]

    def __len__(self) -> int:
        for name, value in self._tensor_fields():
            if name not in self._shared() and hasattr(value, 'shape'):
                return value.shape[0]
        return 0
