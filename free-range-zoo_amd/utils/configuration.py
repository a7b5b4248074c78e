"""Configuration base class (mirrors free_range_zoo/utils/configuration.py:9-44)."""
from abc import ABC, abstractmethod
from dataclasses import dataclass

import torch


@dataclass
class Configuration(ABC):
    """Nested dataclass of environment settings; validated on construction, movable between devices."""

    @abstractmethod
    def validate(self) -> bool:
        """Validate nested configurations (subclasses extend this and raise ValueError on inconsistency)."""
        for value in vars(self).values():
            if hasattr(value, 'validate'):
                value.validate()
        return True

    def to(self, device: torch.device = torch.device('cpu')):
        """Move every tensor / nested configuration to ``device`` in place and return self."""
        for name, value in list(vars(self).items()):
            if hasattr(value, 'to'):
                setattr(self, name, value.to(device))
        return self

    def __post_init__(self):
        self.validate()
