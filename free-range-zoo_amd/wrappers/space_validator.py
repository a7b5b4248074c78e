"""Action space validator (mirrors free_range_zoo/wrappers/space_validator.py:13-98).

The reference checks every env's action against its ``OneOf`` space in a Python loop before the env sees it and raises
``IndexError`` at the first offender.  Here the check is a handful of elementwise device operations on the count-based batched
space (``BatchedOneOfSpace.invalid_actions``) and one host read of the verdict per step — a debugging aid, like the original.
"""
import logging
from typing import Dict

import torch

logger = logging.getLogger('free_range_zoo')


class ActionSpaceValidatorWrapper:
    """Validates ``step`` actions of every agent against ``env.action_space(agent)``; everything else is the env's."""

    def __init__(self, env, allow_flexible_task_tags: bool = True):
        self.env = env
        self.allow_flexible_task_tags = allow_flexible_task_tags

    def __getattr__(self, name):
        return getattr(self.env, name)

    def _validate(self, actions) -> None:
        agents = list(self.env.agents)
        if not isinstance(actions, dict):
            actions = {agent: actions[a] for a, agent in enumerate(agents)}
        verdicts = torch.stack([self.env.action_space(agent).invalid_actions(actions[agent], self.allow_flexible_task_tags) for agent in agents])
        if bool(verdicts.any()):
            a, b = (int(x) for x in torch.nonzero(verdicts)[0])
            agent = agents[a]
            space = self.env.action_space(agent).spaces[b]
            logger.critical(f'{agent} in batch {b} attempted to take an action on a undefined task.\\n'
                            f'Action: {actions[agent][b].tolist()}\\nSpace: {space}')
            raise IndexError(f'{agent} in batch {b}: action {actions[agent][b].tolist()} is outside {space}')

    def step(self, actions, *args, **kwargs):
        self._validate(actions)
        return self.env.step(actions, *args, **kwargs)


def space_validator_wrapper_v0(env, allow_flexible_task_tags: bool = True) -> ActionSpaceValidatorWrapper:
    """Apply the action space validator to the environment."""
    return ActionSpaceValidatorWrapper(env, allow_flexible_task_tags=allow_flexible_task_tags)
