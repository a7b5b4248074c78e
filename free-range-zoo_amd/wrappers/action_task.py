"""Action-task mapping wrapper (mirrors free_range_zoo/wrappers/action_task.py:10-58).

The reference builds it from supersuit's ``shared_wrapper``: every observation handed to an agent becomes
``(observation, {'agent_action_mapping': env.agent_action_mapping[agent]})`` — which is what the scripted baselines consume.
Here it is a thin object around the parallel env: same attribute access, observations of ``reset`` / ``step`` / ``observe`` paired
with the agent's action mapping (views of the env's persistent buffers, nothing is copied).
"""
from typing import Any, Dict, Tuple


class ActionTaskMappingWrapper:
    """``observation -> (observation, {'agent_action_mapping': mapping})`` for every agent of a parallel env."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):  # everything else is the env's
        return getattr(self.env, name)

    def _pair(self, observations: Dict[str, Any]) -> Dict[str, Tuple[Any, Dict[str, Any]]]:
        mapping = self.env.agent_action_mapping
        return {agent: (obs, {'agent_action_mapping': mapping[agent]}) for agent, obs in observations.items()}

    def reset(self, *args, **kwargs):
        observations, infos = self.env.reset(*args, **kwargs)
        return self._pair(observations), infos

    def step(self, *args, **kwargs):
        observations, rewards, terminations, truncations, infos = self.env.step(*args, **kwargs)
        return self._pair(observations), rewards, terminations, truncations, infos

    def observe(self, agent: str):
        return self.env.observe(agent), {'agent_action_mapping': self.env.agent_action_mapping[agent]}


def action_mapping_wrapper_v0(env, **kwargs) -> ActionTaskMappingWrapper:
    """Apply the action-task mapping wrapper to the environment."""
    return ActionTaskMappingWrapper(env)
