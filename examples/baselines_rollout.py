"""The reference's quickstart (docs/source/introduction/quickstart.md) for wildfire with its scripted baselines, on the GPU:
action-task wrapper, one agent object per firefighter, observe -> act -> step, CSV logs of the first envs.
usage: python examples/baselines_rollout.py [parallel_envs] [log_directory]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import configs  # noqa: E402
from free_range_zoo_amd.envs import wildfire_v0  # noqa: E402
from free_range_zoo_amd.envs.wildfire.baselines import NoopBaseline, RandomBaseline, StrongestBaseline, WeakestBaseline  # noqa: E402
from free_range_zoo_amd.wrappers import action_mapping_wrapper_v0  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
log_directory = sys.argv[2] if len(sys.argv) > 2 else None
env = wildfire_v0.parallel_env(configuration=configs.wildfire_rich(), parallel_envs=B, max_steps=30, device=torch.device('cuda'),
                               log_directory=log_directory)
env = action_mapping_wrapper_v0(env)
observations, infos = env.reset(seed=torch.arange(B, dtype=torch.int32), options={'log_description': 'baselines_rollout.py'})
kinds = [StrongestBaseline, WeakestBaseline, RandomBaseline, NoopBaseline]
agents = {name: kinds[i % len(kinds)](name, B) for i, name in enumerate(env.agents)}
totals = {name: torch.zeros(B, device='cuda') for name in agents}
while not torch.all(env.finished):
    for name, agent in agents.items():
        agent.observe(observations[name])
    actions = {name: agents[name].act(action_space=env.action_space(name)) for name in env.agents}
    observations, rewards, terminations, truncations, infos = env.step(actions)
    for name in agents:
        totals[name] += rewards[name]
env.close()
for name, agent in agents.items():
    print(f'{name:16s} {type(agent).__name__:18s} mean episode reward {float(totals[name].mean()):8.2f}')
if log_directory:
    print('logs:', sorted(os.listdir(log_directory))[:4], '...')
