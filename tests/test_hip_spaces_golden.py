"""GPU: `env.action_space(agent)` / `env.observation_space(agent)` of the HIP envs against the objects the unmodified reference handed out
at the reset and after every step of its recorded trajectories (tests/golden/spaces_*.npz: structure — kind, n, start, low, high, members —
as `raw_env.action_space / observation_space` built it; generator: tools/refharness/make_golden.py spaces).  SURVEY §8 rows W13 / R7 / C6."""
import numpy as np
import pytest
import torch

import configs
import golden_util as G
from free_range_zoo_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(configs.WILDFIRE_GOLDEN))
def test_wildfire_spaces_along_the_recorded_trajectory(name):
    from test_hip_wildfire import make_env
    build, kwargs = configs.WILDFIRE_GOLDEN[name]
    traj = np.load(G.golden_path(f'traj_wildfire_{name}.npz'))
    data, table = G.load_spaces('wildfire', name)
    cfg = G.load_cfg(traj, _capi.frz_wildfire_cfg)
    B, A, HW = cfg.parallel_envs, cfg.num_agents, cfg.grid_height * cfg.grid_width
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    G.compare_spaces(env, data, table, 'r_', f'{name} reset')
    for t in range(int(traj['steps'])):
        p = f's{t}_'
        rnd = ((torch.from_numpy(traj[p + 'field_randomness']), torch.from_numpy(traj[p + 'agent_randomness'])) if bool(traj[p + 'stepped'])
               else (torch.zeros(3, B, HW), torch.zeros(5, B, A)))
        env.step({agent: torch.from_numpy(traj[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)}, randomness=rnd)
        G.compare_spaces(env, data, table, p, f'{name} step {t}')


@pytest.mark.parametrize('name', sorted(configs.CYBER_GOLDEN))
def test_cybersecurity_spaces_along_the_recorded_trajectory(name):
    from test_hip_cybersecurity import make_env
    build, kwargs = configs.CYBER_GOLDEN[name]
    traj = np.load(G.golden_path(f'traj_cybersecurity_{name}.npz'))
    data, table = G.load_spaces('cybersecurity', name)
    cfg = G.load_cfg(traj, _capi.frz_cybersecurity_cfg)
    B, A, N = cfg.parallel_envs, cfg.num_attackers + cfg.num_defenders, cfg.num_nodes
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    G.compare_spaces(env, data, table, 'r_', f'{name} reset')
    for t in range(int(traj['steps'])):
        p = f's{t}_'
        rnd = ((torch.from_numpy(traj[p + 'network_randomness']), torch.from_numpy(traj[p + 'agent_randomness'])) if bool(traj[p + 'stepped'])
               else (torch.zeros(1, B, N), torch.zeros(1, B, A)))
        env.step({agent: torch.from_numpy(traj[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)}, randomness=rnd)
        G.compare_spaces(env, data, table, p, f'{name} step {t}')


@pytest.mark.parametrize('name', sorted(configs.RIDESHARE_GOLDEN))
def test_rideshare_spaces_along_the_recorded_trajectory(name):
    from test_hip_rideshare import make_env
    build = configs.RIDESHARE_GOLDEN[name]
    traj = np.load(G.golden_path(f'traj_rideshare_{name}.npz'))
    data, table = G.load_spaces('rideshare', name)
    cfg = G.load_cfg(traj, _capi.frz_rideshare_cfg)
    B = cfg.parallel_envs
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    G.compare_spaces(env, data, table, 'r_', f'{name} reset')
    for t in range(int(traj['steps'])):
        p = f's{t}_'
        env.step({agent: torch.from_numpy(traj[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)})
        G.compare_spaces(env, data, table, p, f'{name} step {t}')
