"""GPU: randomised shapes against the CPU oracle (all three domains, every wildfire kernel family).  The wildfire grid family (cells across lanes, crew + fields, lists an entry per lane) and the
rideshare env launch (crew + fields) are driven over grids, agent counts, batch sizes, flags and RNG modes drawn from a seeded generator —
1 x n and n x 1 grids, widths up to 32 (the stencil's word shifts), one agent, ragged workgroups, more than 64 passenger slots.
FRZ_FUZZ_CASES scales the number of cases (default: a few seconds' worth per domain)."""
import os
from dataclasses import replace

import numpy as np
import pytest
import torch

import configs

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get('FRZ_FUZZ_CASES', '16'))


def wildfire_case(index):
    g = np.random.default_rng(1000 + index)
    H, Wd = int(g.integers(1, 33)), int(g.integers(1, 33))
    if H * Wd < 2:
        Wd = 2
    A = int(g.integers(1, 17))
    flags = {}
    if g.random() < 0.4:
        flags['show_bad_actions'] = True
    if g.random() < 0.5:
        flags['observe_other_suppressant'] = True
    if g.random() < 0.3:
        flags['observe_other_power'] = True
    rich = bool(g.random() < 0.4)
    rng = ['injected', 'philox', 'mt19937'][int(g.integers(0, 3))]
    B = int(g.choice([1, 2, 3, 5, 63, 64, 65, 130, 257, 515]))
    return dict(H=H, Wd=Wd, A=A, flags=flags, rich=rich, rng=rng, B=B, seed=int(g.integers(0, 1000)), policy='device' if rng == 'philox' and g.random() < 0.5 else 'oracle')


def build_wildfire(case):
    cfg = configs.wildfire_grid(case['H'], case['Wd'], case['A'], seed=case['seed'])
    if case['rich']:
        cfg.reward_config = replace(cfg.reward_config, localize_putouts=True, burnout_penalty=0.0, burnout_penalty_scaled=True)
        cfg.stochastic_config = replace(cfg.stochastic_config, fire_fuel=True)
    return cfg


@pytest.mark.parametrize('index', range(CASES))
def test_wildfire_grid_family_on_random_shapes(oracle, index, monkeypatch):
    from test_hip_wildfire import run_against_oracle
    monkeypatch.setenv('FRZ_WF_KERNEL', 'grid')
    case = wildfire_case(index)
    env, o = run_against_oracle(oracle, lambda: build_wildfire(case), case['flags'], case['B'], 9, 11, seed=case['seed'], rng=case['rng'], policy=case['policy'])
    assert env._cells_env_major, case


def rideshare_case(index):
    g = np.random.default_rng(5000 + index)
    A = int(g.integers(1, 9))
    grid = int(g.integers(3, 13))
    per_step = int(g.integers(1, 5))
    return dict(A=A, grid=grid, per_step=per_step, steps=int(g.integers(6, 16)), seed=int(g.integers(0, 1000)), B=int(g.choice([1, 2, 3, 5, 64, 65, 257, 600])),
                waiting=bool(g.random() < 0.5), diagonal=bool(g.random() < 0.3), fast=bool(g.random() < 0.2), pool=int(g.integers(1, 5)),
                contest=float(g.choice([0.0, 0.3, 0.8])), specific=int(g.integers(0, 2)))


def build_rideshare(case):
    over = dict(use_waiting_costs=True, wait_limit=torch.tensor([1, 2, 3]), long_wait_time=4) if case['waiting'] else {}
    cfg = configs.rideshare_busy(A=case['A'], steps=case['steps'], per_step=case['per_step'], grid=case['grid'], seed=case['seed'], env_specific=case['specific'],
                                 B=min(case['B'], 50), **over)
    cfg.agent_config = replace(cfg.agent_config, use_diagonal_travel=case['diagonal'], use_fast_travel=case['fast'], pool_limit=case['pool'])
    return cfg


@pytest.mark.parametrize('index', range(CASES))
def test_rideshare_on_random_configurations(oracle, index):
    from test_hip_rideshare import run_against_oracle
    case = rideshare_case(index)
    run_against_oracle(oracle, lambda: build_rideshare(case), case['B'], case['steps'] + 2, case['steps'] + 4, seed=case['seed'], contest=case['contest'])


def small_wildfire_case(index):
    g = np.random.default_rng(9000 + index)
    H, Wd = int(g.integers(1, 7)), int(g.integers(1, 7))
    while H * Wd < 2 or H * Wd > 24:
        H, Wd = int(g.integers(1, 7)), int(g.integers(1, 7))
    A = int(g.integers(1, 9))
    flags = {}
    if g.random() < 0.4:
        flags['show_bad_actions'] = True
    if g.random() < 0.5:
        flags['observe_other_suppressant'] = True
    if g.random() < 0.3:
        flags['observe_other_power'] = True
    rng = ['injected', 'philox', 'mt19937'][int(g.integers(0, 3))]
    return dict(H=H, Wd=Wd, A=A, flags=flags, rich=bool(g.random() < 0.4), rng=rng, B=int(g.choice([1, 2, 63, 255, 256, 257, 600, 1500])), seed=int(g.integers(0, 1000)),
                policy='device' if rng == 'philox' and g.random() < 0.5 else 'oracle', kernel=['', 'lane', 'grid'][int(g.integers(0, 3))])


@pytest.mark.parametrize('index', range(CASES))
def test_small_wildfire_grids_in_whatever_family_serves_them(oracle, index, monkeypatch):
    """Grids of at most 24 cells / 8 agents: the field / crew kernels of the exact shapes, the lane-per-env kernels, or the grid family
    (FRZ_WF_KERNEL picks among those that accept the shape)."""
    from test_hip_wildfire import run_against_oracle
    case = small_wildfire_case(index)
    if case['kernel']:
        monkeypatch.setenv('FRZ_WF_KERNEL', case['kernel'])
    run_against_oracle(oracle, lambda: build_wildfire(case), case['flags'], case['B'], 9, 11, seed=case['seed'], rng=case['rng'], policy=case['policy'])


def cyber_case(index):
    g = np.random.default_rng(13000 + index)
    N, Att, D = int(g.integers(1, 17)), int(g.integers(1, 9)), int(g.integers(1, 9))
    flags = dict(observe_other_location=bool(g.random() < 0.5), observe_other_presence=bool(g.random() < 0.5), observe_other_power=bool(g.random() < 0.5),
                 partially_observable=bool(g.random() < 0.5), show_bad_actions=bool(g.random() < 0.5))
    return dict(N=N, Att=Att, D=D, flags=flags, rng=['injected', 'philox', 'mt19937'][int(g.integers(0, 3))], B=int(g.choice([1, 2, 63, 255, 256, 257, 900])),
                seed=int(g.integers(0, 1000)), kernel=['', 'lane'][int(g.integers(0, 2))])


@pytest.mark.parametrize('index', range(CASES))
def test_cybersecurity_on_random_networks(oracle, index, monkeypatch):
    from test_hip_cybersecurity import run_against_oracle
    case = cyber_case(index)
    if case['kernel']:
        monkeypatch.setenv('FRZ_CY_KERNEL', case['kernel'])
    run_against_oracle(oracle, lambda: configs.cyber_grid(case['N'], case['Att'], case['D'], seed=case['seed']), case['flags'], case['B'], 8, 10, seed=case['seed'],
                       rng=case['rng'])


def rollout_case(index):
    g = np.random.default_rng(17000 + index)
    H, Wd = int(g.integers(1, 5)), int(g.integers(1, 5))
    while H * Wd < 2:
        H, Wd = int(g.integers(1, 5)), int(g.integers(1, 5))
    A = int(g.integers(1, 5))
    kwargs = {}
    if g.random() < 0.4:
        kwargs['show_bad_actions'] = True
    if g.random() < 0.5:
        kwargs['observe_other_suppressant'] = True
    seed = int(g.integers(0, 1000))
    max_steps = int(g.integers(3, 14))
    return dict(build=lambda: configs.wildfire_grid(H, Wd, A, seed=seed), B=int(g.choice([1, 2, 255, 256, 257, 1025, 2500])), max_steps=max_steps,
                steps=int(g.integers(2, max_steps + 4)), kwargs=kwargs, shape=(H, Wd, A)), [None, False][int(g.integers(0, 2))]


@pytest.mark.parametrize('index', range(CASES))
def test_policy_rollouts_with_every_record_on_random_small_shapes(oracle, index, monkeypatch):
    """`env.rollout(n, record=True)` — one multi-step launch where the library has one for the shape, per-step launches otherwise or when
    the device is not declared exclusive — against the oracle at every step (sampled actions, rewards, flags, the packed lists)."""
    from test_hip_rollouts import check_policy_rollout_against_the_oracle
    case, one_launch = rollout_case(index)
    if index % 2 == 0:  # half of the cases: the multi-step kernel of a runtime shape wherever it exists, not only where the library prefers it
        monkeypatch.setenv('FRZ_WF_MULTI_STEP', 'all')
    check_policy_rollout_against_the_oracle(oracle, case, one_launch)
