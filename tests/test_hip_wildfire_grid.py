"""GPU: the wildfire step for grids above 16 cells (csrc/wildfire_grid.hip: a wavefront per env for the cells, one crew wavefront per four envs for the agents, an output entry per lane for the lists) against the
reference's golden trajectories (run through it with FRZ_WF_KERNEL=grid), against the CPU oracle on grids up to 32 x 32 with 3 .. 16
agents in every RNG mode, and against the env-per-lane kernels on a shape both accept."""
from dataclasses import replace

import numpy as np
import pytest
import torch

import configs
import golden_util as G
from free_range_zoo_amd import _capi
from test_hip_wildfire import compare_snapshots, hip_snapshot, make_env, np_, oracle_snapshot, run_against_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(configs.WILDFIRE_GOLDEN))
def test_golden_trajectories_through_the_grid_kernels(name, monkeypatch):
    """The trajectories recorded from the unmodified reference (2 x 3 and 4 x 5 grids) stepped by the cells-across-lanes kernels."""
    monkeypatch.setenv('FRZ_WF_KERNEL', 'grid')
    build, kwargs = configs.WILDFIRE_GOLDEN[name]
    data = np.load(G.golden_path(f'traj_wildfire_{name}.npz'))
    cfg = G.load_cfg(data, _capi.frz_wildfire_cfg)
    B, A = cfg.parallel_envs, cfg.num_agents
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    assert env._cells_env_major
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    G.compare_wildfire(hip_snapshot(env), data, 'r_', A, f'{name} reset')
    HW = cfg.grid_height * cfg.grid_width
    for t in range(int(data['steps'])):
        p = f's{t}_'
        if bool(data[p + 'stepped']):
            rnd = (torch.from_numpy(data[p + 'field_randomness']), torch.from_numpy(data[p + 'agent_randomness']))
        else:  # reference early-out: nothing drawn; the kernel must not touch anything whatever it is given
            rnd = (torch.zeros(3, B, HW), torch.zeros(5, B, A))
        actions = {agent: torch.from_numpy(data[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)}
        env.step(actions, randomness=rnd)
        G.compare_wildfire(hip_snapshot(env), data, p, A, f'{name} step {t}')
        G.assert_same(np_(env.finished), data[p + 'finished'], f'{name} step {t} finished')
    env.check()


SHAPES = [(5, 5, 3), (7, 9, 5), (8, 8, 12), (15, 15, 4), (16, 16, 6), (20, 13, 16), (32, 32, 12)]  # (15 x 15: four cells per lane with 16-bit scan counts)


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('rng', ['injected', 'philox', 'mt19937'])
def test_grids_match_the_oracle(oracle, shape, rng):
    H, Wd, A = shape
    B = 260 if H * Wd > 256 else 1030  # more than one 256-env chunk of the offsets scan either way; ragged last workgroup
    env, o = run_against_oracle(oracle, lambda: configs.wildfire_grid(H, Wd, A, seed=H + A), {}, B, 12, 14, seed=H * Wd + A, rng=rng)
    assert env._cells_env_major
    assert int(o.env_task_count.max()) > 0


@pytest.mark.parametrize('B', [1, 2, 3, 5, 259])
def test_workgroups_with_fewer_than_four_envs(oracle, B):
    """One crew wavefront serves the (up to) four envs of a workgroup: the last workgroup of these batches has one, two or three, and the
    crew is then one of the wavefronts that exist."""
    run_against_oracle(oracle, lambda: rich_grid(8, 8, 12), dict(show_bad_actions=True, observe_other_suppressant=True), B, 10, 12, seed=40 + B, rng='philox',
                       policy='device')
    run_against_oracle(oracle, lambda: configs.wildfire_grid(16, 16, 6, seed=5), {}, B, 8, 10, seed=50 + B, rng='mt19937')


def rich_grid(H, Wd, A):
    """wildfire_grid with the switches the plain one leaves off: fuel, localized put-outs, scaled burnout penalty."""
    cfg = configs.wildfire_grid(H, Wd, A, seed=3)
    cfg.reward_config = replace(cfg.reward_config, localize_putouts=True, burnout_penalty=0.0, burnout_penalty_scaled=True)
    cfg.stochastic_config = replace(cfg.stochastic_config, fire_fuel=True)
    return cfg


@pytest.mark.parametrize('kwargs', [dict(show_bad_actions=True), dict(observe_other_power=True, observe_other_suppressant=True),
                                    dict(show_bad_actions=True, observe_other_suppressant=True)], ids=['bad', 'observe', 'bad_observe'])
def test_grid_flags_match_the_oracle(oracle, kwargs):
    run_against_oracle(oracle, lambda: rich_grid(9, 11, 7), kwargs, 700, 15, 18, seed=31)
    run_against_oracle(oracle, lambda: rich_grid(6, 6, 4), kwargs, 1300, 15, 18, seed=32, rng='philox', policy='device')


def test_grid_kernels_agree_with_the_lane_kernels(oracle, monkeypatch):
    """A 4 x 5 grid runs either family: same trajectories from the same seeds (state, rewards, lists)."""
    B = 900
    envs = {}
    for family in ('lane', 'grid'):
        monkeypatch.setenv('FRZ_WF_KERNEL', family)
        envs[family] = make_env(configs.wildfire_rich, B, 25, rng='philox', observe_other_suppressant=True)
        envs[family].reset(seed=torch.arange(B, dtype=torch.int32) + 5)
    assert envs['grid']._cells_env_major and not envs['lane']._cells_env_major
    for t in range(28):
        for env in envs.values():
            env.step(env.random_policy_actions(policy_seed=3, policy_step=t))
        compare_snapshots(hip_snapshot(envs['grid']), hip_snapshot(envs['lane']), f'grid vs lane step {t}')
    for env in envs.values():
        env.check()


@pytest.mark.parametrize('rng', ['philox', 'mt19937'])
def test_grid_fused_random_policy_step_equals_policy_then_step(rng):
    B = 1500
    build = lambda: configs.wildfire_grid(8, 8, 12)  # noqa: E731
    two, one = [make_env(build, B, 15, rng=rng, exact_shapes=False) for _ in range(2)]
    for env in (two, one):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 2)
    for t in range(18):
        two.step(two.random_policy_actions(policy_seed=8, policy_step=t))
        live = not bool(one.finished.all())
        one.step_random_policy(policy_seed=8, policy_step=t)
        if live:
            assert torch.equal(two._actions, one._actions), f'actions at step {t}'
        for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards', '_task_offsets', '_task_values', '_act_map_offsets', '_act_map_values'):
            assert torch.equal(getattr(two, name), getattr(one, name)), f'{name} at step {t}'
    one.check()


def test_grid_many_chunks_and_reseeded_resets(oracle):
    """More 256-env chunks than CUs (arrival-order tickets in the offsets launch), and frz_wildfire_reset_reseed on this family."""
    B = 70000
    env = make_env(lambda: configs.wildfire_grid(5, 5, 3), B, 10, rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(4):
        env.step_random_policy(policy_seed=1, policy_step=t)
    off, counts = env._task_offsets, env.environment_task_count
    assert int(off[0]) == 0 and torch.equal(off[1:] - off[:-1], counts) and torch.equal(counts, (env.state().fires > 0).flatten(1).sum(dim=1))
    for a in range(3):
        aoff = env._act_map_offsets[a]
        assert torch.equal(aoff[1:] - aoff[:-1], env.agent_task_count[a].long())
    env.check()
    seeds = env.seeds.clone()
    from free_range_zoo_amd.utils.env import stream_ptr
    _capi.check(env._lib.frz_wildfire_reset_reseed(env._handle, 1000003, stream_ptr(env.device)), 'frz_wildfire_reset_reseed')
    assert torch.equal(env.seeds, seeds + 1000003) and int(env.num_moves.max()) == 0


@pytest.mark.parametrize('which', ['wildfire_grid', 'wildfire_lane', 'cybersecurity_16'])
def test_frozen_steps_leave_the_mt19937_streams_alone(which):
    """Kernels that stage their MT19937 draws in a generator launch (the grid family, the runtime-shape lane kernel, the 16-node
    cybersecurity variant): once every env is finished a step is a no-op in the reference (utils/env.py:211-213) and draws nothing."""
    from free_range_zoo_amd.envs import cybersecurity_v0, wildfire_v0
    B = 700
    if which == 'cybersecurity_16':
        env = cybersecurity_v0.parallel_env(configuration=configs.cyber_grid(16, 8, 8), parallel_envs=B, max_steps=4, device=torch.device('cuda'))
    else:
        shape = (8, 8, 12) if which == 'wildfire_grid' else (4, 5, 4)
        env = wildfire_v0.parallel_env(configuration=configs.wildfire_grid(*shape), parallel_envs=B, max_steps=4, device=torch.device('cuda'))
    assert env.rng == 'mt19937'
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    positions = []
    for t in range(7):
        env.step(env.random_policy_actions(policy_seed=2, policy_step=t).clone())
        positions.append((env.generator.generator_index.clone(), env.generator.generator_states.clone()))
    assert bool(env.finished.all())
    assert not torch.equal(positions[2][0], positions[3][0])  # live steps draw
    for later in positions[4:]:  # frozen ones do not
        assert torch.equal(later[0], positions[3][0]) and torch.equal(later[1], positions[3][1])
    env.check()


@pytest.mark.parametrize('rng', ['philox', 'mt19937'])
@pytest.mark.parametrize('shape', [(5, 7, 3), (8, 8, 12), (16, 16, 6), (20, 13, 16)], ids=lambda s: 'x'.join(map(str, s)))
def test_overlapped_rollout_equals_the_step_loop(shape, rng, monkeypatch):
    """(FRZ_WG_OVERLAP=1: an option, off by default — it measured slower than the plain sequence)  A rollout of the grid family runs the LISTS launch of step t on a second stream beside the ENV launch of step t + 1 (the scan that
    produces the next step's batch totals stays on the step's stream; what the lists launch reads of the env launch is double-buffered by
    the step's parity: csrc/wildfire_grid.hip launch_cpl).  Same results as one `step_random_policy` call per step — odd and even step
    counts, steps past the horizon (which change nothing: their lists launch must leave the last real step's lists alone), a second
    rollout continuing the first, and a plain step afterwards."""
    monkeypatch.setenv('FRZ_WG_OVERLAP', '1')
    H, Wd, A = shape
    B, horizon = (300 if H * Wd > 256 else 1100), 9
    a, b = [make_env(lambda: configs.wildfire_grid(H, Wd, A, seed=3), B, horizon, rng=rng) for _ in range(2)]
    assert a._cells_env_major
    seeds = torch.arange(B, dtype=torch.int32) + 11
    for env in (a, b):
        env.reset(seed=seeds)
    done = 0
    for n in (4, 3, 5):  # 4 + 3 = 7 steps inside the episode, then 5 more: two real ones and three past the horizon
        a.rollout(n, policy_seed=5, first_step=done)
        for t in range(done, done + n):
            b.step_random_policy(5, t)
        done += n
        compare_snapshots(hip_snapshot(a), hip_snapshot(b), f'{shape} {rng}: after {done} steps')
    assert bool(a.finished.all())
    for env in (a, b):
        env.reset(seed=seeds + 1)
        env.step_random_policy(6, 0)
    compare_snapshots(hip_snapshot(a), hip_snapshot(b), f'{shape} {rng}: a plain step after the rollouts')
    a.check(), b.check()


def test_overlapped_rollout_inside_a_graph(monkeypatch):
    """The same through a captured HIP graph (what bench.py's secondary workloads replay): the second stream joins the capture through the
    scan's event and is joined back before the capture ends."""
    monkeypatch.setenv('FRZ_WG_OVERLAP', '1')
    B, horizon = 2048, 12
    graphed, eager = [make_env(lambda: configs.wildfire_grid(8, 8, 12, seed=1), B, horizon, rng='philox', exact_shapes=False) for _ in range(2)]
    seeds = torch.arange(B, dtype=torch.int32)
    for env in (graphed, eager):
        env.reset(seed=seeds)
    graph = graphed.capture_random_rollout(horizon, policy_seed=9, include_reset=True)
    for _ in range(2):  # the second replay starts from the in-graph reset again
        graphed.seeds.copy_(seeds)
        graph.replay()
    torch.cuda.synchronize()
    for t in range(horizon):
        eager.step_random_policy(9, t)
    for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards', '_task_offsets', '_act_map_offsets', '_obs_self'):
        assert torch.equal(getattr(eager, name), getattr(graphed, name)), name
    total = int(eager._task_offsets[-1])
    assert torch.equal(eager._task_values[:total], graphed._task_values[:total])
    for k in range(len(eager.agents)):
        n = int(eager._act_map_offsets[k, -1])
        assert torch.equal(eager._act_map_values[k, :n], graphed._act_map_values[k, :n])
    graphed.check()
