"""Scripted wildfire baselines on the GPU: frz_wildfire_extreme_fire_policy vs the oracle (bit-exact, same
tie-break stream), vs the answers recorded from the reference's own baselines, and through the action-task wrapper in a rollout."""
import ctypes

import numpy as np
import pytest
import torch

import configs
from test_oracle_wildfire import _baseline_cases, check_extreme_answer

pytestmark = pytest.mark.gpu


def hip_extreme(task_values, task_offsets, map_offsets, map_lengths, obs_self, weakest, seed, step, first_env=0):
    from free_range_zoo_amd import _capi
    dev = torch.device('cuda')
    tv = torch.as_tensor(np.ascontiguousarray(task_values, np.int64).reshape(-1, 4), device=dev)
    if tv.shape[0] == 0:
        tv = torch.zeros((1, 4), dtype=torch.int64, device=dev)
    to, mo = torch.as_tensor(task_offsets, device=dev), torch.as_tensor(map_offsets, device=dev)
    ml, ob = torch.as_tensor(map_lengths, device=dev), torch.as_tensor(np.ascontiguousarray(obs_self, np.float32), device=dev)
    B = ml.shape[0]
    out = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    _capi.check(_capi.lib().frz_wildfire_extreme_fire_policy(tv.data_ptr(), to.data_ptr(), mo.data_ptr(), ml.data_ptr(), ob.data_ptr(), B,
                                                             int(weakest), seed, step, first_env, out.data_ptr(),
                                                             torch.cuda.current_stream().cuda_stream), 'frz_wildfire_extreme_fire_policy')
    return out.cpu().numpy()


def test_recorded_reference_answers_and_oracle(oracle):
    for i, case in _baseline_cases():
        counts, lengths = case['task_counts'], case['map_lengths']
        task_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        map_offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        for kind in ('strongest', 'weakest'):
            got = hip_extreme(case['task_values'], task_offsets, map_offsets, lengths, case['obs_self'], kind == 'weakest', 5, i, first_env=3)
            check_extreme_answer(got, case, kind, f'case {i} {kind}')
            want = oracle.wildfire_extreme_policy(case['task_values'], task_offsets, map_offsets, lengths, case['obs_self'], kind == 'weakest',
                                                  seed=5, step=i, first_env=3)
            assert np.array_equal(got, want), f'case {i} {kind}: HIP vs oracle'


def test_many_ties_large_batch_vs_oracle(oracle):
    rng = np.random.default_rng(4)
    B = 70001
    counts = rng.integers(0, 7, B).astype(np.int64)
    lengths = np.minimum(counts, rng.integers(0, 7, B)).astype(np.int64)
    values = np.zeros((int(counts.sum()), 4), np.int64)
    values[:, 3] = rng.integers(1, 4, values.shape[0])  # few distinct intensities: many ties
    obs_self = rng.random((B, 4)).astype(np.float32)
    obs_self[rng.random(B) < 0.3, 3] = 0.0
    task_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    map_offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    for weakest in (False, True):
        got = hip_extreme(values, task_offsets, map_offsets, lengths, obs_self, weakest, 99, 7, first_env=1000)
        want = oracle.wildfire_extreme_policy(values, task_offsets, map_offsets, lengths, obs_self, weakest, seed=99, step=7, first_env=1000)
        assert np.array_equal(got, want)
    empty = hip_extreme(values, task_offsets, np.zeros(B + 1, np.int64), np.zeros(B, np.int64), obs_self, False, 1, 1)
    assert (empty == -1).all()  # an agent with no mapped task anywhere answers [-1, -1] everywhere


@pytest.mark.parametrize('kwargs', [{}, dict(show_bad_actions=True)])
def test_baselines_drive_a_rollout_through_the_action_task_wrapper(oracle, kwargs):
    """The reference's usage: env wrapped with action_mapping_wrapper_v0, agents observe() then act(); every decision is checked
    against the oracle on the very tensors the agent saw, and the env accepts all actions (no error flags)."""
    from free_range_zoo_amd.envs import wildfire_v0
    from free_range_zoo_amd.envs.wildfire.baselines import NoopBaseline, RandomBaseline, StrongestBaseline, WeakestBaseline
    from free_range_zoo_amd.envs.wildfire.baselines._extreme import _jagged_parts
    from free_range_zoo_amd.wrappers import action_mapping_wrapper_v0
    B = 3000
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_rich(), parallel_envs=B, max_steps=20, device=torch.device('cuda'), **kwargs)
    env = action_mapping_wrapper_v0(env)
    observations, _ = env.reset(seed=torch.arange(B, dtype=torch.int32))
    makers = [StrongestBaseline, WeakestBaseline, RandomBaseline, NoopBaseline]
    agents = {name: makers[i % 4](name, B) for i, name in enumerate(env.agents)}
    for t in range(12):
        actions = {}
        for name, agent in agents.items():
            agent.observe(observations[name])
            actions[name] = agent.act(env.action_space(name))
            if isinstance(agent, (StrongestBaseline, WeakestBaseline)):
                obs, mapping = observations[name]
                tv, to, _ = (x.cpu().numpy() for x in _jagged_parts(obs['tasks']))
                _, mo, ml = (x.cpu().numpy() for x in _jagged_parts(mapping['agent_action_mapping']))
                want = oracle.wildfire_extreme_policy(tv, to, mo, ml, obs['self'].cpu().numpy(), agent.weakest, seed=0, step=t)
                assert np.array_equal(actions[name].cpu().numpy(), want), f'{name} step {t}'
        observations, rewards, terminations, truncations, infos = env.step(actions)
    env.check()
