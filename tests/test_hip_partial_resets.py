"""GPU: reset_batches(batch_indices, seed) in the middle of an episode, reset(options={'initial_state': ...}) and
reset(options={'skip_seeding': True}) against runs recorded from the reference (tests/golden/partial_*.npz; generator and caveats:
tools/refharness/make_golden.py:record_partial_resets), and the buffered / single_seeding RandomGenerator against draws recorded from the
reference's own class (tests/golden/rng_modes.npz)."""
import json

import numpy as np
import pytest
import torch

import configs
import golden_util as G
from free_range_zoo_amd import _capi

pytestmark = pytest.mark.gpu


def _replay(env, data, prefix, steps, random_names, sizes, compare, snapshot, A, what):
    B = env.parallel_envs
    for t in range(steps):
        p = f'{prefix}{t}_'
        if bool(data[p + 'stepped']):
            rnd = tuple(torch.from_numpy(data[p + name]) for name in random_names)
        else:
            rnd = tuple(torch.zeros(*size) for size in sizes)
        actions = {agent: torch.from_numpy(data[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)}
        env.step(actions, randomness=rnd)
        compare(snapshot(env), data, p, A, f'{what} {p}')
        G.assert_same(env.finished.cpu().numpy(), data[p + 'finished'], f'{what} {p} finished')


def _run(label, module_name, table, struct, random_names, sizes_of):
    import importlib
    test_module = importlib.import_module(f'test_hip_{module_name}')
    data = np.load(G.golden_path(f'partial_{label}.npz'))
    assert 'initial' in str(data['reset_batches_reference_error'])  # the reference's own call stops at utils/state.py:47
    build, kwargs = table[str(data['variant'])]
    cfg = G.load_cfg(data, struct)
    B = cfg.parallel_envs
    A = cfg.num_agents if hasattr(cfg, 'num_agents') else cfg.num_attackers + cfg.num_defenders
    compare = G.compare_wildfire if module_name == 'wildfire' else test_module.compare_cyber
    snapshot, make_env = test_module.hip_snapshot, test_module.make_env
    sizes = sizes_of(cfg, B, A)
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    compare(snapshot(env), data, 'r_', A, f'{label} reset')
    _replay(env, data, 'a', 6, random_names, sizes, compare, snapshot, A, label)
    env.reset_batches(torch.from_numpy(data['batch_indices']), seed=torch.from_numpy(data['batch_seeds']))
    compare(snapshot(env), data, 'p_', A, f'{label} after reset_batches')
    for key, mine in (('p_rewards', env.rewards), ('p_terminations', env.terminations), ('p_truncations', env.truncations)):
        got = np.stack([mine[agent].cpu().numpy() for agent in env.agents])
        G.assert_same(got.astype(data[key].dtype), data[key], f'{label} {key}')
    assert torch.equal(env.seeds[torch.from_numpy(data['batch_indices']).cuda()].cpu(), torch.from_numpy(data['batch_seeds']))
    _replay(env, data, 'b', 5, random_names, sizes, compare, snapshot, A, label)
    saved = env.state().clone()
    env2 = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env2.reset(seed=torch.arange(B, dtype=torch.int32), options={'initial_state': saved})
    compare(snapshot(env2), data, 'i_', A, f'{label} restarted from a saved state')
    _replay(env2, data, 'c', 4, random_names, sizes, compare, snapshot, A, label)
    env2.reset(options={'skip_seeding': True})
    compare(snapshot(env2), data, 'k_', A, f'{label} reset without reseeding')
    env.check(), env2.check()


@pytest.mark.parametrize('label', ['wildfire', 'wildfire_bad_actions', 'wildfire_grid8x8'])  # (the last one: the cells-across-lanes family)
def test_wildfire_partial_resets_match_the_reference(label):
    _run(label, 'wildfire', configs.WILDFIRE_GOLDEN, _capi.frz_wildfire_cfg, ('field_randomness', 'agent_randomness'),
         lambda cfg, B, A: ((3, B, cfg.grid_height * cfg.grid_width), (5, B, A)))


def test_cybersecurity_partial_resets_match_the_reference():
    _run('cybersecurity', 'cybersecurity', configs.CYBER_GOLDEN, _capi.frz_cybersecurity_cfg, ('network_randomness', 'agent_randomness'),
         lambda cfg, B, A: ((1, B, cfg.num_nodes), (1, B, A)))


def test_random_generator_modes_match_the_reference_class():
    """utils/random_generator.py:49-146: keyed buffers (one per key and shape, refilled every buffer_size calls) and single_seeding (one
    default-seeded torch stream for every env) reproduce, value for value, what the reference's RandomGenerator returned."""
    from free_range_zoo_amd.utils.random_generator import RandomGenerator
    data = np.load(G.golden_path('rng_modes.npz'))
    calls = json.loads(str(data['calls']))
    for name, case in json.loads(str(data['cases'])).items():
        generator = RandomGenerator(parallel_envs=case['parallel_envs'], buffer_size=case['buffer_size'], single_seeding=case['single_seeding'],
                                    device=torch.device('cuda'))
        B = case['parallel_envs']
        if not case['single_seeding']:  # per-env device streams live in buffers the env normally provides
            generator.attach(seeds=torch.zeros(B, dtype=torch.int32, device='cuda'), states=torch.zeros((624, B), dtype=torch.int32, device='cuda'),
                             index=torch.zeros(B, dtype=torch.int32, device='cuda'))
        generator.seed(torch.tensor(case['seed'], dtype=torch.int32))
        for i, (key, events, shape) in enumerate(calls):
            got = generator.generate(B, events, tuple(shape), key=key)
            want = data[f'{name}_{i}']
            assert tuple(got.shape) == want.shape, (name, i)
            G.assert_same(got.cpu().numpy(), want, f'{name} call {i} ({key})')
