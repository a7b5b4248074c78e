"""GPU: the HIP cybersecurity step path against golden vectors of the reference and the CPU oracle."""
import numpy as np
import pytest
import torch

import configs
import golden_util as G
from free_range_zoo_amd import _capi
from test_oracle_cybersecurity import compare_cyber, oracle_snapshot

pytestmark = pytest.mark.gpu


def make_env(build, B, max_steps, **kwargs):
    from free_range_zoo_amd.envs import cybersecurity_v0
    return cybersecurity_v0.parallel_env(configuration=build(), parallel_envs=B, max_steps=max_steps, device=torch.device('cuda'), **kwargs)


def np_(t):
    return t.detach().cpu().numpy()


def hip_snapshot(env):
    st = env.state()
    snap = {'network_state': np_(st.network_state), 'location': np_(st.location), 'presence': np_(st.presence),
            'num_moves': np_(env.num_moves), 'env_task_count': np_(env.environment_task_count),
            'agent_task_count': np_(env.agent_task_count)}
    snap['rewards'] = np.stack([np_(env.rewards[a]) for a in env.agents])
    snap['terminations'] = np.stack([np_(env.terminations[a]) for a in env.agents])
    snap['truncations'] = np.stack([np_(env.truncations[a]) for a in env.agents])
    for a, agent in enumerate(env.agents):
        m = env.agent_action_mapping[agent]
        snap[f'act_map_values_{a}'], snap[f'act_map_offsets_{a}'] = np_(m.values()), np_(m.offsets())
        m = env.agent_observation_mapping[agent]
        snap[f'obs_map_values_{a}'], snap[f'obs_map_offsets_{a}'] = np_(m.values()), np_(m.offsets())
        obs = env.observe(agent)
        snap[f'obs_self_{a}'], snap[f'obs_others_{a}'], snap[f'obs_tasks_{a}'] = np_(obs['self']), np_(obs['others']), np_(obs['tasks'])
        snap[f'cumulative_rewards_{a}'] = np_(env._cumulative_rewards[agent])
    return snap


def compare_snapshots(got, want, what):
    for key, w in want.items():
        g = got[key]
        if key in ('terminations', 'truncations', 'presence'):
            g, w = np.asarray(g).astype(bool), np.asarray(w).astype(bool)
        G.assert_same(g, w, f'{what} {key}')


@pytest.mark.parametrize('name', sorted(configs.CYBER_GOLDEN))
def test_golden_trajectory(name):
    build, kwargs = configs.CYBER_GOLDEN[name]
    data = np.load(G.golden_path(f'traj_cybersecurity_{name}.npz'))
    cfg = G.load_cfg(data, _capi.frz_cybersecurity_cfg)
    B, N, A = cfg.parallel_envs, cfg.num_nodes, cfg.num_attackers + cfg.num_defenders
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    compare_cyber(hip_snapshot(env), data, 'r_', A, f'{name} reset')
    for t in range(int(data['steps'])):
        p = f's{t}_'
        if bool(data[p + 'stepped']):
            rnd = (torch.from_numpy(data[p + 'network_randomness']), torch.from_numpy(data[p + 'agent_randomness']))
        else:
            rnd = (torch.zeros(1, B, N), torch.zeros(1, B, A))
        actions = {agent: torch.from_numpy(data[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)}
        obs, rewards, terminations, truncations, infos = env.step(actions, randomness=rnd)
        compare_cyber(hip_snapshot(env), data, p, A, f'{name} step {t}')
        G.assert_same(np_(env.finished), data[p + 'finished'], f'{name} step {t} finished')
    env.check()


def run_against_oracle(oracle, build, kwargs, B, max_steps, steps, seed, rng='injected'):
    from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct
    flags = dict(configs.CYBER_DEFAULT_FLAGS)
    flags.update(kwargs)
    cfg = to_cstruct(build(), B, max_steps, **flags)
    o = oracle.CybersecurityOracle(cfg)
    o.reset()
    env = make_env(build, B, max_steps, rng='philox' if rng == 'philox' else 'mt19937', **kwargs)
    seeds = torch.arange(B, dtype=torch.int32) * 5 + seed
    env.reset(seed=seeds)
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), f'reset B={B}')
    gen = np.random.default_rng(seed)
    N, A = cfg.num_nodes, cfg.num_attackers + cfg.num_defenders
    mt_state, mt_index = oracle.mt19937_seed(seeds.numpy())
    for t in range(steps):
        actions_dev = env.random_policy_actions(policy_seed=99 + seed, policy_step=t).clone()
        actions = oracle.cybersecurity_random_policy(cfg, o.agent_task_count, o.location, seeds.numpy(), 99 + seed, t)
        G.assert_same(np_(actions_dev), actions, f'policy step {t}')
        if rng == 'injected':
            nr, ar = gen.random((1, B, N), dtype=np.float32), gen.random((1, B, A), dtype=np.float32)
            env.step(actions_dev, randomness=(torch.from_numpy(nr), torch.from_numpy(ar)))
        elif rng == 'philox':
            nr, ar = oracle.cybersecurity_philox_randomness(cfg, seeds.numpy(), o.num_moves)
            env.step(actions_dev)
        else:
            nr = oracle.mt19937_generate(mt_state, mt_index, 1, N)
            ar = oracle.mt19937_generate(mt_state, mt_index, 1, A)
            env.step(actions_dev)
        o.step(actions, nr, ar)
        compare_snapshots(hip_snapshot(env), oracle_snapshot(o), f'B={B} rng={rng} step {t}')
    env.check()
    assert int(o.error_flags[0]) == 0


@pytest.mark.parametrize('B', [1, 255, 257, 4000])
def test_vs_oracle_ragged_batches(oracle, B):
    run_against_oracle(oracle, configs.cyber_openness, {}, B, 30, 33, seed=B)


@pytest.mark.parametrize('name', ['rich', 'rich_fully_observable', 'openness_no_bad_actions'])
def test_vs_oracle_variants(oracle, name):
    build, kwargs = configs.CYBER_GOLDEN[name]
    run_against_oracle(oracle, build, kwargs, 2500, 25, 27, seed=4)


def test_vs_oracle_multi_round_and_rng_modes(oracle):
    run_against_oracle(oracle, configs.cyber_openness, {}, 300000, 10, 5, seed=6)
    run_against_oracle(oracle, configs.cyber_openness, {}, 3001, 20, 12, seed=7, rng='philox')
    run_against_oracle(oracle, configs.cyber_rich, {}, 700, 20, 12, seed=8, rng='philox')
    run_against_oracle(oracle, configs.cyber_openness, {}, 1200, 30, 10, seed=9, rng='mt19937')


@pytest.mark.parametrize('shape', [(12, 5, 7), (16, 3, 6), (7, 8, 8)])
def test_large_variants_match_the_oracle(oracle, shape):
    """<16,16> kernel variant: more than 8 nodes or agents; 2^(Att+D) danger entries beyond the LDS-staged 1024 read from HBM."""
    build = lambda: configs.cyber_grid(*shape)
    run_against_oracle(oracle, build, {}, 600, 20, 12, seed=41)
    run_against_oracle(oracle, build, dict(show_bad_actions=False, observe_other_presence=True, observe_other_location=True), 300, 20, 10,
                       seed=42, rng='philox')


def test_full_size_properties():
    """BASELINE.json config 4: B = 65 536, 2 attackers + 2 defenders, agent openness on."""
    B = 65536
    envs = [make_env(configs.cyber_openness, B, 50, rng='philox', exact_shapes=False) for _ in range(2)]
    for env in envs:
        env.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(50):
        for env in envs:
            env.step(env.random_policy_actions(policy_seed=3, policy_step=t))
        if t % 9 == 0 or t == 49:
            a, b = envs
            for name in ('_network_state', '_location', '_presence', '_rewards', '_act_map_offsets', '_tasks'):
                assert torch.equal(getattr(a, name), getattr(b, name)), name
            assert bool((a._network_state >= 0).all()) and bool((a._network_state <= 4).all())
            assert bool((a._location >= -1).all()) and bool((a._location < 3).all())
            for ag in range(4):
                off, present = a._act_map_offsets[ag], a._presence[ag]
                assert torch.equal(off[1:] - off[:-1], present.long() * 3)
                assert torch.equal(a.agent_task_count[ag], present.int() * 3)
                total = int(off[-1])
                assert torch.equal(a._act_map_values[ag, :total].view(-1, 3), torch.arange(3, device='cuda', dtype=torch.int32).expand(total // 3, 3))
            # zero-sum network reward between attackers and defenders (patch_reward = 0): cybersecurity.py:401-409
            assert torch.equal(a._rewards[0], -a._rewards[2]) and torch.equal(a._rewards[0], a._rewards[1])
    assert bool(envs[0].truncated.all()) and not bool(envs[0].terminated.any())
    envs[0].check()


def test_invalid_target_is_flagged():
    env = make_env(configs.cyber_non_stochastic, 32, 10)
    env.reset(seed=torch.arange(32, dtype=torch.int32))
    actions = {agent: torch.tensor([[0, -1]], dtype=torch.int32).repeat(32, 1).cuda() for agent in env.agents}
    actions['attacker_1'][3] = torch.tensor([7, 0], dtype=torch.int32)  # node 7 does not exist (cybersecurity.py:341-346 raises)
    env.step(actions)
    with pytest.raises(ValueError):
        env.check()


def test_action_space_members():
    env = make_env(configs.cyber_openness, 512, 20, show_bad_actions=False)
    env.reset(seed=torch.arange(512, dtype=torch.int32))
    for t in range(6):
        actions = {}
        for a, agent in enumerate(env.agents):
            sample = env.action_space(agent).sample_nested()
            counts = env.agent_task_count[a].long()
            move = sample[:, 1] == 0
            assert bool((sample[move, 0] < counts[move]).all())
            assert bool((sample[counts == 0, 1] == -1).all())  # absent agents can only noop
            if agent.startswith('defender'):
                home = env._location[a - 2] == -1
                assert not bool(((sample[:, 1] == -2) & home).any())  # no patch at the home node (spaces/actions.py:62-68)
            actions[agent] = sample
        env.step(actions)
    env.check()


@pytest.mark.parametrize('rng', ['philox', 'mt19937'])
@pytest.mark.parametrize('build,kwargs', [(configs.cyber_openness, {}), (configs.cyber_rich, dict(show_bad_actions=False)),
                                          (lambda: configs.cyber_grid(12, 5, 7), dict(partially_observable=False))])
def test_fused_random_policy_step_equals_policy_then_step(rng, build, kwargs):
    """frz_cybersecurity_step_random_policy (policy sampled inside the step launch) == random_policy_actions() + step(): same
    actions, state, rewards and observations, every step of an episode that runs into the all-finished early-out."""
    B = 1500
    two, one = (make_env(build, B, 9, rng=rng, **kwargs) for _ in range(2))
    for env in (two, one):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 3)
    for t in range(12):
        live = not bool(two.finished.all())
        acts = two.random_policy_actions(policy_seed=77, policy_step=t).clone()
        two.step(acts)
        one.step_random_policy(policy_seed=77, policy_step=t)
        if live:  # a frozen step does not write the sampled actions
            assert torch.equal(one._actions, acts), f'step {t}: actions'
        a, b = hip_snapshot(two), hip_snapshot(one)
        for key in a:
            G.assert_same(b[key], a[key], f'step {t} {key}')
        for agent in two.agents:
            assert torch.equal(two.rewards[agent], one.rewards[agent]) and torch.equal(two.truncations[agent], one.truncations[agent])
            assert torch.equal(two.observations[agent]['tasks'], one.observations[agent]['tasks'])
            assert torch.equal(two.observations[agent]['self'], one.observations[agent]['self'])
    two.check(), one.check()


def test_graph_replayed_rollout_equals_eager_rollout():
    B = 2000
    eager, replay = (make_env(configs.cyber_openness, B, 15, rng='philox') for _ in range(2))
    replay.set_exclusive_device(True)  # the graph then holds the rollout as one multi-step launch
    seed = torch.arange(B, dtype=torch.int32) + 11
    eager.reset(seed=seed), replay.reset(seed=seed)
    graph = replay.capture_random_rollout(15, policy_seed=5, include_reset=True)
    graph.replay()
    eager.reset(seed=seed)
    for t in range(15):
        eager.step_random_policy(policy_seed=5, policy_step=t)
    torch.cuda.synchronize()
    replay._publish()  # the replay wrote the persistent buffers behind Python's back: refresh the exact-shape views
    a, b = hip_snapshot(eager), hip_snapshot(replay)
    for key in a:
        G.assert_same(b[key], a[key], key)
    assert torch.equal(eager._cumulative, replay._cumulative) and torch.equal(eager._actions, replay._actions)


@pytest.mark.parametrize('build,steps', [(configs.cyber_openness, 100), (configs.cyber_rich, 60), (lambda: configs.cyber_grid(12, 5, 7), 12)])
def test_mt19937_streams_advanced_in_the_step_match_the_generator(oracle, build, steps):
    """Default rng: the <4,4> and <8,8> step kernels advance the per-env MT19937 streams themselves (the 16-node one stages the
    draws with a generator launch); same trajectory as the oracle fed from the oracle's MT19937, across the 624-word wrap."""
    run_against_oracle(oracle, build, {}, 300, 200, steps, seed=51, rng='mt19937')


def test_mt19937_stream_is_left_alone_by_frozen_steps(oracle):
    """Once every env is truncated the reference returns before drawing (utils/env.py:211-213): the in-kernel streams agree with a
    generator that stopped drawing at the last real step."""
    B = 500
    env = make_env(configs.cyber_openness, B, 5, rng='mt19937')
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(9):
        env.step(env.random_policy_actions(policy_seed=1, policy_step=t))
    N, A = env._N, len(env.agents)
    mt_state, mt_index = oracle.mt19937_seed(np.arange(B, dtype=np.int32))
    for t in range(5):
        oracle.mt19937_generate(mt_state, mt_index, 1, N)
        oracle.mt19937_generate(mt_state, mt_index, 1, A)
    assert np.array_equal(np_(env._view(env._bufs.mt_index, (B, ), torch.int32)), mt_index)


@pytest.mark.parametrize('shape', [(4, 2, 2), (3, 1, 3), (2, 1, 1), (3, 3, 1), (6, 2, 2)])
def test_runtime_shape_small_variants_match_the_oracle(oracle, shape):
    """The 3-node / 2-attacker / 2-defender shape has an exact instantiation; every other shape of <= 4 (<= 8) nodes and agents runs
    the runtime-shape <4,4> (<8,8>) kernels: injected, Philox, in-kernel MT19937 and the fused policy on those."""
    build = lambda: configs.cyber_grid(*shape)
    run_against_oracle(oracle, build, {}, 500, 20, 12, seed=71)
    run_against_oracle(oracle, build, dict(show_bad_actions=False, observe_other_location=True), 400, 20, 10, seed=72, rng='philox')
    run_against_oracle(oracle, build, dict(partially_observable=False), 300, 150, 40, seed=73, rng='mt19937')
    B = 700
    two, one = (make_env(build, B, 9, rng='philox') for _ in range(2))
    for env in (two, one):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 9)
    for t in range(11):
        two.step(two.random_policy_actions(policy_seed=5, policy_step=t).clone())
        one.step_random_policy(policy_seed=5, policy_step=t)
        a, b = hip_snapshot(two), hip_snapshot(one)
        for key in a:
            G.assert_same(b[key], a[key], f'{shape} step {t} {key}')


# ------------------------------------------------------------------------------------------------------------------
# multi-step launches: frz_cybersecurity_rollout_random_policy as ONE launch (cy_roles_kernel<..., PERSIST>)
# ------------------------------------------------------------------------------------------------------------------
def assert_same_rollout(one, many, what):
    a, b = hip_snapshot(one), hip_snapshot(many)
    for key in a:
        G.assert_same(b[key], a[key], f'{what}: {key}')
    for name in ('_cumulative', '_actions', '_frozen_scaled', '_act_map_offsets', 'num_moves'):
        assert torch.equal(getattr(one, name), getattr(many, name)), f'{what}: {name}'


@pytest.mark.parametrize('build,B,max_steps,steps,kwargs,launches', [
    (configs.cyber_openness, 65536, 50, 50, {}, 1),                                   # the bench workload (exact <3,4,2> kernel)
    (configs.cyber_openness, 1000, 50, 7, {}, 1),                                     # ragged last chunk
    (configs.cyber_rich, 3000, 40, 9, dict(show_bad_actions=False), 1),               # runtime-shape kernel, every observation flag
    (lambda: configs.cyber_grid(7, 4, 4), 1300, 30, 8, {}, 1),                        # <8,8>
    (lambda: configs.cyber_grid(12, 5, 7), 600, 30, 5, {}, 5),                        # 16-node kernel: no multi-step launch, one per step
    (configs.cyber_openness, 1500, 5, 8, {}, 1),                                      # every env truncated after 5 of the 8 steps
    (configs.cyber_openness, 1500, 5, 9, {}, 1),
], ids=['bench', 'ragged', 'rich', '7n8a', 'fallback_12n', 'ends_at_5_of_8', 'ends_at_5_of_9'])
def test_multi_step_launch_equals_single_step_launches(build, B, max_steps, steps, kwargs, launches):
    """rollout_random_policy(n) — one launch whose state role keeps its envs in registers across the n steps — leaves exactly what n
    step_random_policy launches leave (state, rewards, observations, sampled actions, action mappings), also when the episode ends
    inside the launch; off unless the caller declares the device its own."""
    one, many = [make_env(build, B, max_steps, rng='philox', **kwargs) for _ in range(2)]
    many.set_exclusive_device(True)
    assert many._lib.frz_cybersecurity_rollout_launches(many._handle, steps, _capi.FRZ_RNG_PHILOX) == launches
    assert one._lib.frz_cybersecurity_rollout_launches(one._handle, steps, _capi.FRZ_RNG_PHILOX) == steps
    for env in (one, many):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 5)
    for t in range(steps):
        one.step_random_policy(policy_seed=3, policy_step=t)
    many.rollout_random_policy(steps, policy_seed=3, first_step=0)
    assert_same_rollout(one, many, 'first rollout')
    for t in range(steps, steps + 3):
        one.step_random_policy(policy_seed=3, policy_step=t)
    many.rollout_random_policy(3, policy_seed=3, first_step=steps)
    assert_same_rollout(one, many, 'second rollout')
    for env in (one, many):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 6)
    one.step_random_policy(policy_seed=3, policy_step=0), one.step_random_policy(policy_seed=3, policy_step=1)
    many.rollout_random_policy(2, policy_seed=3, first_step=0)
    assert_same_rollout(one, many, 'after a reset')
    one.check()
    many.check()
