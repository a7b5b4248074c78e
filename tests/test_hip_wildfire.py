"""GPU: the HIP wildfire step path (through the Python boundary and the C-ABI) against golden vectors and the oracle."""
import numpy as np
import pytest
import torch

import configs
import golden_util as G
from free_range_zoo_amd import _capi

pytestmark = pytest.mark.gpu


def make_env(build, B, max_steps, **kwargs):
    from free_range_zoo_amd.envs import wildfire_v0
    return wildfire_v0.parallel_env(configuration=build(), parallel_envs=B, max_steps=max_steps, device=torch.device('cuda'), **kwargs)


def np_(t):
    return t.detach().cpu().numpy()


def hip_snapshot(env):
    """Same naming/layout as the golden snapshots (batch-major state) from the env's public attributes."""
    A, B = len(env.agents), env.parallel_envs
    st = env.state()
    snap = {name: np_(getattr(st, name)).reshape(B, -1) if name in ('fires', 'intensity', 'fuel') else np_(getattr(st, name))
            for name in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment')}
    snap['num_moves'], snap['num_burnouts'] = np_(env.num_moves), np_(env.num_burnouts)
    snap['env_task_count'], snap['agent_task_count'] = np_(env.environment_task_count), np_(env.agent_task_count)
    snap['task_values'], snap['task_offsets'] = np_(env.task_store.values()), np_(env.task_store.offsets())
    snap['rewards'] = np.stack([np_(env.rewards[a]) for a in env.agents])
    snap['terminations'] = np.stack([np_(env.terminations[a]) for a in env.agents])
    snap['truncations'] = np.stack([np_(env.truncations[a]) for a in env.agents])
    snap['burnouts'], snap['putouts'] = np_(env._burnouts), np_(env._putouts)
    for a, agent in enumerate(env.agents):
        m = env.agent_action_mapping[agent]
        snap[f'act_map_values_{a}'], snap[f'act_map_offsets_{a}'] = np_(m.values()), np_(m.offsets())
        m = env.agent_observation_mapping[agent]
        snap[f'obs_map_values_{a}'], snap[f'obs_map_offsets_{a}'] = np_(m.values()), np_(m.offsets())
        if env.show_bad_actions:
            m = env.agent_bad_actions[agent]
            snap[f'bad_map_values_{a}'], snap[f'bad_map_offsets_{a}'] = np_(m.values()), np_(m.offsets())
        obs = env.observe(agent)
        snap[f'obs_self_{a}'], snap[f'obs_others_{a}'] = np_(obs['self']), np_(obs['others'])
        snap[f'cumulative_rewards_{a}'] = np_(env._cumulative_rewards[agent])
    return snap


def oracle_snapshot(o):
    from test_oracle_wildfire import oracle_snapshot as snap
    return snap(o)


def compare_snapshots(got, want, what, keys=None):
    for key in (keys or want.keys()):
        rtol = 0.0  # HIP vs oracle: bit-exact, rewards included (same float32 operation order on both sides)
        w = want[key]
        g = got[key]
        if key in ('terminations', 'truncations'):
            g, w = np.asarray(g).astype(bool), np.asarray(w).astype(bool)
        G.assert_same(g, w, f'{what} {key}', rtol)


# ------------------------------------------------------------------------------------------------------------
# 1. golden trajectories of the unmodified reference
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', sorted(configs.WILDFIRE_GOLDEN))
def test_golden_trajectory(name):
    build, kwargs = configs.WILDFIRE_GOLDEN[name]
    data = np.load(G.golden_path(f'traj_wildfire_{name}.npz'))
    cfg = G.load_cfg(data, _capi.frz_wildfire_cfg)
    B, A = cfg.parallel_envs, cfg.num_agents
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    obs, infos = env.reset(seed=torch.arange(B, dtype=torch.int32))
    assert set(obs) == set(env.agents) and set(infos) == set(env.agents)
    G.compare_wildfire(hip_snapshot(env), data, 'r_', A, f'{name} reset')
    HW = cfg.grid_height * cfg.grid_width
    for t in range(int(data['steps'])):
        p = f's{t}_'
        if bool(data[p + 'stepped']):
            rnd = (torch.from_numpy(data[p + 'field_randomness']), torch.from_numpy(data[p + 'agent_randomness']))
        else:  # reference early-out: nothing drawn; the kernel must not touch anything whatever it is given
            rnd = (torch.zeros(3, B, HW), torch.zeros(5, B, A))
        actions = {agent: torch.from_numpy(data[p + 'actions'][a]).cuda() for a, agent in enumerate(env.agents)}
        obs, rewards, terminations, truncations, infos = env.step(actions, randomness=rnd)
        G.compare_wildfire(hip_snapshot(env), data, p, A, f'{name} step {t}')
        G.assert_same(np_(env.finished), data[p + 'finished'], f'{name} step {t} finished')
        assert infos['burnouts'].dtype == torch.int64 and rewards[env.agents[0]].dtype == torch.float32
        assert terminations[env.agents[0]].dtype == torch.bool
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 2. HIP vs oracle on seeded inputs: multi-chunk batches, ragged tails, every RNG mode
# ------------------------------------------------------------------------------------------------------------
def run_against_oracle(oracle, build, kwargs, B, max_steps, steps, seed, rng='injected', policy='oracle'):
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
    flags.update(kwargs)
    cfg = to_cstruct(build(), B, max_steps, **flags)
    o = oracle.WildfireOracle(cfg)
    o.reset()
    env = make_env(build, B, max_steps, rng='philox' if rng == 'philox' else 'mt19937', **kwargs)
    seeds = torch.arange(B, dtype=torch.int32) * 7 + seed
    env.reset(seed=seeds)
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), f'reset B={B}')
    gen = np.random.default_rng(seed)
    HW, A = cfg.grid_height * cfg.grid_width, cfg.num_agents
    mt_state, mt_index = oracle.mt19937_seed(seeds.numpy())
    for t in range(steps):
        counts_a, counts_e = o.agent_task_count, o.env_task_count
        if policy == 'device':
            actions_dev = env.random_policy_actions(policy_seed=1234 + seed, policy_step=t).clone()
            actions = oracle.wildfire_random_policy(cfg, counts_a, counts_e, seeds.numpy(), 1234 + seed, t)
            G.assert_same(np_(actions_dev), actions, f'policy step {t}')
        else:
            n = (counts_e[None, :] if cfg.show_bad_actions else counts_a).astype(np.int64)
            j = np.minimum((gen.random((A, B)) * (n + 1)).astype(np.int64), n)
            actions = np.stack([j, np.where(j < n, 0, -1)], axis=-1).astype(np.int32)
        if rng == 'injected':
            fr, ar = gen.random((3, B, HW), dtype=np.float32), gen.random((5, B, A), dtype=np.float32)
            env.step(torch.from_numpy(actions).cuda(), randomness=(torch.from_numpy(fr), torch.from_numpy(ar)))
        elif rng == 'philox':
            fr, ar = oracle.wildfire_philox_randomness(cfg, seeds.numpy(), o.num_moves)
            env.step(torch.from_numpy(actions).cuda())
        else:  # mt19937: the env's generator must reproduce the per-env reference streams
            fr = oracle.mt19937_generate(mt_state, mt_index, 3, HW)
            ar = oracle.mt19937_generate(mt_state, mt_index, 5, A)
            if bool(o.terminations[0].all() or o.truncations[0].all()):
                pass  # frozen: the reference draws nothing (step returns before step_environment)
            env.step(torch.from_numpy(actions).cuda())
        o.step(actions, fr, ar)
        compare_snapshots(hip_snapshot(env), oracle_snapshot(o), f'B={B} rng={rng} step {t}')
    env.check()
    assert int(o.error_flags[0]) == 0
    return env, o


@pytest.mark.parametrize('B', [1, 255, 256, 257, 1000, 5000])
def test_vs_oracle_ragged_batches(oracle, B):
    run_against_oracle(oracle, configs.wildfire_openness, {}, B, 30, 34, seed=B)


@pytest.mark.parametrize('name', ['rich_localized', 'rich_plain_bad_actions', 'openness_bad_actions', 'openness_observe_all'])
def test_vs_oracle_variants(oracle, name):
    build, kwargs = configs.WILDFIRE_GOLDEN[name]
    run_against_oracle(oracle, build, kwargs, 3000, 25, 28, seed=3)


def test_vs_oracle_multi_round_persistent_grid(oracle):
    """More chunks than resident workgroups: the persistent loop and the round-to-round prefix hand-off."""
    run_against_oracle(oracle, configs.wildfire_openness, {}, 300000, 12, 6, seed=5)


def test_philox_mode_matches_oracle_stream(oracle):
    run_against_oracle(oracle, configs.wildfire_openness, {}, 4097, 20, 22, seed=9, rng='philox')
    run_against_oracle(oracle, configs.wildfire_rich, dict(observe_other_suppressant=True), 777, 20, 12, seed=10, rng='philox')


def test_mt19937_mode_reproduces_reference_seeded_streams(oracle):
    """Default rng: same seeds -> same randomness as the reference's per-env CPU generators -> same trajectory."""
    run_against_oracle(oracle, configs.wildfire_openness, {}, 1500, 50, 20, seed=11, rng='mt19937')


def test_device_random_policy_matches_oracle(oracle):
    run_against_oracle(oracle, configs.wildfire_openness, {}, 2049, 20, 10, seed=12, policy='device')
    run_against_oracle(oracle, configs.wildfire_openness, dict(show_bad_actions=True), 513, 20, 10, seed=13, policy='device')


@pytest.mark.parametrize('rng', ['philox', 'mt19937'])
@pytest.mark.parametrize('kwargs', [{}, dict(show_bad_actions=True)])
def test_fused_random_policy_step_equals_policy_then_step(rng, kwargs):
    """frz_wildfire_step_random_policy (one launch) leaves exactly what random_policy + step (two launches) leave."""
    B, steps = 3001, 25  # partial last chunk
    two, one = [make_env(configs.wildfire_openness, B, 20, rng=rng, exact_shapes=False, **kwargs) for _ in range(2)]
    for env in (two, one):
        env.reset(seed=torch.arange(B, dtype=torch.int32) * 3 + 1)
    names = ('_fires', '_intensity', '_fuel', '_suppressants', '_capacity', '_equipment', '_rewards', '_task_offsets', '_task_values',
             '_act_map_offsets', '_act_map_values', '_obs_map_values', '_burnouts', '_putouts')
    for t in range(steps):
        if rng == 'mt19937':  # device-side MT19937 streams on both paths
            acts = two.random_policy_actions(policy_seed=5, policy_step=t)
            from free_range_zoo_amd import _capi
            from free_range_zoo_amd.utils.env import stream_ptr
            two.generator._ensure_streams()
            _capi.check(two._lib.frz_wildfire_step(two._handle, acts.data_ptr(), _capi.FRZ_RNG_MT19937, None, None, stream_ptr(two.device)), 'step')
        else:
            two.step(two.random_policy_actions(policy_seed=5, policy_step=t))
        live = not bool(one.finished.all())  # a frozen batch ignores its actions: the fused launch leaves the buffer untouched
        one.step_random_policy(policy_seed=5, policy_step=t)
        if live:
            assert torch.equal(two._actions, one._actions), f'actions at step {t}'
        for name in names:
            assert torch.equal(getattr(two, name), getattr(one, name)), f'{name} at step {t}'
        assert torch.equal(two.finished, one.finished)
    one.check()


@pytest.mark.parametrize('kernel', ['roles', 'lane'])
def test_mt19937_streams_advanced_in_the_step_equal_generated_tensors(kernel, monkeypatch):
    """FRZ_RNG_MT19937 through frz_wildfire_step (field/crew kernel: the stream is advanced inside the step; lane kernel:
    two frz_mt19937_generate launches) draws exactly what RandomGenerator.generate() hands to an injected step — across
    the 624-word wrap of the streams (33 draws per step: wraps at steps 19 and 38) and after a partial re-seed."""
    monkeypatch.setenv('FRZ_WF_KERNEL', kernel)
    B = 777
    a, b = [make_env(configs.wildfire_openness, B, 50, rng='mt19937', exact_shapes=False) for _ in range(2)]
    seeds = torch.arange(B, dtype=torch.int32) * 11 + 3
    a.reset(seed=seeds)
    b.reset(seed=seeds)
    for t in range(45):
        if t == 7:  # streams at different positions inside one wavefront
            idx = torch.arange(0, B, 3, dtype=torch.int32)
            for env in (a, b):
                env.generator.seed(torch.arange(idx.numel(), dtype=torch.int32) + 1000, partial_seeding=idx)
        acts = a.random_policy_actions(policy_seed=9, policy_step=t).clone()
        field = b.generator.generate(B, 3, (2, 3), key='field')
        agent = b.generator.generate(B, 5, (3, ), key='agent')
        a.step(acts)                              # in-kernel / staged MT19937 through the C-ABI
        b.step(acts, randomness=(field, agent))   # the same streams drawn by the generator API
        for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_capacity', '_equipment', '_rewards', '_task_offsets'):
            assert torch.equal(getattr(a, name), getattr(b, name)), f'{name} at step {t}'
    assert torch.equal(a.generator.generator_index, b.generator.generator_index)
    assert torch.equal(a.generator.generator_states, b.generator.generator_states)


def test_single_seeding_shares_one_host_stream():
    """single_seeding=True (random_generator.py:59-65, 103-106): one torch CPU generator for all envs, started from a fresh
    generator's state; the env consumes its draws like injected randomness."""
    B = 33
    env = make_env(configs.wildfire_openness, B, 20, rng='mt19937', single_seeding=True)
    env.reset(seed=torch.full((B, ), 5, dtype=torch.int32))
    want = torch.Generator(device='cpu')
    field = torch.rand((B, 3, 2, 3), generator=want).transpose(1, 0)
    agent = torch.rand((B, 5, 3), generator=want).transpose(1, 0)
    twin = make_env(configs.wildfire_openness, B, 20, rng='mt19937')
    twin.reset(seed=torch.full((B, ), 5, dtype=torch.int32))
    acts = env.random_policy_actions(policy_seed=1, policy_step=0).clone()
    env.step(acts)
    twin.step(acts, randomness=(field, agent))
    for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards'):
        assert torch.equal(getattr(env, name), getattr(twin, name)), name
    got = env.generator.generate(B, 2, (4, ))
    assert torch.equal(got.cpu(), torch.rand((B, 2, 4), generator=want).transpose(1, 0))
    with pytest.raises(NotImplementedError):
        env.step_random_policy(policy_seed=1, policy_step=1)


@pytest.mark.parametrize('rng', ['philox', 'mt19937'])
@pytest.mark.parametrize('B,kwargs', [(4096, {}), (3001, dict(show_bad_actions=True))])
def test_rollout_equals_step_by_step(rng, B, kwargs):
    """frz_wildfire_rollout_random_policy(n) leaves exactly what n calls of step_random_policy leave — through truncation of every
    env (max_steps 20 < 26 steps)."""
    steps = 26
    one, many = [make_env(configs.wildfire_openness, B, 20, rng=rng, exact_shapes=False, **kwargs) for _ in range(2)]
    for env in (one, many):
        env.reset(seed=torch.arange(B, dtype=torch.int32) * 5 + 2)
    names = ('_fires', '_intensity', '_fuel', '_suppressants', '_capacity', '_equipment', '_rewards', '_cumulative', '_task_offsets',
             '_task_values', '_act_map_offsets', '_act_map_values', '_obs_map_values', '_burnouts', '_putouts', '_obs_self', '_obs_others')
    done = 0
    for n in (1, 7, 13, 5):  # several rollouts back to back, the last one runs into the frozen batch
        one.rollout_random_policy(n, policy_seed=4, first_step=done)
        for t in range(done, done + n):
            many.step_random_policy(policy_seed=4, policy_step=t)
        done += n
        for name in names:
            assert torch.equal(getattr(one, name), getattr(many, name)), f'{name} after {done} steps'
        assert torch.equal(one.num_moves, many.num_moves) and torch.equal(one.finished, many.finished)
    assert done == steps and bool(one.finished.all())
    one.check()


def test_episode_metrics_in_one_launch():
    """frz_wildfire_episode_metrics accumulates what the torch reductions of utils/sharding.py give."""
    from free_range_zoo_amd.utils import sharding
    B = 20011
    env = make_env(configs.wildfire_openness, B, 20, rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    metrics = torch.zeros(5, dtype=torch.float64, device='cuda')
    for episode in range(2):
        for t in range(13 + 9 * episode):  # the second episode runs into truncation
            env.step_random_policy(policy_seed=2, policy_step=t)
        before = metrics.clone()
        env.accumulate_episode_metrics(metrics)
        want = sharding.episode_metrics(env._cumulative, env.finished, int(env.num_moves.sum()))
        torch.testing.assert_close(metrics - before, want, rtol=1e-12, atol=1e-9)
    assert float(metrics[4] - before[4]) == B  # every env is truncated at max_steps = 20 by the end of the second block


def test_reference_evaluation_protocol_runs_unchanged():
    """The rollout loop of the reference's evaluation script (docs/source/events/moasei-2026/evaluation: 3 firefighters,
    max_steps=50, buffer_size=50, single_seeding=True, show_bad_actions=False, reset(seed=0), loop until finished)."""
    from free_range_zoo_amd.envs import wildfire_v0
    env = wildfire_v0.parallel_env(parallel_envs=64, max_steps=50, configuration=configs.wildfire_openness(), device=torch.device('cuda'),
                                   buffer_size=50, single_seeding=True, show_bad_actions=False)
    observations, infos = env.reset(seed=0)
    assert set(observations) == set(env.agents) and len(env.agents) == 3
    steps = 0
    while not torch.all(env.finished):
        actions = {agent: env.action_space(agent).sample_nested() for agent in env.agents}  # int32 [B, 2] on the device
        observations, rewards, terminations, truncations, infos = env.step(actions)
        steps += 1
        assert steps <= 50
    assert steps == 50 or bool(env.terminated.all())
    assert all(rewards[a].shape == (64, ) for a in env.agents)
    env.check()


def test_timed_rollout_runs_the_same_steps():
    """frz_wildfire_timed_rollout (measurement aid): same state as the untimed launches, one positive duration per step."""
    import ctypes
    from free_range_zoo_amd import _capi
    from free_range_zoo_amd.utils.env import stream_ptr
    B, n = 5000, 12
    plain, timed = [make_env(configs.wildfire_openness, B, 50, rng='philox', exact_shapes=False) for _ in range(2)]
    for env in (plain, timed):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 5)
    for t in range(n):
        plain.step_random_policy(policy_seed=3, policy_step=t)
    out = (ctypes.c_float * n)()
    _capi.check(timed._lib.frz_wildfire_timed_rollout(timed._handle, 3, 0, n, timed._actions.data_ptr(), _capi.FRZ_RNG_PHILOX,
                                                      stream_ptr(timed.device), out), 'frz_wildfire_timed_rollout')
    assert all(0.0 < out[i] < 5.0 for i in range(n)), list(out)
    for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards', '_task_offsets', '_act_map_offsets'):
        assert torch.equal(getattr(plain, name), getattr(timed, name)), name


@pytest.mark.parametrize('shape,kernel', [((2, 4, 4), 'roles'), ((2, 4, 4), 'lane'), ((1, 7, 3), 'roles'), ((3, 3, 3), 'roles'), ((3, 3, 3), 'lane'),
                                          ((3, 4, 4), 'roles'), ((4, 4, 4), 'roles'), ((4, 4, 2), 'lane'), ((2, 7, 1), 'roles'),
                                          ((2, 3, 6), 'roles'), ((2, 4, 8), 'roles'), ((1, 5, 5), 'roles'), ((2, 2, 7), 'roles'), ((2, 4, 8), 'lane'),
                                          ((4, 4, 6), 'lane'), ((2, 5, 7), 'lane'), ((8, 8, 12), 'lane'), ((5, 9, 16), 'lane')])
def test_every_kernel_variant_matches_the_oracle(oracle, shape, kernel, monkeypatch):
    """Grid shapes that select the runtime-shape variants: <8,4>, <16,4> and <8,8> (field/crew with staged draws, and lane), and the lane
    kernel's <16,8>, <24,8> (wildfire_rich, 4x5, is the other user of it) and <64,16>."""
    monkeypatch.setenv('FRZ_WF_KERNEL', kernel)
    H, Wd, A = shape
    build = lambda: configs.wildfire_grid(H, Wd, A)
    run_against_oracle(oracle, build, {}, 700, 20, 12, seed=31)
    run_against_oracle(oracle, build, dict(show_bad_actions=True, observe_other_power=True), 300, 20, 10, seed=32, rng='philox')
    run_against_oracle(oracle, build, dict(observe_other_suppressant=True), 300, 20, 10, seed=33, policy='device')


@pytest.mark.parametrize('kernel', ['lane', 'roles'])
def test_both_small_grid_kernels_match_the_oracle(oracle, kernel, monkeypatch):
    """Grids of <= 8 cells have two kernels (FRZ_WF_KERNEL): the lane-per-env one and the field/crew wavefront pairs."""
    monkeypatch.setenv('FRZ_WF_KERNEL', kernel)
    run_against_oracle(oracle, configs.wildfire_openness, {}, 1500, 20, 14, seed=21, rng='philox')
    run_against_oracle(oracle, configs.wildfire_openness, dict(show_bad_actions=True), 700, 20, 12, seed=22)
    run_against_oracle(oracle, configs.wildfire_rich, dict(observe_other_suppressant=True), 300, 20, 10, seed=23, rng='philox')


# ------------------------------------------------------------------------------------------------------------
# 3. full-size properties (BASELINE.json config 2: B = 65 536)
# ------------------------------------------------------------------------------------------------------------
def test_full_size_properties():
    B, steps = 65536, 50
    envs = [make_env(configs.wildfire_openness, B, 50, rng='philox', exact_shapes=False) for _ in range(2)]
    for env in envs:
        env.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(steps):
        for env in envs:
            env.step(env.random_policy_actions(policy_seed=99, policy_step=t))
        a, b = envs
        if t % 7 == 0 or t == steps - 1:
            # determinism: two envs, same seeds, same policy stream
            for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards', '_task_offsets', '_act_map_offsets'):
                assert torch.equal(getattr(a, name), getattr(b, name)), name
            off = a._task_offsets
            counts = a.environment_task_count
            # offsets are the exclusive prefix sum of the counts; counts = number of lit cells
            assert int(off[0]) == 0 and torch.equal(off[1:] - off[:-1], counts)
            assert torch.equal(counts, (a._fires > 0).sum(dim=0))
            total = int(off[-1])
            rows = a._task_values[:total]
            # every task row is a lit cell of its env, in row-major order
            env_of_row = torch.repeat_interleave(torch.arange(B, device='cuda'), counts)
            cell = rows[:, 0] * a.max_x + rows[:, 1]
            assert torch.equal(a._fires[cell, env_of_row].long(), rows[:, 2]) and bool((rows[:, 2] > 0).all())
            assert torch.equal(a._intensity[cell, env_of_row].long(), rows[:, 3])
            key = env_of_row * 64 + cell
            assert bool((key[1:] > key[:-1]).all())
            # local indices restart at 0 in every env
            assert torch.equal(a._obs_map_values[:total], torch.arange(total, device='cuda') - off[env_of_row])
            for ag in range(3):
                aoff = a._act_map_offsets[ag]
                assert torch.equal(aoff[1:] - aoff[:-1], a.agent_task_count[ag].long())
                vals = a._act_map_values[ag, :int(aoff[-1])]
                owner = torch.repeat_interleave(torch.arange(B, device='cuda'), a.agent_task_count[ag].long())
                assert bool((vals >= 0).all()) and bool((vals < counts[owner]).all())
                # no suppressant -> empty action mapping
                assert bool((a.agent_task_count[ag][a._suppressants[ag] <= 0] == 0).all())
    assert bool(envs[0].truncated.all()) and int(envs[0].num_moves.min()) == 50
    envs[0].check()


def test_cfg5_batch_on_one_gpu_equals_its_shards():
    """BASELINE.json config 5 (B = 524 288 = 8 x 65 536) stepped on ONE GPU (ticketed path, more chunks than CUs), and the sharding
    property the 8-GPU run rests on: the envs [r * 65 536, (r + 1) * 65 536) of the big batch, seeded with their global index, go through
    exactly what rank r's own 65 536-env batch goes through (state, rewards, task lists), env for env."""
    from free_range_zoo_amd.utils import sharding
    per_rank, ranks, steps = 65536, 8, 12
    B = per_rank * ranks
    big = make_env(configs.wildfire_openness, B, 50, rng='philox', exact_shapes=False)
    big.reset(seed=torch.arange(B, dtype=torch.int32))
    shards = {r: make_env(configs.wildfire_openness, per_rank, 50, rng='philox', exact_shapes=False) for r in (0, 5, 7)}
    for r, env in shards.items():
        env.reset(seed=sharding.shard_seeds(r, per_rank))
    for t in range(steps):
        big.step_random_policy(policy_seed=31, policy_step=t)
        for env in shards.values():
            env.step_random_policy(policy_seed=31, policy_step=t)
    off = big._task_offsets
    counts = big.environment_task_count
    assert int(off[0]) == 0 and torch.equal(off[1:] - off[:-1], counts) and torch.equal(counts, (big._fires > 0).sum(dim=0))
    assert int(big.num_moves.min()) == steps and int(big.num_moves.max()) == steps
    for r, env in shards.items():
        lo, hi = sharding.shard_range(r, per_rank)
        for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_capacity', '_equipment', '_rewards', '_cumulative', '_terminations',
                     'agent_task_count', '_actions'):
            assert torch.equal(getattr(big, name)[:, lo:hi], getattr(env, name)), (r, name)
        assert torch.equal(counts[lo:hi], env.environment_task_count)
        first, last = int(off[lo]), int(off[hi])
        assert last - first == int(env._task_offsets[-1])
        assert torch.equal(big._task_values[first:last], env._task_values[:last - first])
        for ag in range(3):
            a0, a1 = int(big._act_map_offsets[ag, lo]), int(big._act_map_offsets[ag, hi])
            assert torch.equal(big._act_map_values[ag, a0:a1], env._act_map_values[ag, :a1 - a0])
    big.check()


# ------------------------------------------------------------------------------------------------------------
# 4. boundary behaviour
# ------------------------------------------------------------------------------------------------------------
def test_invalid_action_is_flagged(oracle):
    env = make_env(configs.wildfire_non_stochastic, 64, 10)
    env.reset(seed=torch.arange(64, dtype=torch.int32))
    actions = {agent: torch.tensor([[0, 0]], dtype=torch.int32).repeat(64, 1).cuda() for agent in env.agents}
    actions[env.agents[0]][5, 0] = 9  # agent 0 has 2 attackable fires
    env.step(actions)
    with pytest.raises(ValueError):
        env.check()
    env.check()  # flags are cleared once raised


def test_reset_batches_and_initial_state(oracle):
    B = 600
    env = make_env(configs.wildfire_openness, B, 30)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    first = hip_snapshot(env)
    for t in range(12):
        env.step(env.random_policy_actions(policy_seed=5, policy_step=t))
    idx = torch.tensor([0, 3, 255, 256, 599], device='cuda')
    before = hip_snapshot(env)
    env.reset_batches(idx, seed=torch.tensor([0, 3, 255, 256, 599], dtype=torch.int32))
    after = hip_snapshot(env)
    keep = np.ones(B, bool)
    keep[np_(idx)] = False
    for name in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment', 'num_moves', 'env_task_count'):
        G.assert_same(after[name][~keep], first[name][~keep], f'reset_batches restored {name}')
        G.assert_same(after[name][keep], before[name][keep], f'reset_batches untouched {name}')
    assert int(after['num_moves'][~keep].max()) == 0 and not after['truncations'][:, ~keep].any()
    # restart from a saved state: reset(options={'initial_state': ...}) (wildfire.py:341-345)
    saved = env.state().clone()
    env2 = make_env(configs.wildfire_openness, B, 30)
    env2.reset(seed=torch.arange(B, dtype=torch.int32), options={'initial_state': saved})
    s2 = hip_snapshot(env2)
    for name in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment', 'env_task_count', 'agent_task_count',
                 'task_values', 'task_offsets'):
        G.assert_same(s2[name], after[name], f'initial_state {name}')
    with pytest.raises(ValueError):
        env2.reset(options={'initial_state': saved[torch.arange(10, device='cuda')]})


def test_action_space_sampling_is_valid():
    env = make_env(configs.wildfire_openness, 2048, 20)
    env.reset(seed=torch.arange(2048, dtype=torch.int32))
    for t in range(15):
        actions = {}
        for a, agent in enumerate(env.agents):
            space = env.action_space(agent)
            sample = space.sample_nested()
            counts = env.agent_task_count[a].long()
            assert sample.shape == (2048, 2) and sample.dtype == torch.int32
            fight = sample[:, 1] == 0
            assert bool((sample[fight, 0] < counts[fight]).all()) and bool((sample[~fight, 1] == -1).all())
            assert bool((sample[~fight, 0] == counts[~fight]).all())
            actions[agent] = sample
        env.step(actions)
    env.check()
    spaces = env.action_space(env.agents[0]).spaces
    assert len(spaces) == 2048 and len(spaces[0]) == int(env.agent_task_count[0][0]) + 1


@pytest.mark.parametrize('rng', ['philox', 'mt19937'])
def test_graph_replayed_rollout_equals_eager_rollout(rng):
    """capture_random_rollout() replays exactly the launches of the eager loop (reset + policy + step)."""
    B, steps = 3000, 12
    seeds = torch.arange(B, dtype=torch.int32) * 3 + 1
    eager = make_env(configs.wildfire_openness, B, 50, rng=rng, exact_shapes=False)
    eager.reset(seed=seeds)
    for t in range(steps):
        eager.step(eager.random_policy_actions(policy_seed=77, policy_step=t))
    graphed = make_env(configs.wildfire_openness, B, 50, rng=rng, exact_shapes=False)
    graphed.set_exclusive_device(True)  # (philox: the graph then holds the rollout as one multi-step launch)
    graphed.reset(seed=seeds)
    graph = graphed.capture_random_rollout(steps, policy_seed=77, include_reset=True)
    for _ in range(2):  # the second replay starts from the in-graph reset again
        graphed.seeds.copy_(seeds)
        graph.replay()
    torch.cuda.synchronize()
    for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_capacity', '_equipment', '_rewards', '_cumulative', 'num_moves',
                 '_task_offsets', '_act_map_offsets', 'environment_task_count', 'agent_task_count', '_obs_self'):
        assert torch.equal(getattr(eager, name), getattr(graphed, name)), name
    total = int(eager._task_offsets[-1])
    assert torch.equal(eager._task_values[:total], graphed._task_values[:total])
    graphed.check()


def test_env_from_a_configuration_pickled_by_the_reference():
    """The reference distributes its competition configurations as pickles of its Configuration classes: such a file loads through
    utils/compat.py and drives the env like the configuration built by hand."""
    import os
    from free_range_zoo_amd.utils.compat import load_reference_pickle
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_configuration_wildfire.pkl')
    build, kwargs = configs.WILDFIRE_GOLDEN['rich_localized']
    B = 500
    loaded = make_env(lambda: load_reference_pickle(path), B, 20, rng='philox', **kwargs)
    built = make_env(build, B, 20, rng='philox', **kwargs)
    for env in (loaded, built):
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        for t in range(10):
            env.step_random_policy(policy_seed=4, policy_step=t) if hasattr(env, 'step_random_policy') else None
    a, b = hip_snapshot(loaded), hip_snapshot(built)
    for key in a:
        G.assert_same(a[key], b[key], key)
    loaded.check()


def test_update_observations_after_editing_the_state_in_place():
    """Planning-style use: edit env.state() between steps, call update_observations() / update_actions() (the reference's hooks),
    and the published task lists / observations / mappings are those of an env reset to that very state."""
    B = 600
    edited = make_env(configs.wildfire_rich, B, 30, rng='philox')
    edited.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(4):
        edited.step_random_policy(policy_seed=2, policy_step=t)
    state = edited.state()
    state.fires[::3] = state.fires[::3].abs()          # light every fire cell of every third env
    state.intensity[::3] = torch.where(state.fires[::3] > 0, torch.ones_like(state.intensity[::3]), state.intensity[::3])
    state.suppressants[1::4] = 0.0                      # empty tanks: no attackable task for those agents
    edited.update_observations()
    edited.update_actions()
    fresh = make_env(configs.wildfire_rich, B, 30, rng='philox')
    fresh.reset(seed=torch.arange(B, dtype=torch.int32), options={'initial_state': edited.state().clone()})
    a, b = hip_snapshot(edited), hip_snapshot(fresh)
    for key in ('fires', 'intensity', 'fuel', 'suppressants', 'env_task_count', 'agent_task_count', 'task_values', 'task_offsets'):
        G.assert_same(a[key], b[key], key)
    for k in [k for k in a if k.startswith(('act_map', 'obs_'))]:
        G.assert_same(a[k], b[k], k)
    assert int(a['env_task_count'][0]) > int(a['env_task_count'][1]) or int(a['env_task_count'][::3].sum()) > 0


@pytest.mark.parametrize('kernel', ['roles', 'lane'])
@pytest.mark.parametrize('shape', [(3, 3, 3), (3, 3, 4), (2, 4, 3), (4, 2, 4), (3, 4, 3), (4, 3, 4), (4, 4, 3), (2, 8, 4), (2, 2, 2), (2, 2, 3), (2, 4, 2),
                                   (3, 3, 2), (3, 4, 2), (4, 4, 2), (2, 3, 4)])
def test_exact_variants_match_the_oracle_in_every_rng_mode(oracle, shape, kernel, monkeypatch):
    """Grids of 4, 6, 8, 9, 12 and 16 cells with 2 to 4 agents run exact instantiations (FRZ_WF_VARIANT_LIST): loops of their own size,
    Philox and MT19937 inside the step launch (no staging launch), the fused random policy."""
    monkeypatch.setenv('FRZ_WF_KERNEL', kernel)
    build = lambda: configs.wildfire_grid(*shape)
    run_against_oracle(oracle, build, {}, 700, 20, 12, seed=61)
    run_against_oracle(oracle, build, dict(show_bad_actions=True, observe_other_power=True), 600, 20, 12, seed=62, rng='philox')
    run_against_oracle(oracle, build, dict(observe_other_suppressant=True), 500, 60, 30, seed=63, rng='mt19937', policy='device')


@pytest.mark.parametrize('shape', [(2, 4, 8), (2, 3, 6), (3, 4, 3), (2, 2, 2)])
def test_fused_random_policy_on_other_shapes(shape):
    """The fused policy draws agent a from word a % 4 of block a / 4: more than four agents need a second block; exact and runtime-shape
    field/crew instantiations alike leave what policy launch + step launch leave."""
    B = 1200
    build = lambda: configs.wildfire_grid(*shape)
    two, one = [make_env(build, B, 15, rng='philox', exact_shapes=False) for _ in range(2)]
    for env in (two, one):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 2)
    for t in range(18):
        two.step(two.random_policy_actions(policy_seed=8, policy_step=t))
        live = not bool(one.finished.all())
        one.step_random_policy(policy_seed=8, policy_step=t)
        if live:
            assert torch.equal(two._actions, one._actions), f'{shape}: actions at step {t}'
        for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards', '_task_offsets', '_act_map_offsets', '_act_map_values'):
            assert torch.equal(getattr(two, name), getattr(one, name)), f'{shape}: {name} at step {t}'
    one.check()


# ------------------------------------------------------------------------------------------------------------------
# multi-step launches: frz_wildfire_rollout_random_policy as ONE launch (wf_roles_kernel<..., PERSIST>)
# ------------------------------------------------------------------------------------------------------------------
ROLLOUT_FIELDS = ('_fires', '_intensity', '_fuel', '_suppressants', '_capacity', '_equipment', '_rewards', '_cumulative', '_terminations',
                  '_truncations', 'num_moves', 'num_burnouts', '_burnouts', '_putouts', '_obs_self', '_obs_others', '_actions', '_task_offsets',
                  '_act_map_offsets', '_bad_map_offsets', 'environment_task_count', 'agent_task_count', '_frozen_scaled')


def assert_same_env(one, many, what):
    for name in ROLLOUT_FIELDS:
        assert torch.equal(getattr(one, name), getattr(many, name)), f'{what}: {name}'
    total = int(one._task_offsets[-1])
    assert torch.equal(one._task_values[:total], many._task_values[:total]), f'{what}: task rows'
    assert torch.equal(one._obs_map_values[:total], many._obs_map_values[:total]), f'{what}: observation map'
    for a in range(len(one.agents)):
        n = int(one._act_map_offsets[a, -1])
        assert torch.equal(one._act_map_values[a, :n], many._act_map_values[a, :n]), f'{what}: action map of agent {a}'
        if one.show_bad_actions:
            n = int(one._bad_map_offsets[a, -1])
            assert torch.equal(one._bad_map_values[a, :n], many._bad_map_values[a, :n]), f'{what}: bad-action map of agent {a}'


def _openness_with_an_agent_out_of_range():
    """cfg2 with firefighter_1 reaching its own (fire-free) cell only: it has no attackable task in ANY env, the batch-global condition of
    the skip quirk (wildfire.py:434-435) — with show_bad_actions its attacks on listed fires are then ignored instead of penalised.  A
    multi-step launch decodes before the batch totals arrive, assuming nobody is skipped: here that assumption fails at every step."""
    from dataclasses import replace
    cfg = configs.wildfire_openness()
    ranges = cfg.agent_config.attack_range.clone()
    ranges[0] = 0
    cfg.agent_config = replace(cfg.agent_config, attack_range=ranges)
    return cfg


@pytest.mark.parametrize('case', [
    dict(build=configs.wildfire_openness, B=65536, max_steps=50, steps=50),           # the bench workload, one launch per episode
    dict(build=_openness_with_an_agent_out_of_range, B=2500, max_steps=40, steps=11, kwargs=dict(show_bad_actions=True), skipped_agent=0),
    dict(build=configs.wildfire_openness, B=1000, max_steps=50, steps=7),             # ragged last chunk, odd step count
    dict(build=configs.wildfire_openness, B=3000, max_steps=40, steps=12, kwargs=dict(show_bad_actions=True, observe_other_suppressant=True, observe_other_power=True)),
    dict(build=configs.wildfire_rich, B=3000, max_steps=40, steps=6, kwargs=dict(show_bad_actions=True), launches=6),  # 4 x 5 grid: no multi-step kernel, one launch per step
    dict(build=lambda: configs.wildfire_grid(3, 3, 4), B=2048, max_steps=30, steps=9),   # 16-bit cell masks
    dict(build=lambda: configs.wildfire_grid(4, 4, 2), B=700, max_steps=30, steps=10),
], ids=['bench', 'skip_quirk', 'ragged', 'bad_actions', 'fallback_4x5', '3x3a4', '4x4a2'])
def test_multi_step_launch_equals_single_step_launches(case):
    """rollout_random_policy(n) — one launch whose workgroups keep their envs in registers across the n steps — leaves exactly what n
    step_random_policy launches leave: state, rewards, observations, sampled actions, every list."""
    kwargs = dict(rng='philox', exact_shapes=False, **case.get('kwargs', {}))
    one, many = [make_env(case['build'], case['B'], case['max_steps'], **kwargs) for _ in range(2)]
    many.set_exclusive_device(True)
    assert many._lib.frz_wildfire_rollout_launches(many._handle, case['steps'], _capi.FRZ_RNG_PHILOX) == case.get('launches', 1)
    assert one._lib.frz_wildfire_rollout_launches(one._handle, case['steps'], _capi.FRZ_RNG_PHILOX) == case['steps']  # off by default
    for env in (one, many):
        env.reset(seed=torch.arange(case['B'], dtype=torch.int32) + 3)
    for t in range(case['steps']):
        one.step_random_policy(policy_seed=5, policy_step=t)
    many.rollout_random_policy(case['steps'], policy_seed=5, first_step=0)
    assert_same_env(one, many, 'first rollout')
    if 'skipped_agent' in case:  # the quirk was in force: that agent attacked listed fires it cannot reach and was never penalised
        a = case['skipped_agent']
        assert int(one.agent_task_count[a].max()) == 0 and bool((one._actions[a, :, 1] == 0).any())
    # and again from where it stands (policy steps continue), to cover a launch that does not start at a reset
    for t in range(case['steps'], case['steps'] + 3):
        one.step_random_policy(policy_seed=5, policy_step=t)
    many.rollout_random_policy(3, policy_seed=5, first_step=case['steps'])
    assert_same_env(one, many, 'second rollout')
    one.check()
    many.check()


@pytest.mark.parametrize('steps', [8, 9, 5, 6])
def test_multi_step_launch_running_into_the_end_of_the_episode(steps):
    """Every env is truncated after 5 steps: the launch stops stepping there (utils/env.py:211-213), scales the stale rewards once and
    leaves the lists of the last executed step in the caller's buffers whichever copy that step wrote (both parities of the planned
    step count; 5 and 6 end exactly at / one past the horizon)."""
    B = 1500
    one, many = [make_env(configs.wildfire_openness, B, 5, rng='philox', exact_shapes=False) for _ in range(2)]
    many.set_exclusive_device(True)
    for env in (one, many):
        env.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(steps):
        one.step_random_policy(policy_seed=2, policy_step=t)
    many.rollout_random_policy(steps, policy_seed=2, first_step=0)
    assert bool(one.finished.all()) and int(one.num_moves.max()) == 5
    assert_same_env(one, many, f'{steps} steps planned')
    # the next launches are no-ops on both
    one.step_random_policy(policy_seed=2, policy_step=steps)
    many.rollout_random_policy(4, policy_seed=2, first_step=steps)
    assert_same_env(one, many, 'after the end')
    # and a reset starts both again
    for env in (one, many):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 9)
    for t in range(3):
        one.step_random_policy(policy_seed=2, policy_step=t)
    many.rollout_random_policy(3, policy_seed=2, first_step=0)
    assert_same_env(one, many, 'after a reset')


@pytest.mark.parametrize('B,steps,max_steps', [(65536, 50, 50), (1500, 7, 50), (1500, 9, 5)])
def test_rollout_with_metrics_equals_rollout_then_metrics(B, steps, max_steps):
    """frz_wildfire_rollout_random_policy_metrics: the episode reductions made in the tail of the multi-step launch are, bit for bit, the
    float64 sums frz_wildfire_episode_metrics makes afterwards (same summation order) — also when the episode ends mid-launch."""
    from free_range_zoo_amd.utils.env import stream_ptr
    two, one = [make_env(configs.wildfire_openness, B, max_steps, rng='philox', exact_shapes=False, track_cumulative_rewards=True) for _ in range(2)]
    one.set_exclusive_device(True)
    for env in (two, one):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 11)
    m_two = torch.full((len(two.agents) + 2, ), 0.25, dtype=torch.float64, device='cuda')
    m_one = m_two.clone()
    for t in range(steps):
        two.step_random_policy(policy_seed=4, policy_step=t)
    two.accumulate_episode_metrics(m_two)
    assert one._lib.frz_wildfire_rollout_launches(one._handle, steps, _capi.FRZ_RNG_PHILOX) == 1
    _capi.check(one._lib.frz_wildfire_rollout_random_policy_metrics(one._handle, 4, 0, steps, one._actions.data_ptr(), _capi.FRZ_RNG_PHILOX,
                                                                    m_one.data_ptr(), stream_ptr(one.device)), 'frz_wildfire_rollout_random_policy_metrics')
    torch.cuda.synchronize()
    assert torch.equal(m_two, m_one), (m_two, m_one)
    assert float(m_two[len(two.agents)]) == 0.25 + B * min(steps, max_steps)
    # a second episode accumulates on top, like the standalone launch
    two.accumulate_episode_metrics(m_two)
    one.accumulate_episode_metrics(m_one)
    assert torch.equal(m_two, m_one)


@pytest.mark.parametrize('B,steps,max_steps', [(65536, 50, 50), (1000, 7, 50), (1500, 9, 5), (1500, 8, 5)])
def test_multi_step_launch_with_mt19937_streams(B, steps, max_steps):
    """The default RNG (per-env MT19937 streams, bit-identical to the reference's generator) through the multi-step launch: same results and
    the same stream positions as single-step launches — also when the episode ends inside the launch, where the finished batch must leave
    the streams where they are (the reference returns before drawing, utils/env.py:211-213)."""
    one, many = [make_env(configs.wildfire_openness, B, max_steps, rng='mt19937', exact_shapes=False) for _ in range(2)]
    many.set_exclusive_device(True)
    for env in (one, many):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 21)
    assert many._lib.frz_wildfire_rollout_launches(many._handle, steps, _capi.FRZ_RNG_MT19937) == 1
    for t in range(steps):
        one.step_random_policy(policy_seed=6, policy_step=t)
    many.rollout_random_policy(steps, policy_seed=6, first_step=0)
    assert_same_env(one, many, 'mt19937 rollout')
    assert torch.equal(one.generator.generator_index, many.generator.generator_index), 'stream positions'
    assert torch.equal(one.generator.generator_states, many.generator.generator_states), 'stream states'
    assert int(one.num_moves.max()) == min(steps, max_steps)
    one.check()
    many.check()


EXACT_SHAPES = [(2, 3, 3), (2, 3, 2), (3, 3, 3), (3, 3, 4), (2, 4, 3), (2, 4, 4), (3, 4, 3), (3, 4, 4), (4, 4, 3), (4, 4, 4), (2, 2, 2), (2, 2, 3),
                (2, 4, 2), (3, 3, 2), (3, 4, 2), (4, 4, 2), (2, 3, 4)]


RUNTIME_SHAPES = [(1, 7, 3), (3, 5, 2), (2, 5, 4), (1, 5, 1), (2, 7, 4), (1, 3, 2), (2, 4, 8), (1, 6, 5), (2, 3, 7), (3, 4, 5), (4, 4, 6),
                  (3, 3, 8)]  # <8, 4>, <16, 4>, <8, 8> and <16, 8>


@pytest.mark.parametrize('shape', RUNTIME_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_multi_step_launch_of_runtime_shapes(shape, monkeypatch):
    """Grids without an exact instantiation run the runtime-shape field/crew variants (<8, 4>, <16, 4>, <8, 8>, <16, 8>: H * W and A read from the
    configuration); since round 4 those have a multi-step kernel too — in-kernel Philox and in-kernel MT19937 streams with the draw numbers
    resolved at run time.  Each against single-step launches (which read STAGED draws: wf_philox_fill_kernel / frz_mt19937_generate), fully
    stochastic configuration, ragged batch, both RNG modes."""
    monkeypatch.setenv('FRZ_WF_MULTI_STEP', 'all')  # (by default the library takes the multi-step kernel of a runtime shape only where it is the faster way)
    H, Wd, A = shape
    B = 777
    for rng, mode in (('philox', _capi.FRZ_RNG_PHILOX), ('mt19937', _capi.FRZ_RNG_MT19937)):
        build = lambda: configs.wildfire_grid(H, Wd, A, seed=H * 5 + Wd * 3 + A)  # noqa: E731
        one, many = [make_env(build, B, 30, rng=rng, exact_shapes=False) for _ in range(2)]
        many.set_exclusive_device(True)
        assert many._lib.frz_wildfire_rollout_launches(many._handle, 7, mode) == 1, f'{shape} has no multi-step launch'
        for env in (one, many):
            env.reset(seed=torch.arange(B, dtype=torch.int32) + 2)
        for t in range(7):
            one.step_random_policy(policy_seed=9, policy_step=t)
        many.rollout_random_policy(7, policy_seed=9, first_step=0)
        assert_same_env(one, many, f'{shape} {rng}')
        if rng == 'mt19937':
            assert torch.equal(one.generator.generator_index, many.generator.generator_index), 'stream positions'
            assert torch.equal(one.generator.generator_states, many.generator.generator_states), 'stream states'
        many.check()


@pytest.mark.parametrize('shape', EXACT_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_multi_step_launch_of_every_exact_shape(shape):
    """Every shape with an exact field/crew instantiation (FRZ_WF_VARIANT_LIST) has a multi-step kernel: each against single-step launches,
    fully stochastic configuration, ragged batch, both RNG modes that draw in-kernel."""
    H, Wd, A = shape
    B = 777
    for rng, mode in (('philox', _capi.FRZ_RNG_PHILOX), ('mt19937', _capi.FRZ_RNG_MT19937)):
        build = lambda: configs.wildfire_grid(H, Wd, A, seed=H * 7 + Wd * 3 + A)  # noqa: E731
        one, many = [make_env(build, B, 30, rng=rng, exact_shapes=False) for _ in range(2)]
        many.set_exclusive_device(True)
        assert many._lib.frz_wildfire_rollout_launches(many._handle, 6, mode) == 1, f'{shape} has no multi-step launch'
        for env in (one, many):
            env.reset(seed=torch.arange(B, dtype=torch.int32) + 2)
        for t in range(6):
            one.step_random_policy(policy_seed=9, policy_step=t)
        many.rollout_random_policy(6, policy_seed=9, first_step=0)
        assert_same_env(one, many, f'{shape} {rng}')
        many.check()
