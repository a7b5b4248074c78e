"""GPU: the multi-step launches (frz_wildfire_rollout — the kernel bench.py times) compared DIRECTLY with the oracle and with the
reference's recorded trajectories, step by step: every step's sampled actions, rewards, flags and packed lists are kept by the launch
(`record=True`: reward / done / action tapes and the list record) and read back; nothing here goes through single-step launches.

Also: the action-tape mode (the golden trajectories of the unmodified reference replayed through ONE launch), the opening reset folded
into the launch, device-side auto-reset against `step(); reset_batches(finished)` of the oracle, and `reset_finished(mask)` against the
reference's partial-reset recordings."""
import ctypes

import numpy as np
import pytest
import torch

import configs
import golden_util as G
from free_range_zoo_amd import _capi
from test_hip_wildfire import compare_snapshots, hip_snapshot, make_env, np_, oracle_snapshot

pytestmark = pytest.mark.gpu


def list_views(env, block: np.ndarray):
    """The arrays of one packed-list block (a copy of the env's own block, uint8) by the names the snapshots use."""
    A, B = len(env.agents), env.parallel_envs
    cap = B * env.max_y * env.max_x
    base = env._arena.data_ptr() + env._list_block_offset
    bufs = env._bufs

    def view(ptr, count, dtype):
        off = ptr - base
        return block[off:off + count * np.dtype(dtype).itemsize].view(dtype)

    out = {'task_offsets': view(bufs.task_offsets, B + 1, np.int64)}
    total = int(out['task_offsets'][-1])
    out['task_values'] = view(bufs.task_values, cap * 4, np.int64).reshape(cap, 4)[:total]
    out['obs_map_values'] = view(bufs.obs_map_values, cap, np.int64)[:total]
    for a in range(A):
        off = view(bufs.act_map_offsets, A * (B + 1), np.int64).reshape(A, B + 1)[a]
        out[f'act_map_offsets_{a}'] = off
        out[f'act_map_values_{a}'] = view(bufs.act_map_values, A * cap, np.int64).reshape(A, cap)[a, :int(off[-1])]
        if env.show_bad_actions:
            off = view(bufs.bad_map_offsets, A * (B + 1), np.int64).reshape(A, B + 1)[a]
            out[f'bad_map_offsets_{a}'] = off
            out[f'bad_map_values_{a}'] = view(bufs.bad_map_values, A * cap, np.int64).reshape(A, cap)[a, :int(off[-1])]
    return out


def oracle_lists(o, show_bad):
    snap = oracle_snapshot(o)
    keys = ['task_offsets', 'task_values']
    out = {k: snap[k].copy() for k in keys}
    out['obs_map_values'] = o.obs_map_values[:o.total_tasks()].copy()
    for a in range(o.cfg.num_agents):
        v, off = o.action_map(a)
        out[f'act_map_values_{a}'], out[f'act_map_offsets_{a}'] = v.copy(), off.copy()
        if show_bad:
            v, off = o.bad_map(a)
            out[f'bad_map_values_{a}'], out[f'bad_map_offsets_{a}'] = v.copy(), off.copy()
    return out


def prepare(env, rec):
    env._list_block_offset = rec['list_block_offset']


def compare_lists(got, want, what):
    for key, w in want.items():
        G.assert_same(got[key], w, f'{what} {key}')


# ------------------------------------------------------------------------------------------------------------
# 1. the bench's launch against the oracle: policy sampled in-kernel, Philox draws, 50 steps, one launch
# ------------------------------------------------------------------------------------------------------------
ORACLE_CASES = [
    dict(build=configs.wildfire_openness, B=65536, max_steps=50, steps=50, kwargs={}),                                   # cfg2 as bench.py runs it
    dict(build=configs.wildfire_openness, B=1000, max_steps=50, steps=50, kwargs={}),                                    # ragged last chunk
    dict(build=configs.wildfire_openness, B=3001, max_steps=30, steps=34, kwargs=dict(show_bad_actions=True, observe_other_suppressant=True)),
    dict(build=lambda: configs.wildfire_grid(3, 3, 4, seed=5), B=2049, max_steps=25, steps=20, kwargs={}),               # 16-bit cell masks
    # runtime shapes (no exact instantiation: <8, 4> and <16, 4> with H * W and A read from the configuration): in-kernel Philox through the
    # per-env scratch column (round 4)
    dict(build=lambda: configs.wildfire_grid(1, 7, 3, seed=2), B=1500, max_steps=30, steps=24, kwargs={}),
    dict(build=lambda: configs.wildfire_grid(3, 5, 2, seed=4), B=700, max_steps=12, steps=16, kwargs=dict(show_bad_actions=True, observe_other_power=True)),
]


@pytest.mark.parametrize('case', ORACLE_CASES, ids=['cfg2_B65536', 'cfg2_ragged', 'bad_actions_past_the_horizon', '3x3a4', 'runtime_1x7a3', 'runtime_3x5a2_past_the_horizon'])
def test_multi_step_launch_against_the_oracle(oracle, case, monkeypatch):
    """rollout(n) as ONE launch (frz_wildfire_rollout_launches == 1) vs n oracle steps: the sampled actions, rewards, terminations /
    truncations and every packed list OF EVERY STEP (tapes + list record), then the whole final state."""
    monkeypatch.setenv('FRZ_WF_MULTI_STEP', 'all')  # (runtime shapes: the multi-step kernel wherever it exists, not only where it is the default)
    check_policy_rollout_against_the_oracle(oracle, case, one_launch=True)


def check_policy_rollout_against_the_oracle(oracle, case, one_launch):
    """(also driven by tests/test_hip_fuzz.py over random shapes; one_launch=None: whatever the library decides for the shape)"""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    B, steps, kwargs = case['B'], case['steps'], case['kwargs']
    flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
    flags.update(kwargs)
    cfg = to_cstruct(case['build'](), B, case['max_steps'], **flags)
    env = make_env(case['build'], B, case['max_steps'], rng='philox', **kwargs)
    if one_launch is not False:
        env.set_exclusive_device(True)
    if one_launch:
        assert env._lib.frz_wildfire_rollout_launches(env._handle, steps, _capi.FRZ_RNG_PHILOX) == 1
    seeds = (torch.arange(B, dtype=torch.int32) * 3 + 11)
    env.reset(seed=seeds)
    rec = env.rollout(steps, policy_seed=77, first_step=0, record=True)
    prepare(env, rec)
    torch.cuda.synchronize()
    rewards, dones, actions, lists = (np_(rec[k]) for k in ('rewards', 'dones', 'actions', 'lists'))

    o = oracle.WildfireOracle(cfg)
    o.reset()
    executed = 0
    for t in range(steps):
        if bool(o.terminations[0].all() or o.truncations[0].all()):
            break  # utils/env.py:211-213: the launch stops stepping here too
        acts = oracle.wildfire_random_policy(cfg, o.agent_task_count, o.env_task_count, seeds.numpy(), 77, t)
        fr, ar = oracle.wildfire_philox_randomness(cfg, seeds.numpy(), o.num_moves)
        o.step(acts, fr, ar)
        executed = t + 1
        G.assert_same(actions[t], acts, f'step {t} sampled actions')
        G.assert_same(rewards[t], o.rewards, f'step {t} rewards')
        G.assert_same(dones[t, 0].astype(bool), o.terminations[0].astype(bool), f'step {t} terminations')
        G.assert_same(dones[t, 1].astype(bool), o.truncations[0].astype(bool), f'step {t} truncations')
        if t < steps - 1 and not bool(o.terminations[0].all() or o.truncations[0].all()):
            compare_lists(list_views(env, lists[t]), oracle_lists(o, env.show_bad_actions), f'step {t} (list record)')
    assert executed == min(steps, case['max_steps']) or (one_launch is not True and bool(o.terminations[0].all() or o.truncations[0].all()))  # (every env of a small random batch may burn out early)
    if executed < steps:  # the frozen steps that follow scale the stale rewards once (utils/conversions.py:87-90)
        acts = np.zeros((cfg.num_agents, B, 2), np.int32)
        o.step(acts, np.zeros((3, B, cfg.grid_height * cfg.grid_width), np.float32), np.zeros((5, B, cfg.num_agents), np.float32))
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), f'after {steps} steps in one launch')
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 2. the reference's recorded trajectories through the multi-step launch (action tape + randomness tapes)
# ------------------------------------------------------------------------------------------------------------
def golden_tapes(data, cfg, steps):
    B, A, HW = cfg.parallel_envs, cfg.num_agents, cfg.grid_height * cfg.grid_width
    acts = np.zeros((steps, A, B, 2), np.int32)
    field, agent = np.zeros((steps, 3, B, HW), np.float32), np.zeros((steps, 5, B, A), np.float32)
    for t in range(steps):
        p = f's{t}_'
        acts[t] = data[p + 'actions']
        if bool(data[p + 'stepped']):  # (a frozen step drew nothing: whatever the tape holds there must not matter)
            field[t], agent[t] = data[p + 'field_randomness'].reshape(3, B, HW), data[p + 'agent_randomness']
    return torch.from_numpy(acts).cuda(), torch.from_numpy(field).cuda(), torch.from_numpy(agent).cuda()


@pytest.mark.parametrize('name', sorted(configs.WILDFIRE_GOLDEN))
def test_golden_trajectory_through_the_multi_step_launch(name):
    """Every recorded trajectory of the unmodified reference replayed by `rollout(actions=tape, randomness=tapes)`: (a) the whole
    trajectory as ONE launch with every step's rewards / flags / lists recorded and compared; (b) launches of 1 .. T steps from the reset,
    each compared with the reference's full snapshot after that step (state, observations, bookkeeping, lists)."""
    build, kwargs = configs.WILDFIRE_GOLDEN[name]
    data = np.load(G.golden_path(f'traj_wildfire_{name}.npz'))
    cfg = G.load_cfg(data, _capi.frz_wildfire_cfg)
    B, A, T = cfg.parallel_envs, cfg.num_agents, int(data['steps'])
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.set_exclusive_device(True)
    one_launch = env._lib.frz_wildfire_rollout_launches(env._handle, T, _capi.FRZ_RNG_INJECTED) == 1
    exact_shape = (cfg.grid_height * cfg.grid_width, A) in {(6, 3), (6, 2), (9, 3), (9, 4), (8, 3), (8, 4), (12, 3), (12, 4), (16, 3), (16, 4),
                                                            (4, 2), (4, 3), (8, 2), (9, 2), (12, 2), (16, 2), (6, 4)}
    assert one_launch == exact_shape, 'every exact field/crew shape has a multi-step launch with injected randomness'
    acts, field, agent = golden_tapes(data, cfg, T)
    seeds = torch.arange(B, dtype=torch.int32)
    # (a) one launch, everything recorded
    env.reset(seed=seeds)
    rec = env.rollout(T, actions=acts, randomness=(field, agent), record=True)
    prepare(env, rec)
    rewards, dones, lists = (np_(rec[k]) for k in ('rewards', 'dones', 'lists'))
    for t in range(T):
        p = f's{t}_'
        if not bool(data[p + 'stepped']):
            break  # the launch stopped stepping (utils/env.py:211-213); (b) checks what the frozen steps leave
        G.assert_same(rewards[t], data[p + 'rewards'], f'{name} step {t} reward tape', G.REWARD_RTOL)
        G.assert_same(dones[t, 0].astype(bool), data[p + 'terminations'][0], f'{name} step {t} termination tape')
        G.assert_same(dones[t, 1].astype(bool), data[p + 'truncations'][0], f'{name} step {t} truncation tape')
        if t < T - 1 and bool(data[f's{t + 1}_stepped']):
            got = list_views(env, lists[t])
            want = {'task_offsets': data[p + 'task_offsets'], 'task_values': data[p + 'task_values']}
            for a in range(A):
                for key in ('act_map_values', 'act_map_offsets') + (('bad_map_values', 'bad_map_offsets') if env.show_bad_actions else ()):
                    if f'{p}{key}_{a}' in data.files:
                        want[f'{key}_{a}'] = data[f'{p}{key}_{a}']
            if env.show_bad_actions:  # (the action mapping then lists every fire: wildfire.py:656-660; what the record holds are the attackable ones)
                want = {k: v for k, v in want.items() if not k.startswith('act_map')}
            compare_lists(got, want, f'{name} step {t} (list record)')
    G.compare_wildfire(hip_snapshot(env), data, f's{T - 1}_', A, f'{name} after the whole trajectory in one launch')
    # (b) prefixes
    for n in range(1, T + 1):
        env.reset(seed=seeds)
        env.rollout(n, actions=acts[:n].contiguous(), randomness=(field[:n].contiguous(), agent[:n].contiguous()))
        G.compare_wildfire(hip_snapshot(env), data, f's{n - 1}_', A, f'{name}: {n} steps in one launch')
        G.assert_same(np_(env.finished), data[f's{n - 1}_finished'], f'{name}: {n} steps, finished')
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 2b. rollouts a learner can consume: every step's OBSERVATIONS and STATE beside rewards / dones / actions, against the
#     reference's recordings (the loop of utils/conversions.py:92-99 hands the observations back at every step)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('form', ['compact', 'full'])
@pytest.mark.parametrize('name', sorted(configs.WILDFIRE_GOLDEN))
def test_golden_trajectory_observation_and_state_tapes(name, form):
    """`rollout(..., record=True, record_observations=form, record_state=True)` over a recorded trajectory of the unmodified reference: step
    t's `{agent: self, others, tasks}` rebuilt from the tapes (`recorded_observations`) and its state (`recorded_state`) equal what the
    reference's t-th step() returned / left.  'compact' (the suppressant column only) rides in the ONE multi-step launch where the shape has
    one; 'full' copies the observation block out between per-step launches."""
    build, kwargs = configs.WILDFIRE_GOLDEN[name]
    data = np.load(G.golden_path(f'traj_wildfire_{name}.npz'))
    cfg = G.load_cfg(data, _capi.frz_wildfire_cfg)
    B, A, T = cfg.parallel_envs, cfg.num_agents, int(data['steps'])
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.set_exclusive_device(True)
    acts, field, agent = golden_tapes(data, cfg, T)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    rec = env.rollout(T, actions=acts, randomness=(field, agent), record=True, record_observations=form, record_state=True)
    for t in range(T):
        p = f's{t}_'
        if not bool(data[p + 'stepped']):
            break  # (a multi-step launch stops stepping with the batch: what it leaves for later steps is unspecified)
        state = env.recorded_state(rec, t)
        for key in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment'):
            want = data[p + key]
            G.assert_same(np_(getattr(state, key)).reshape(want.shape), want, f'{name} step {t} state tape {key}')
        if t < T - 1 and not bool(data[f's{t + 1}_stepped']):
            continue  # (its lists were written once more into the env's own buffers, not into the record: see the list record test)
        obs = env.recorded_observations(rec, t)
        for a, agent_name in enumerate(env.agents):
            G.assert_same(np_(obs[agent_name]['self']), data[f'{p}obs_self_{a}'], f'{name} step {t} {form} observations: self[{a}]')
            G.assert_same(np_(obs[agent_name]['others']), data[f'{p}obs_others_{a}'], f'{name} step {t} {form} observations: others[{a}]')
            G.assert_same(np_(obs[agent_name]['tasks'].values()), data[p + 'task_values'], f'{name} step {t} {form} observations: tasks')
            G.assert_same(np_(obs[agent_name]['tasks'].offsets()), data[p + 'task_offsets'], f'{name} step {t} {form} observations: task offsets')
    env.check()


@pytest.mark.parametrize('family', ['roles', 'lane', 'grid'])
@pytest.mark.parametrize('auto_reset', [False, True], ids=['episodic', 'auto_reset'])
def test_observation_and_state_tapes_against_the_oracle(oracle, family, auto_reset, monkeypatch):
    """The tapes of a policy-driven rollout (in-kernel policy, Philox draws) in every wildfire kernel family, with and without the
    device-side restart: state and suppressant tapes against the oracle's per-step loop (a restarted env shows its fresh state)."""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    monkeypatch.setenv('FRZ_WF_KERNEL', family)
    B, horizon, steps, stride = (65536 if family == 'roles' else 3000), 9, 23, 1000003
    cfg = to_cstruct(configs.wildfire_openness(), B, horizon, track_cumulative_rewards=True, observe_other_suppressant=True)
    env = make_env(configs.wildfire_openness, B, horizon, rng='philox', track_cumulative_rewards=True, observe_other_suppressant=True)
    env.set_exclusive_device(True)
    one_launch = env._lib.frz_wildfire_rollout_launches(env._handle, steps, _capi.FRZ_RNG_PHILOX) == 1
    assert one_launch == (family == 'roles')
    seeds = torch.arange(B, dtype=torch.int32) * 7 + 3
    env.reset(seed=seeds)
    rec = env.rollout(steps, policy_seed=19, auto_reset=auto_reset, seed_stride=stride, record=True, record_observations='compact', record_state=True)
    o = oracle.WildfireOracle(cfg)
    o.reset()
    s = seeds.numpy().copy()
    supp_tape = np_(rec['observations'])
    for t in range(steps):
        if not auto_reset and bool((o.terminations[0].astype(bool) | o.truncations[0].astype(bool)).all()):
            break
        acts = oracle.wildfire_random_policy(cfg, o.agent_task_count, o.env_task_count, s, 19, t)
        fr, ar = oracle.wildfire_philox_randomness(cfg, s, o.num_moves)
        o.step(acts, fr, ar)
        if auto_reset:
            o.reset_masked(None, s, stride)
        state = env.recorded_state(rec, t)
        for key in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment'):
            want = getattr(o, key)
            G.assert_same(np_(getattr(state, key)).reshape(want.shape), want, f'{family} step {t} state tape {key}')
        G.assert_same(supp_tape[t], o.suppressants.T, f'{family} step {t} suppressant tape')
        if t in (0, 7, steps - 1):
            obs = env.recorded_observations(rec, t)
            want_self, want_others = oracle_observations(o)
            for a, agent_name in enumerate(env.agents):
                G.assert_same(np_(obs[agent_name]['self']), want_self[a], f'{family} step {t}: self[{a}] rebuilt from the compact tape')
                G.assert_same(np_(obs[agent_name]['others']), want_others[a], f'{family} step {t}: others[{a}] rebuilt from the compact tape')
    assert t > horizon or not auto_reset
    env.check()


def oracle_observations(o):
    """(self [A][B][4], others [A][B][A - 1][k]) of the oracle's current step, as numpy."""
    snap = oracle_snapshot(o)
    A = o.cfg.num_agents
    return [snap[f'obs_self_{a}'] for a in range(A)], [snap[f'obs_others_{a}'] for a in range(A)]


@pytest.mark.parametrize('name', sorted(configs.CYBER_GOLDEN))
def test_cybersecurity_golden_trajectory_observation_and_state_tapes(name):
    """Cybersecurity: every step's observation rows (self / others / tasks of every agent) and state from the tapes against the reference's
    recording (the tapes are copied out between per-step launches in this domain)."""
    import test_hip_cybersecurity as C
    build, kwargs = configs.CYBER_GOLDEN[name]
    data = np.load(G.golden_path(f'traj_cybersecurity_{name}.npz'))
    cfg = G.load_cfg(data, _capi.frz_cybersecurity_cfg)
    B, N, A, T = cfg.parallel_envs, cfg.num_nodes, cfg.num_attackers + cfg.num_defenders, int(data['steps'])
    env = C.make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.set_exclusive_device(True)
    acts = np.zeros((T, A, B, 2), np.int32)
    net, agent = np.zeros((T, B, N), np.float32), np.zeros((T, B, A), np.float32)
    for t in range(T):
        acts[t] = data[f's{t}_actions']
        if bool(data[f's{t}_stepped']):
            net[t], agent[t] = data[f's{t}_network_randomness'].reshape(B, N), data[f's{t}_agent_randomness'].reshape(B, A)
    acts, net, agent = torch.from_numpy(acts).cuda(), torch.from_numpy(net).cuda(), torch.from_numpy(agent).cuda()
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    rec = env.rollout(T, actions=acts, randomness=(net, agent), record=True, record_observations='full', record_state=True)
    for t in range(T):
        p = f's{t}_'
        if not bool(data[p + 'stepped']):
            break
        state = env.recorded_state(rec, t)
        G.assert_same(np_(state.network_state), data[p + 'network_state'], f'{name} step {t} state tape: network_state')
        G.assert_same(np_(state.location), data[p + 'location'], f'{name} step {t} state tape: location')
        G.assert_same(np_(state.presence).astype(bool), data[p + 'presence'].astype(bool), f'{name} step {t} state tape: presence')
        obs = env.recorded_observations(rec, t)
        for a, agent_name in enumerate(env.agents):
            for part in ('self', 'others', 'tasks'):
                G.assert_same(np_(obs[agent_name][part]), data[f'{p}obs_{part}_{a}'], f'{name} step {t} observation tape: {part}[{a}]')
    with pytest.raises(ValueError, match='compact'):
        env.rollout(2, policy_seed=1, record_observations='compact')
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 3. the opening reset inside the launch
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B', [65536, 1000])
def test_rollout_with_the_reset_folded_in_equals_reset_then_rollout(oracle, B):
    """FRZ_ROLLOUT_RESET_FIRST: `reset(seed + increment); rollout(n)` as one launch, from whatever state the env is in; checked against
    the oracle started from a reset with the moved-on seeds."""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    cfg = to_cstruct(configs.wildfire_openness(), B, 50)
    env = make_env(configs.wildfire_openness, B, 50, rng='philox', track_cumulative_rewards=True)
    env.set_exclusive_device(True)
    seeds = torch.arange(B, dtype=torch.int32) + 5
    env.reset(seed=seeds)
    env.rollout(13, policy_seed=3)  # somewhere in the middle of an episode
    metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device='cuda')
    env.rollout(21, policy_seed=9, reset_first=True, seed_increment=1000003, metrics=metrics)
    new_seeds = (seeds.numpy().astype(np.int64) + 1000003).astype(np.int32)
    assert np.array_equal(np_(env.seeds), new_seeds)
    o = oracle.WildfireOracle(cfg)
    o.reset()
    o.rollout(new_seeds, 9, 0, 21)
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), 'reset folded into the launch')
    want = np.concatenate([o.cumulative_rewards.astype(np.float64).sum(axis=1), [float(o.num_moves.sum())],
                           [float((o.terminations[0].astype(bool) | o.truncations[0].astype(bool)).sum())]])
    np.testing.assert_allclose(metrics.cpu().numpy(), want, rtol=1e-12)
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 4. device-side auto-reset: step(); reset_batches(finished) — in one launch, and launch by launch
# ------------------------------------------------------------------------------------------------------------
def oracle_auto_reset_rollout(oracle, cfg, seeds, policy_seed, steps, stride, show_bad):
    o = oracle.WildfireOracle(cfg)
    o.reset()
    seeds = seeds.copy()
    A = cfg.num_agents
    out = dict(actions=[], rewards=[], term=[], trunc=[], lists=[])
    returns, ended = np.zeros(A, np.float64), 0
    for t in range(steps):
        acts = oracle.wildfire_random_policy(cfg, o.agent_task_count, o.env_task_count, seeds, policy_seed, t)
        fr, ar = oracle.wildfire_philox_randomness(cfg, seeds, o.num_moves)
        o.step(acts, fr, ar)
        out['actions'].append(acts.copy()), out['rewards'].append(o.rewards.copy())
        out['term'].append(o.terminations[0].astype(bool).copy()), out['trunc'].append(o.truncations[0].astype(bool).copy())
        finished = out['term'][-1] | out['trunc'][-1]
        returns += o.cumulative_rewards[:, finished].astype(np.float64).sum(axis=1)
        ended += int(finished.sum())
        o.reset_masked(None, seeds, stride)
        out['lists'].append(oracle_lists(o, show_bad))
    out['metrics'] = np.concatenate([returns, [float(steps * cfg.parallel_envs)], [float(ended)]])
    return o, seeds, out


@pytest.mark.parametrize('exclusive', [True, False], ids=['one_launch', 'launch_per_step'])
@pytest.mark.parametrize('case', [dict(B=65536, max_steps=50, steps=120, kwargs={}), dict(B=1500, max_steps=7, steps=40, kwargs={}),
                                  dict(B=777, max_steps=9, steps=25, kwargs=dict(show_bad_actions=True)),
                                  dict(B=601, max_steps=6, steps=20, kwargs=dict(observe_other_suppressant=True), build=lambda: configs.wildfire_grid(8, 8, 5, seed=2)),
                                  dict(B=515, max_steps=5, steps=16, kwargs=dict(show_bad_actions=True), build=configs.wildfire_rich_plain)],
                         ids=['cfg2_B65536', 'short_horizon_ragged', 'bad_actions', 'grid_family_8x8', 'lane_family_4x5'])
def test_auto_reset_against_the_oracle(oracle, case, exclusive):
    """Continuous rollouts at fixed B: an env that finishes at step t restarts inside step t with seed + stride.  Equals
    the oracle's `step(); reset_batches(finished, seed + stride)` loop: every step's actions, rewards, flags (as the step set them), the
    lists (as the reset left them), the final state, the seeds, and the returns of the episodes that ended."""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    B, steps, kwargs, stride = case['B'], case['steps'], case['kwargs'], 1000003
    if B == 65536 and not exclusive:
        steps = 60
    flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
    flags.update(kwargs)
    build = case.get('build', configs.wildfire_openness)  # (the other families have no multi-step kernel: one launch sequence per step)
    cfg = to_cstruct(build(), B, case['max_steps'], track_cumulative_rewards=True, **flags)
    env = make_env(build, B, case['max_steps'], rng='philox', track_cumulative_rewards=True, **kwargs)
    env.set_exclusive_device(exclusive)
    assert env._lib.frz_wildfire_rollout_launches(env._handle, steps, _capi.FRZ_RNG_PHILOX) == (1 if exclusive and 'build' not in case else steps)
    seeds = torch.arange(B, dtype=torch.int32) * 5 + 1
    env.reset(seed=seeds)
    metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device='cuda')
    rec = env.rollout(steps, policy_seed=41, auto_reset=True, seed_stride=stride, record=True, metrics=metrics)
    prepare(env, rec)
    torch.cuda.synchronize()
    o, want_seeds, want = oracle_auto_reset_rollout(oracle, cfg, seeds.numpy(), 41, steps, stride, env.show_bad_actions)
    rewards, dones, actions, lists = (np_(rec[k]) for k in ('rewards', 'dones', 'actions', 'lists'))
    for t in range(steps):
        G.assert_same(actions[t], want['actions'][t], f'step {t} sampled actions')
        G.assert_same(rewards[t], want['rewards'][t], f'step {t} rewards')
        G.assert_same(dones[t, 0].astype(bool), want['term'][t], f'step {t} terminations')
        G.assert_same(dones[t, 1].astype(bool), want['trunc'][t], f'step {t} truncations')
        if t < steps - 1 and (t % 7 == 0 or B < 5000):
            compare_lists(list_views(env, lists[t]), want['lists'][t], f'step {t} (list record)')
    assert sum(int((a | b).sum()) for a, b in zip(want['term'], want['trunc'])) > B // 2, 'the case must actually reset envs'
    G.assert_same(np_(env.seeds), want_seeds, 'seeds')
    # the final state is the oracle's after its last reset_batches (rewards / flags of the envs it reset zeroed, utils/env.py:176-188)
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), 'after the rollout')
    np.testing.assert_allclose(metrics.cpu().numpy(), want['metrics'], rtol=1e-9)
    env.check()


def test_auto_reset_continues_across_launches(oracle):
    """A second auto-reset launch picks up where the first stopped (here: right after a step that reset every env of the batch)."""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    B, stride = 2000, 17
    cfg = to_cstruct(configs.wildfire_openness(), B, 6, track_cumulative_rewards=True)
    env = make_env(configs.wildfire_openness, B, 6, rng='philox', track_cumulative_rewards=True)
    env.set_exclusive_device(True)
    seeds = torch.arange(B, dtype=torch.int32)
    env.reset(seed=seeds)
    env.rollout(6, policy_seed=8, auto_reset=True, seed_stride=stride)   # every env is truncated (and reset) by the last step
    assert not bool(env.finished.any())  # every env that hit the horizon (or burnt out earlier) started over
    env.rollout(9, policy_seed=8, first_step=6, auto_reset=True, seed_stride=stride)
    o = oracle.WildfireOracle(cfg)
    o.reset()
    s = seeds.numpy().copy()
    for t in range(15):
        acts = oracle.wildfire_random_policy(cfg, o.agent_task_count, o.env_task_count, s, 8, t)
        fr, ar = oracle.wildfire_philox_randomness(cfg, s, o.num_moves)
        o.step(acts, fr, ar)
        o.reset_masked(None, s, stride)
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), 'two auto-reset launches')
    G.assert_same(np_(env.seeds), s, 'seeds')


# ------------------------------------------------------------------------------------------------------------
# 4b. ONE-step rollouts honour every option of the spec (`frz_*_rollout_launches(env, 1, mode)` returns n_steps = 1, which
#     used to send a one-step rollout down the multi-step branch, where the single-step kernel ignored the spec)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('family', ['roles', 'lane', 'grid'])
def test_one_step_rollouts_honour_every_option(oracle, family, monkeypatch):
    """rollout(1, auto_reset=True, record=True, metrics=...) step by step against the oracle's `step(); reset_batches(finished)` loop, then
    rollout(1, reset_first=True, ...): in each wildfire kernel family, with the device declared exclusive (the case that went wrong)."""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    monkeypatch.setenv('FRZ_WF_KERNEL', family)
    B, horizon, steps, stride = 1500, 4, 11, 1000003
    cfg = to_cstruct(configs.wildfire_openness(), B, horizon, track_cumulative_rewards=True)
    env = make_env(configs.wildfire_openness, B, horizon, rng='philox', track_cumulative_rewards=True)
    env.set_exclusive_device(True)
    seeds = torch.arange(B, dtype=torch.int32) * 3 + 4
    env.reset(seed=seeds)
    metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device='cuda')
    o, want_seeds, want = oracle_auto_reset_rollout(oracle, cfg, seeds.numpy(), 23, steps, stride, False)
    replay = oracle.WildfireOracle(cfg)
    replay.reset()
    s = seeds.numpy().copy()
    for t in range(steps):
        rec = env.rollout(1, policy_seed=23, first_step=t, auto_reset=True, seed_stride=stride, record=True, metrics=metrics)
        assert rec['lists'].shape[0] == 0  # (n - 1 copies: the step's own lists are in the env's buffers)
        G.assert_same(np_(rec['actions'])[0], want['actions'][t], f'{family} step {t} sampled actions')
        G.assert_same(np_(rec['rewards'])[0], want['rewards'][t], f'{family} step {t} rewards')
        G.assert_same(np_(rec['dones'])[0, 0].astype(bool), want['term'][t], f'{family} step {t} terminations')
        G.assert_same(np_(rec['dones'])[0, 1].astype(bool), want['trunc'][t], f'{family} step {t} truncations')
        acts = oracle.wildfire_random_policy(cfg, replay.agent_task_count, replay.env_task_count, s, 23, t)
        fr, ar = oracle.wildfire_philox_randomness(cfg, s, replay.num_moves)
        replay.step(acts, fr, ar)
        replay.reset_masked(None, s, stride)
        compare_snapshots(hip_snapshot(env), oracle_snapshot(replay), f'{family}: after one-step rollout {t} with auto-reset')
    assert sum(int((a | b).sum()) for a, b in zip(want['term'], want['trunc'])) > B, 'the case must actually reset envs'
    G.assert_same(np_(env.seeds), want_seeds, 'seeds')
    np.testing.assert_allclose(metrics.cpu().numpy(), want['metrics'], rtol=1e-9)
    # the opening reset of a one-step rollout (fresh seeds), its tapes and its metrics
    m2 = torch.zeros_like(metrics)
    rec = env.rollout(1, policy_seed=5, reset_first=True, seed_increment=77, record=True, metrics=m2)
    new_seeds = (want_seeds.astype(np.int64) + 77).astype(np.int32)
    G.assert_same(np_(env.seeds), new_seeds, 'seeds after the folded-in reset')
    fresh = oracle.WildfireOracle(cfg)
    fresh.reset()
    acts = oracle.wildfire_random_policy(cfg, fresh.agent_task_count, fresh.env_task_count, new_seeds, 5, 0)
    fr, ar = oracle.wildfire_philox_randomness(cfg, new_seeds, fresh.num_moves)
    fresh.step(acts, fr, ar)
    G.assert_same(np_(rec['actions'])[0], acts, f'{family}: actions of the one-step rollout with the reset folded in')
    G.assert_same(np_(rec['rewards'])[0], fresh.rewards, f'{family}: rewards of the one-step rollout with the reset folded in')
    compare_snapshots(hip_snapshot(env), oracle_snapshot(fresh), f'{family}: one-step rollout with the reset folded in')
    want_m = np.concatenate([fresh.cumulative_rewards.astype(np.float64).sum(axis=1), [float(fresh.num_moves.sum())],
                             [float((fresh.terminations[0].astype(bool) | fresh.truncations[0].astype(bool)).sum())]])
    np.testing.assert_allclose(m2.cpu().numpy(), want_m, rtol=1e-12)
    env.check()


def test_one_step_graph_capture_keeps_the_reset_and_the_metrics(oracle):
    """capture_random_rollout with steps % episode_length == 1: the trailing one-step episode must still reset and count."""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    B, horizon = 2048, 5
    cfg = to_cstruct(configs.wildfire_openness(), B, horizon, track_cumulative_rewards=True)
    env = make_env(configs.wildfire_openness, B, horizon, rng='philox', track_cumulative_rewards=True)
    env.set_exclusive_device(True)
    seeds = torch.arange(B, dtype=torch.int32)
    env.reset(seed=seeds)
    metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device='cuda')
    graph = env.capture_random_rollout(horizon + 1, policy_seed=12, include_reset=True, episode_length=horizon, seed_stride=9, metrics=metrics)
    graph.replay()
    torch.cuda.synchronize()
    env.refresh()  # (a replay moves the env without going through step(): publish what the buffers hold now)
    s1 = (seeds.numpy().astype(np.int64) + 9).astype(np.int32)
    first = oracle.WildfireOracle(cfg)
    first.reset()
    first.rollout(s1, 12, 0, horizon)
    s2 = (s1.astype(np.int64) + 9).astype(np.int32)
    last = oracle.WildfireOracle(cfg)
    last.reset()
    last.rollout(s2, 12, 0, 1)
    compare_snapshots(hip_snapshot(env), oracle_snapshot(last), 'after the trailing one-step episode of the graph')
    want = np.zeros(len(env.agents) + 2)
    for o in (first, last):
        want += np.concatenate([o.cumulative_rewards.astype(np.float64).sum(axis=1), [float(o.num_moves.sum())],
                                [float((o.terminations[0].astype(bool) | o.truncations[0].astype(bool)).sum())]])
    np.testing.assert_allclose(metrics.cpu().numpy(), want, rtol=1e-12)


@pytest.mark.parametrize('case', [dict(B=65536, max_steps=50, steps=50), dict(B=1000, max_steps=30, steps=34)], ids=['cfg4_B65536', 'ragged_past_the_horizon'])
def test_cybersecurity_tapes_in_the_multi_step_launch_against_the_oracle(oracle, case):
    """Observation and state tapes written INSIDE the one multi-step launch (every step but the last writes only its tape copy of the
    observation rows; a launch that finds the batch finished writes the last executed step's rows once more into the env's own buffers):
    every step's state and every agent's self / others / tasks rows against the oracle's loop, then the env's own outputs."""
    import test_hip_cybersecurity as C
    from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct
    B, steps = case['B'], case['steps']
    cfg = to_cstruct(configs.cyber_openness(), B, case['max_steps'], **configs.CYBER_DEFAULT_FLAGS)
    env = C.make_env(configs.cyber_openness, B, case['max_steps'], rng='philox')
    env.set_exclusive_device(True)
    assert env._lib.frz_cybersecurity_rollout_launches(env._handle, steps, _capi.FRZ_RNG_PHILOX) == 1
    seeds = torch.arange(B, dtype=torch.int32) * 3 + 1
    env.reset(seed=seeds)
    rec = env.rollout(steps, policy_seed=13, record=True, record_observations='full', record_state=True)
    o = oracle.CybersecurityOracle(cfg)
    o.reset()
    executed = 0
    for t in range(steps):
        if bool(o.truncations[0].all()):
            break
        acts = oracle.cybersecurity_random_policy(cfg, o.agent_task_count, o.location, seeds.numpy(), 13, t)
        nr, ar = oracle.cybersecurity_philox_randomness(cfg, seeds.numpy(), o.num_moves)
        o.step(acts, nr, ar)
        executed = t + 1
        if B > 5000 and t % 7 and t != steps - 1:
            continue
        want = C.oracle_snapshot(o)
        state = env.recorded_state(rec, t)
        G.assert_same(np_(state.network_state), want['network_state'], f'step {t} state tape: network_state')
        G.assert_same(np_(state.location), want['location'], f'step {t} state tape: location')
        G.assert_same(np_(state.presence).astype(bool), np.asarray(want['presence']).astype(bool), f'step {t} state tape: presence')
        obs = env.recorded_observations(rec, t)
        for a, agent_name in enumerate(env.agents):
            for part in ('self', 'others', 'tasks'):
                G.assert_same(np_(obs[agent_name][part]), want[f'obs_{part}_{a}'], f'step {t} observation tape: {part}[{a}]')
    assert executed == min(steps, case['max_steps'])
    if executed < steps:  # what the frozen steps leave (rewards summed once per agent) — and the env's own observation rows, rewritten on the way out
        A = len(env.agents)
        o.step(np.zeros((A, B, 2), np.int32), np.zeros((1, B, cfg.num_nodes), np.float32), np.zeros((1, B, A), np.float32))
    C.compare_snapshots(C.hip_snapshot(env), C.oracle_snapshot(o), f'the env after {steps} steps in one launch with tapes')
    env.check()


def test_reset_finished_says_so_where_the_library_has_no_device_side_partial_reset():
    from free_range_zoo_amd.envs import rideshare_v0
    env = rideshare_v0.parallel_env(configuration=configs.rideshare_busy(A=4, steps=10, per_step=2, seed=2), parallel_envs=64, max_steps=10,
                                    device=torch.device('cuda'))
    env.reset(seed=torch.arange(64, dtype=torch.int32))
    with pytest.raises(NotImplementedError, match='reset_batches'):
        env.reset_finished()


@pytest.mark.parametrize('custom_initial_state', [False, True], ids=['configured_state', 'saved_state'])
@pytest.mark.parametrize('rng', ['philox', 'mt19937'])
def test_cybersecurity_masked_reset_equals_reset_batches(rng, custom_initial_state):
    """`reset_finished(mask)` for cybersecurity (frz_cybersecurity_reset_masked: reset_batches with the selection on the device) against the
    host-index `reset_batches` (itself pinned by tests/golden/partial_cybersecurity.npz) on a twin env: an arbitrary mask mid-episode, then
    the finished envs (mask None) at the horizon — also after `reset(options={'initial_state': ...})`, where both must restore that state."""
    import test_hip_cybersecurity as C
    B, horizon = 900, 7
    a, b = [C.make_env(configs.cyber_openness, B, horizon, rng=rng) for _ in range(2)]
    seeds = torch.arange(B, dtype=torch.int32) + 9
    for env in (a, b):
        env.reset(seed=seeds)
    if custom_initial_state:
        a.rollout(3, policy_seed=1)
        custom = a.state().clone()
        for env in (a, b):
            env.reset(seed=seeds, options={'initial_state': custom})
    picks = torch.zeros(B, dtype=torch.bool)
    picks[torch.arange(0, B, 7)] = True
    for t in range(2 * horizon + 1):
        a.step_random_policy(5, t), b.step_random_policy(5, t)
        if t == 2:  # an arbitrary selection, mid-episode
            a.reset_finished(picks.cuda(), seed_increment=21)
            idx = picks.nonzero().reshape(-1).cuda()
            b.reset_batches(idx, seed=(b.seeds[idx].to(torch.int64) + 21).to(torch.int32))
        elif bool(b.finished.any()):  # the envs that reached the horizon (those reset at t == 2 reach it later)
            a.reset_finished(seed_increment=4)
            idx = b.finished.nonzero().reshape(-1)
            b.reset_batches(idx, seed=(b.seeds[idx].to(torch.int64) + 4).to(torch.int32))
        C.compare_snapshots(C.hip_snapshot(a), C.hip_snapshot(b), f'{rng} step {t}')
        G.assert_same(np_(a.seeds), np_(b.seeds), f'seeds at step {t}')
        G.assert_same(np_(a._actions), np_(b._actions), f'staged actions at step {t}')
    assert int(np_(a.num_moves).min()) < horizon
    a.check(), b.check()


def test_cybersecurity_one_step_rollout_honours_the_spec(oracle):
    import test_hip_cybersecurity as C
    from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct
    B = 1800
    cfg = to_cstruct(configs.cyber_openness(), B, 50, **configs.CYBER_DEFAULT_FLAGS)
    env = C.make_env(configs.cyber_openness, B, 50, rng='philox')
    env.set_exclusive_device(True)
    seeds = torch.arange(B, dtype=torch.int32) + 11
    env.reset(seed=seeds)
    env.rollout(6, policy_seed=1)
    rec = env.rollout(1, policy_seed=4, reset_first=True, record=True)
    o = oracle.CybersecurityOracle(cfg)
    o.reset()
    acts = oracle.cybersecurity_random_policy(cfg, o.agent_task_count, o.location, seeds.numpy(), 4, 0)
    nr, ar = oracle.cybersecurity_philox_randomness(cfg, seeds.numpy(), o.num_moves)
    o.step(acts, nr, ar)
    G.assert_same(np_(rec['actions'])[0], acts, 'actions of the one-step rollout')
    G.assert_same(np_(rec['rewards'])[0], o.rewards, 'rewards of the one-step rollout')
    G.assert_same(np_(rec['dones'])[0, 1].astype(bool), o.truncations[0].astype(bool), 'truncations of the one-step rollout')
    C.compare_snapshots(C.hip_snapshot(env), C.oracle_snapshot(o), 'one-step rollout with the reset folded in')
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 5. reset_finished(mask): reset_batches with the selection on the device, against the reference's recordings
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('label', ['wildfire', 'wildfire_bad_actions', 'wildfire_grid8x8'])
def test_masked_reset_matches_the_recorded_partial_reset(label):
    """tests/golden/partial_*.npz (reset_batches(batch_indices, seed) recorded from the reference, see test_hip_partial_resets.py) through
    the device-mask entry point: same state, bookkeeping, observations, lists and continuation — no indices on the host."""
    from test_hip_partial_resets import _replay
    data = np.load(G.golden_path(f'partial_{label}.npz'))
    build, kwargs = configs.WILDFIRE_GOLDEN[str(data['variant'])]
    cfg = G.load_cfg(data, _capi.frz_wildfire_cfg)
    B, A = cfg.parallel_envs, cfg.num_agents
    sizes = ((3, B, cfg.grid_height * cfg.grid_width), (5, B, A))
    names = ('field_randomness', 'agent_randomness')
    env = make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    _replay(env, data, 'a', 6, names, sizes, G.compare_wildfire, hip_snapshot, A, label)
    mask = torch.zeros(B, dtype=torch.bool)
    mask[torch.from_numpy(data['batch_indices']).long()] = True
    # the recorded call passes new seeds for the selected envs; the device entry adds an increment: set the seeds first, add zero
    env.generator.seed(torch.from_numpy(data['batch_seeds']), partial_seeding=torch.from_numpy(data['batch_indices']))
    env.reset_finished(mask.cuda(), seed_increment=0)
    G.compare_wildfire(hip_snapshot(env), data, 'p_', A, f'{label} after the masked reset')
    for key, mine in (('p_rewards', env.rewards), ('p_terminations', env.terminations), ('p_truncations', env.truncations)):
        got = np.stack([mine[agent].cpu().numpy() for agent in env.agents])
        G.assert_same(got.astype(data[key].dtype), data[key], f'{label} {key}')
    _replay(env, data, 'b', 5, names, sizes, G.compare_wildfire, hip_snapshot, A, label)
    env.check()


@pytest.mark.parametrize('family', ['roles', 'lane', 'grid'])
@pytest.mark.parametrize('via', ['reset_finished', 'auto_reset_rollout'])
def test_device_side_partial_resets_restore_the_saved_state(oracle, family, via, monkeypatch):
    """After reset(options={'initial_state': ...}) the reference's reset_batches restores THAT state (`self._state.restore_initial`,
    wildfire.py:391), not the configured one: the device-mask reset and the auto-reset rollouts must too.  Checked
    against the host-index path (`reset_batches`, itself pinned by tests/golden/partial_wildfire_grid8x8.npz) on a twin env."""
    monkeypatch.setenv('FRZ_WF_KERNEL', family)
    B, horizon = 700, 6
    envs = [make_env(configs.wildfire_openness, B, horizon, rng='philox') for _ in range(2)]
    seeds = torch.arange(B, dtype=torch.int32) + 3
    for env in envs:
        env.set_exclusive_device(True)
        env.reset(seed=seeds)
    envs[0].rollout(3, policy_seed=2)  # a state that is not the configured one: three steps in, per env different
    custom = envs[0].state().clone()
    for env in envs:
        env.reset(seed=seeds, options={'initial_state': custom})
    a, b = envs
    for t in range(10):
        if via == 'reset_finished':
            a.rollout(1, policy_seed=6, first_step=t)
            a.reset_finished(seed_increment=13)
        else:
            a.rollout(1, policy_seed=6, first_step=t, auto_reset=True, seed_stride=13)
        b.rollout(1, policy_seed=6, first_step=t)
        finished = b.finished.nonzero().reshape(-1)
        if finished.numel():
            b.reset_batches(finished, seed=(b.seeds[finished].to(torch.int64) + 13).to(torch.int32))
        compare_snapshots(hip_snapshot(a), hip_snapshot(b), f'{family}/{via}: step {t}')
        G.assert_same(np_(a.seeds), np_(b.seeds), 'seeds')
    # a multi-step auto-reset rollout with a saved state takes the per-step path and lands in the same place
    if via == 'auto_reset_rollout':
        a.rollout(7, policy_seed=6, first_step=10, auto_reset=True, seed_stride=13)
        for t in range(10, 17):
            b.rollout(1, policy_seed=6, first_step=t)
            finished = b.finished.nonzero().reshape(-1)
            if finished.numel():
                b.reset_batches(finished, seed=(b.seeds[finished].to(torch.int64) + 13).to(torch.int32))
        compare_snapshots(hip_snapshot(a), hip_snapshot(b), f'{family}: seven more steps in one call')
    assert int(np_(a.num_moves).min()) < 3, 'the case must actually restart envs'
    for env in envs:
        env.check()


def test_multi_step_launch_where_some_chunks_cannot_answer_for_the_batch(oracle):
    """Between the steps of a multi-step launch a workgroup whose OWN 256 envs already show that every batch total a step asks about is
    non-zero (an attackable task for every agent, an env that is not terminated, one that is not truncated) does not wait for the
    totals (round 4); a workgroup whose own sums leave a question open still does.  A batch built so that both kinds sit in one launch:
    chunk 0 without any fire (no tasks at all: its sums are zero in every list channel), chunk 1 with agent 0 out of suppressant (that
    agent's channel zero there), the other chunks ordinary — against the same rollout run one launch per step on a twin env."""
    B, horizon, steps = 1100, 40, 14
    envs = [make_env(configs.wildfire_openness, B, horizon, rng='philox') for _ in range(2)]
    seeds = torch.arange(B, dtype=torch.int32) * 5 + 1
    for env in envs:
        env.reset(seed=seeds)
    custom = envs[0].state().clone()
    custom.fires[:256] = -custom.fires[:256].abs()   # unlit everywhere: nothing to fight in chunk 0
    custom.intensity[:256] = 0
    custom.suppressants[256:512, 0] = 0.0            # agent 0 cannot fight anywhere in chunk 1
    for env in envs:
        env.reset(seed=seeds, options={'initial_state': custom})
    a, b = envs
    a.set_exclusive_device(True)
    assert a._lib.frz_wildfire_rollout_launches(a._handle, steps, _capi.FRZ_RNG_PHILOX) == 1
    ra = a.rollout(steps, policy_seed=9, first_step=0, record=True)
    rb = b.rollout(steps, policy_seed=9, first_step=0, record=True)  # (not declared exclusive: one launch per step)
    for key in ('rewards', 'dones', 'actions'):
        G.assert_same(np_(ra[key]), np_(rb[key]), f'recorded {key}')
    compare_snapshots(hip_snapshot(a), hip_snapshot(b), 'one launch against one launch per step')
    for env in envs:
        env.check()


@pytest.mark.parametrize('family', ['roles', 'lane', 'grid'])
def test_masked_reset_of_the_finished_envs_in_every_kernel_family(oracle, family, monkeypatch):
    """reset_finished() (mask=None: the finished envs, decided on the device) against the oracle's reset_batches in each kernel family —
    the grid family keeps its cells env-major."""
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    monkeypatch.setenv('FRZ_WF_KERNEL', family)
    B = 1234
    cfg = to_cstruct(configs.wildfire_openness(), B, 9)
    env = make_env(configs.wildfire_openness, B, 9, rng='philox')
    seeds = torch.arange(B, dtype=torch.int32) + 2
    env.reset(seed=seeds)
    o = oracle.WildfireOracle(cfg)
    o.reset()
    s = seeds.numpy().copy()
    for t in range(14):
        acts = oracle.wildfire_random_policy(cfg, o.agent_task_count, o.env_task_count, s, 4, t)
        fr, ar = oracle.wildfire_philox_randomness(cfg, s, o.num_moves)
        env.step(torch.from_numpy(acts).cuda())
        o.step(acts, fr, ar)
        if t in (5, 8, 11):
            env.reset_finished(seed_increment=31)
            o.reset_masked(None, s, 31)
            compare_snapshots(hip_snapshot(env), oracle_snapshot(o), f'{family}: after the masked reset at step {t}')
            G.assert_same(np_(env.seeds), s, 'seeds')
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), f'{family}: end')
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 6. cybersecurity: the multi-step launch against the oracle and the reference's recorded trajectories
# ------------------------------------------------------------------------------------------------------------
def cyber_list_views(env, block: np.ndarray):
    A, B, N = len(env.agents), env.parallel_envs, env._N
    base = env._arena.data_ptr() + env._list_block_offset
    values = block[env._bufs.act_map_values - base:][:A * B * N * 4].view(np.int32).reshape(A, B * N)
    offsets = block[env._bufs.act_map_offsets - base:][:A * (B + 1) * 8].view(np.int64).reshape(A, B + 1)
    return {f'act_map_offsets_{a}': offsets[a] for a in range(A)} | {f'act_map_values_{a}': values[a, :int(offsets[a, -1])] for a in range(A)}


@pytest.mark.parametrize('case', [dict(B=65536, max_steps=50, steps=50, kwargs={}),                       # cfg4 as bench.py's secondary workload runs it
                                  dict(B=1000, max_steps=30, steps=34, kwargs={}),                          # ragged, past the horizon
                                  dict(B=2500, max_steps=40, steps=21, kwargs=dict(show_bad_actions=False, observe_other_presence=True)),
                                  # 9-16 nodes (round 4: the state / view kernel reaches 16 nodes with up to 8 agents; its multi-step instantiation there
                                  # spills and is slower than one launch per step, so `rollout` runs these one launch per step — same results)
                                  dict(B=1300, max_steps=25, steps=20, kwargs={}, build=lambda: configs.cyber_grid(12, 3, 3)),
                                  dict(B=700, max_steps=12, steps=15, kwargs=dict(partially_observable=True), build=lambda: configs.cyber_grid(16, 4, 4, seed=5))],
                         ids=['cfg4_B65536', 'ragged_past_the_horizon', 'no_bad_actions', '12_nodes', '16_nodes_past_the_horizon'])
def test_cybersecurity_multi_step_launch_against_the_oracle(oracle, case):
    """frz_cybersecurity_rollout as ONE launch vs the oracle's steps (frz_oracle_cybersecurity_rollout's loop, unrolled here to look at every
    step): sampled actions, rewards, truncations and action mappings of every step, then the whole final state."""
    import test_hip_cybersecurity as C
    from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct
    B, steps, kwargs = case['B'], case['steps'], case['kwargs']
    flags = dict(configs.CYBER_DEFAULT_FLAGS)
    flags.update(kwargs)
    build = case.get('build', configs.cyber_openness)
    cfg = to_cstruct(build(), B, case['max_steps'], **flags)
    env = C.make_env(build, B, case['max_steps'], rng='philox', **kwargs)
    env.set_exclusive_device(True)
    assert env._lib.frz_cybersecurity_rollout_launches(env._handle, steps, _capi.FRZ_RNG_PHILOX) == (1 if env._N <= 8 else steps)
    seeds = torch.arange(B, dtype=torch.int32) * 3 + 2
    env.reset(seed=seeds)
    rec = env.rollout(steps, policy_seed=31, record=True)
    prepare(env, rec)
    rewards, dones, actions, lists = (np_(rec[k]) for k in ('rewards', 'dones', 'actions', 'lists'))
    o = oracle.CybersecurityOracle(cfg)
    o.reset()
    executed = 0
    for t in range(steps):
        if bool(o.truncations[0].all()):
            break
        acts = oracle.cybersecurity_random_policy(cfg, o.agent_task_count, o.location, seeds.numpy(), 31, t)
        nr, ar = oracle.cybersecurity_philox_randomness(cfg, seeds.numpy(), o.num_moves)
        o.step(acts, nr, ar)
        executed = t + 1
        G.assert_same(actions[t], acts, f'step {t} sampled actions')
        G.assert_same(rewards[t], o.rewards, f'step {t} rewards')
        G.assert_same(dones[t, 1].astype(bool), o.truncations[0].astype(bool), f'step {t} truncations')
        assert not dones[t, 0].any()
        if t < steps - 1 and not bool(o.truncations[0].all()):
            got = cyber_list_views(env, lists[t])
            for a in range(len(env.agents)):
                v, off = o.action_map(a)
                G.assert_same(got[f'act_map_offsets_{a}'], off, f'step {t} (list record) offsets of agent {a}')
                G.assert_same(got[f'act_map_values_{a}'], v, f'step {t} (list record) values of agent {a}')
    assert executed == min(steps, case['max_steps'])
    if executed < steps:
        A = len(env.agents)
        o.step(np.zeros((A, B, 2), np.int32), np.zeros((1, B, cfg.num_nodes), np.float32), np.zeros((1, B, A), np.float32))
    C.compare_snapshots(C.hip_snapshot(env), C.oracle_snapshot(o), f'after {steps} steps in one launch')
    # the C twin of this loop (what a CPU baseline would time) ends in the same state
    twin = oracle.CybersecurityOracle(cfg)
    twin.reset()
    twin.rollout(seeds.numpy(), 31, 0, steps)
    assert np.array_equal(twin.network_state, o.network_state) and np.array_equal(twin.rewards, o.rewards)
    env.check()


@pytest.mark.parametrize('name', sorted(configs.CYBER_GOLDEN))
def test_cybersecurity_golden_trajectory_through_the_multi_step_launch(name):
    """The recorded trajectories of the unmodified reference through `rollout(actions=tape, randomness=tapes)`: the whole trajectory as one
    launch (reward / truncation tapes compared step by step) and prefixes of 1 .. T steps against the reference's snapshot after each."""
    import test_hip_cybersecurity as C
    from test_oracle_cybersecurity import compare_cyber
    build, kwargs = configs.CYBER_GOLDEN[name]
    data = np.load(G.golden_path(f'traj_cybersecurity_{name}.npz'))
    cfg = G.load_cfg(data, _capi.frz_cybersecurity_cfg)
    B, N, A, T = cfg.parallel_envs, cfg.num_nodes, cfg.num_attackers + cfg.num_defenders, int(data['steps'])
    env = C.make_env(build, B, None if cfg.max_steps < 0 else cfg.max_steps, **kwargs)
    env.set_exclusive_device(True)
    assert env._lib.frz_cybersecurity_rollout_launches(env._handle, T, _capi.FRZ_RNG_INJECTED) == (1 if max(N, A) <= 8 else T)
    acts = np.zeros((T, A, B, 2), np.int32)
    net, agent = np.zeros((T, B, N), np.float32), np.zeros((T, B, A), np.float32)
    for t in range(T):
        acts[t] = data[f's{t}_actions']
        if bool(data[f's{t}_stepped']):
            net[t], agent[t] = data[f's{t}_network_randomness'].reshape(B, N), data[f's{t}_agent_randomness'].reshape(B, A)
    acts, net, agent = torch.from_numpy(acts).cuda(), torch.from_numpy(net).cuda(), torch.from_numpy(agent).cuda()
    seeds = torch.arange(B, dtype=torch.int32)
    env.reset(seed=seeds)
    rec = env.rollout(T, actions=acts, randomness=(net, agent), record=True)
    rewards, dones = np_(rec['rewards']), np_(rec['dones'])
    for t in range(T):
        if not bool(data[f's{t}_stepped']):
            break
        G.assert_same(rewards[t], data[f's{t}_rewards'], f'{name} step {t} reward tape', G.REWARD_RTOL)
        G.assert_same(dones[t, 1].astype(bool), data[f's{t}_truncations'][0], f'{name} step {t} truncation tape')
    compare_cyber(C.hip_snapshot(env), data, f's{T - 1}_', A, f'{name} after the whole trajectory in one launch')
    for n in range(1, T + 1):
        env.reset(seed=seeds)
        env.rollout(n, actions=acts[:n].contiguous(), randomness=(net[:n].contiguous(), agent[:n].contiguous()))
        compare_cyber(C.hip_snapshot(env), data, f's{n - 1}_', A, f'{name}: {n} steps in one launch')
    env.check()


def test_cybersecurity_rollout_with_the_reset_folded_in(oracle):
    import test_hip_cybersecurity as C
    from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct
    B = 3000
    cfg = to_cstruct(configs.cyber_openness(), B, 50, **configs.CYBER_DEFAULT_FLAGS)
    env = C.make_env(configs.cyber_openness, B, 50, rng='philox')
    env.set_exclusive_device(True)
    seeds = torch.arange(B, dtype=torch.int32) + 7
    env.reset(seed=seeds)
    env.rollout(9, policy_seed=1)
    env.rollout(17, policy_seed=2, reset_first=True)
    o = oracle.CybersecurityOracle(cfg)
    o.reset()
    o.rollout(seeds.numpy(), 2, 0, 17)
    C.compare_snapshots(C.hip_snapshot(env), C.oracle_snapshot(o), 'reset folded into the launch')
    env.check()


# ------------------------------------------------------------------------------------------------------------
# 7. the residency guard of set_exclusive_device
# ------------------------------------------------------------------------------------------------------------
def test_exclusive_device_is_refused_when_the_launch_would_not_fit(oracle, monkeypatch):
    """The workgroups of a multi-step launch wait for each other inside the kernel, so all of them must be resident: the library checks its
    own part (occupancy x compute units of the arena's device >= chunks, no CU mask) and refuses otherwise — rollouts then take one launch per
    step, with the same results.  FRZ_ASSUME_COMPUTE_UNITS stands in for a smaller / partitioned device."""
    import test_hip_cybersecurity as C
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    B = 8192  # 32 chunks
    env = make_env(configs.wildfire_openness, B, 30, rng='philox')
    cy = C.make_env(configs.cyber_openness, B, 30, rng='philox')
    monkeypatch.setenv('FRZ_ASSUME_COMPUTE_UNITS', '16')
    assert env.set_exclusive_device(True) is False and cy.set_exclusive_device(True) is False
    assert env._lib.frz_wildfire_rollout_launches(env._handle, 10, _capi.FRZ_RNG_PHILOX) == 10
    assert cy._lib.frz_cybersecurity_rollout_launches(cy._handle, 10, _capi.FRZ_RNG_PHILOX) == 10
    monkeypatch.setenv('ROC_GLOBAL_CU_MASK', '0xffff')
    monkeypatch.setenv('FRZ_ASSUME_COMPUTE_UNITS', '256')
    assert env.set_exclusive_device(True) is False  # a CU mask: the reported properties cannot be trusted
    monkeypatch.delenv('ROC_GLOBAL_CU_MASK')
    seeds = torch.arange(B, dtype=torch.int32)
    env.reset(seed=seeds)
    monkeypatch.setenv('FRZ_ASSUME_COMPUTE_UNITS', '16')
    assert env.set_exclusive_device(True) is False
    env.rollout(12, policy_seed=3)  # launch per step
    cfg = to_cstruct(configs.wildfire_openness(), B, 30)
    o = oracle.WildfireOracle(cfg)
    o.reset()
    o.rollout(seeds.numpy(), 3, 0, 12)
    compare_snapshots(hip_snapshot(env), oracle_snapshot(o), 'refused: one launch per step')
    monkeypatch.setenv('FRZ_ASSUME_COMPUTE_UNITS', '64')
    assert env.set_exclusive_device(True) is True and env._lib.frz_wildfire_rollout_launches(env._handle, 10, _capi.FRZ_RNG_PHILOX) == 1


@pytest.mark.gpu
def test_rideshare_rollout_equals_the_step_loop():
    """frz_rideshare_rollout (one launch sequence per step): an action tape, then the in-launch policy with the reset folded in, against the
    same steps taken one `step()` at a time; every record (rewards, flags, actions, the packed-list block) is compared."""
    from free_range_zoo_amd.envs import rideshare_v0
    B, steps = 900, 9
    build = lambda: configs.rideshare_busy(A=4, steps=20, per_step=2, seed=4)
    make = lambda: rideshare_v0.parallel_env(configuration=build(), parallel_envs=B, max_steps=12, device=torch.device('cuda'))
    loop, roll = make(), make()
    for env in (loop, roll):
        env.reset(seed=torch.arange(B, dtype=torch.int32))
    rewards, dones, actions, lists = [], [], [], []
    block, nbytes = ctypes.c_void_p(), ctypes.c_int64()
    _capi.check(loop._lib.frz_rideshare_list_block(loop._handle, ctypes.byref(block), ctypes.byref(nbytes)), 'list_block')
    offset = block.value - loop._arena.data_ptr()
    for t in range(steps):
        loop.step_random_policy(policy_seed=21, policy_step=t)
        rewards.append(loop._rewards.clone()), actions.append(loop._actions.clone())
        dones.append(torch.stack([loop._terminations[0], loop._truncations[0]]).to(torch.uint8))
        lists.append(loop._arena[offset:offset + nbytes.value].clone())
    tape = torch.stack(actions).contiguous()
    out = roll.rollout(steps, actions=tape, record=True)
    assert torch.equal(out['rewards'], torch.stack(rewards)) and torch.equal(out['dones'], torch.stack(dones))
    assert out['list_block_offset'] == offset
    for t in range(steps - 1):
        assert torch.equal(out['lists'][t], lists[t]), f'lists after step {t}'
    assert torch.equal(roll._arena[offset:offset + nbytes.value], lists[-1])
    assert torch.equal(roll._passengers, loop._passengers) and torch.equal(roll._agents, loop._agents)
    # the in-launch policy with the reset folded in = reset + the same policy steps
    loop.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(steps):
        loop.step_random_policy(policy_seed=21, policy_step=t)
    out = roll.rollout(steps, policy_seed=21, first_step=0, reset_first=True, record=True)
    assert torch.equal(out['actions'], tape) and torch.equal(out['rewards'], torch.stack(rewards))
    assert torch.equal(roll._passengers, loop._passengers) and torch.equal(roll._rewards, loop._rewards)
    roll.check()
    with pytest.raises(Exception):
        roll.rollout(2, auto_reset=True)  # no partial reset in this domain (rideshare.py:246)
