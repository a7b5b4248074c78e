"""CPU: pin the oracle's cybersecurity restatement against the reference's golden vectors."""
import ctypes

import numpy as np
import pytest

import golden_util as G
from free_range_zoo_amd import _capi


def oracle_snapshot(o):
    cfg = o.cfg
    B, N, Att, D = cfg.parallel_envs, cfg.num_nodes, cfg.num_attackers, cfg.num_defenders
    A = Att + D
    ka = int(cfg.observe_other_power) + int(cfg.observe_other_presence)
    kd = ka + int(cfg.observe_other_location)
    snap = {k: o.arrays[k] for k in ('network_state', 'location', 'num_moves', 'env_task_count', 'agent_task_count', 'rewards',
                                     'terminations', 'truncations')}
    snap['presence'] = o.presence.astype(bool)
    for a in range(A):
        snap[f'act_map_values_{a}'], snap[f'act_map_offsets_{a}'] = o.action_map(a)
        snap[f'obs_map_values_{a}'], snap[f'obs_map_offsets_{a}'] = o.obs_map_values, o.obs_map_offsets
        if a < Att:
            snap[f'obs_self_{a}'] = o.obs_self_attackers[a]
            snap[f'obs_others_{a}'] = o.obs_others_attackers[a].reshape(B, Att - 1, ka)
        else:
            snap[f'obs_self_{a}'] = o.obs_self_defenders[a - Att]
            snap[f'obs_others_{a}'] = o.obs_others_defenders[a - Att].reshape(B, D - 1, kd)
        snap[f'obs_tasks_{a}'] = o.obs_tasks[a]
        snap[f'cumulative_rewards_{a}'] = o.cumulative_rewards[a]
    return snap


def compare_cyber(snap, data, prefix, A, what):
    for name in ('network_state', 'location', 'presence', 'num_moves', 'env_task_count', 'agent_task_count'):
        G.assert_same(snap[name], data[prefix + name], f'{what} {name}')
        assert np.asarray(snap[name]).dtype == data[prefix + name].dtype, f'{what} {name} dtype'
    for a in range(A):
        for name in ('act_map_values', 'act_map_offsets', 'obs_map_values', 'obs_map_offsets', 'obs_self', 'obs_others', 'obs_tasks'):
            got, want = snap[f'{name}_{a}'], data[f'{prefix}{name}_{a}']
            G.assert_same(got, want, f'{what} {name}[{a}]')
            assert np.asarray(got).dtype == want.dtype, f'{what} {name}[{a}] dtype {np.asarray(got).dtype} != {want.dtype}'
        G.assert_same(snap[f'cumulative_rewards_{a}'], data[f'{prefix}cumulative_rewards_{a}'], f'{what} cumulative[{a}]', G.REWARD_RTOL)
    if prefix + 'rewards' in data.files:  # a snapshot taken after a reset has none
        G.assert_same(snap['rewards'], data[prefix + 'rewards'], f'{what} rewards', G.REWARD_RTOL)
        G.assert_same(snap['terminations'].astype(bool), data[prefix + 'terminations'], f'{what} terminations')
        G.assert_same(snap['truncations'].astype(bool), data[prefix + 'truncations'], f'{what} truncations')


@pytest.mark.parametrize('name', G.trajectories('cybersecurity'))
def test_trajectory_matches_reference(oracle, name):
    data = np.load(G.golden_path(name))
    cfg = G.load_cfg(data, _capi.frz_cybersecurity_cfg)
    o = oracle.CybersecurityOracle(cfg)
    o.reset()
    A = cfg.num_attackers + cfg.num_defenders
    compare_cyber(oracle_snapshot(o), data, 'r_', A, f'{name} reset')
    B, N = cfg.parallel_envs, cfg.num_nodes
    for t in range(int(data['steps'])):
        p = f's{t}_'
        if bool(data[p + 'stepped']):
            nr, ar = data[p + 'network_randomness'], data[p + 'agent_randomness']
        else:
            nr, ar = np.zeros((1, B, N), np.float32), np.zeros((1, B, A), np.float32)
        o.step(data[p + 'actions'], nr, ar)
        compare_cyber(oracle_snapshot(o), data, p, A, f'{name} step {t}')
        finished = o.terminations.all(axis=0) | o.truncations.all(axis=0)
        G.assert_same(finished, data[p + 'finished'], f'{name} step {t} finished')
    assert int(o.error_flags[0]) == 0


def test_known_answer_transitions(oracle):
    """Every forward() call of the reference's own cybersecurity transition tests (movement, presence, subnetwork)."""
    lib = oracle.lib()
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
    seen = set()
    for case in G.known_answers('cybersecurity'):
        cls = case['cls']
        seen.add(cls)
        what = f"{cls} / {case['test']}"
        net = np.ascontiguousarray(case['in_network_state'], np.int32)
        loc = np.ascontiguousarray(case['in_location'], np.int32)
        pres = np.ascontiguousarray(case['in_presence'], np.uint8)
        B = net.shape[0]
        cfg = _capi.frz_cybersecurity_cfg()
        cfg.parallel_envs, cfg.num_nodes, cfg.num_defenders = B, net.shape[1], loc.shape[1]
        cfg.num_attackers = pres.shape[1] - loc.shape[1]
        if cls == 'MovementTransition':
            targets = np.ascontiguousarray(case['arg_movement_targets'], np.int32)
            mask = np.ascontiguousarray(case['arg_movement_mask'], np.uint8)
            lib.frz_oracle_cy_movement(ptr(loc), ptr(targets), ptr(mask), ctypes.c_int64(loc.size))
        elif cls == 'PresenceTransition':
            assert cfg.num_attackers == int(case['buf_num_attackers'])
            persist, back = np.atleast_2d(case['buf_persist_probs']), np.atleast_2d(case['buf_return_probs'])
            assert (persist == persist[0]).all() and (back == back[0]).all()  # the reference's tests use per-env copies of one row
            for a in range(pres.shape[1]):
                cfg.persist_probs[a], cfg.return_probs[a] = float(persist[0, a]), float(back[0, a])
            r = np.ascontiguousarray(np.broadcast_to(case['arg_randomness_source'], pres.shape), np.float32)
            lib.frz_oracle_cy_presence(ctypes.byref(cfg), ptr(pres), ptr(loc), ptr(r), ctypes.c_int64(B))
        elif cls == 'SubnetworkTransition':
            cfg.temperature = float(case['buf_temperature'])
            cfg.stochastic_state = int(case['buf_stochastic_state'])
            cfg.num_states = int(case['buf_patched_states']) + int(case['buf_vulnerable_states']) + int(case['buf_exploited_states'])
            patches = np.ascontiguousarray(case['arg_patches'], np.float32)
            attacks = np.ascontiguousarray(case['arg_attacks'], np.float32)
            r = np.ascontiguousarray(np.broadcast_to(case['arg_randomness_source'], net.shape), np.float32)
            lib.frz_oracle_cy_subnetwork(ctypes.byref(cfg), ptr(net), ptr(patches), ptr(attacks), ptr(r), ctypes.c_int64(net.size))
        else:
            raise AssertionError(cls)
        G.assert_same(net, case['out_network_state'], what + ' network_state')
        G.assert_same(loc, case['out_location'], what + ' location')
        G.assert_same(pres.astype(bool), case['out_presence'], what + ' presence')
    assert seen == {'MovementTransition', 'PresenceTransition', 'SubnetworkTransition'}


# ------------------------------------------------------------------------------------------------------------------
# scripted baselines: the reference's stateful patched / exploited / camp agents along recorded trajectories
# ------------------------------------------------------------------------------------------------------------------
CYBER_BOTS = {'attacker': ('patched_attacker', 'exploited_attacker'), 'defender': ('patched_defender', 'exploited_defender', 'camp_defender')}


def _baseline_cases():
    data = np.load(G.golden_path('baselines_cybersecurity.npz'))
    for i in range(int(data['cases'])):
        p = f'c{i}_'
        yield i, {k[len(p):]: data[k] for k in data.files if k.startswith(p)}


def check_focus_policy(policy, case, kind, what):
    """`policy(kind, target, focused, actions, forced_pick)` updates the three state arrays in place (returns tie counts or None).
    Starting from the reference agent's recorded state and replaying its recorded torch.randint draws, state and answer after
    observe() are the reference's."""
    B = case['tasks'].shape[0]
    if kind == 'camp_defender':
        target, focused = np.zeros(B, np.int32), np.zeros(B, np.int32)
        actions = np.ascontiguousarray(case['camp_defender_pre'], np.int32).copy()
        policy(kind, target, focused, actions, None)
        assert np.array_equal(actions, case['camp_defender_post']), f'{what}: answers'
        return
    pre, post, draws = case[kind + '_pre'], case[kind + '_post'], case[kind + '_draws']
    target, focused = pre[0].copy(), pre[1].copy()
    actions = np.ascontiguousarray(pre[2:].T).copy()
    ties = policy(kind, target, focused, actions, draws[:, 1].copy() if len(draws) else None)
    if ties is not None and len(draws):
        assert np.array_equal(ties, draws[:, 0]), f'{what}: tied candidates'
    assert np.array_equal(actions, post[2:].T), f'{what}: answers'
    assert np.array_equal(target, post[0]) and np.array_equal(focused, post[1]), f'{what}: agent state'


def test_scripted_baselines_match_reference_answers(oracle):
    seen = dict(tied=0, early=0, patch=0, monitor=0, reset=0)
    for i, case in _baseline_cases():
        for kind in CYBER_BOTS[str(case['role'])]:
            def policy(kind, target, focused, actions, forced):
                return oracle.cyber_focus_policy(case['tasks'], case['obs_self'], kind, target, focused, actions,
                                                 subnetwork_states=int(case['subnetwork_states']), camp_target=int(case.get('camp_target', 0)),
                                                 mapping_numel=int(case['mapping_numel']), forced_pick=forced)

            check_focus_policy(policy, case, kind, f'case {i} {kind}')
            if kind != 'camp_defender':
                draws, pre, post = case[kind + '_draws'], case[kind + '_pre'], case[kind + '_post']
                seen['tied'] += int(len(draws) and (draws[:, 0] > 1).sum())
                seen['early'] += int(len(draws) == 0)
                seen['patch'] += int((post[3] == -2).sum())
                seen['monitor'] += int((post[3] == -3).sum())
                seen['reset'] += int(((pre[1] == 2) & (post[1] == 0)).sum())
    assert all(v > 0 for v in seen.values()), seen
