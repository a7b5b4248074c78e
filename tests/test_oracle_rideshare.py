"""CPU: pin the oracle's rideshare restatement against the reference's golden vectors."""
import ctypes

import numpy as np
import pytest

import golden_util as G
from free_range_zoo_amd import _capi


def oracle_snapshot(o):
    cfg = o.cfg
    A, B = cfg.num_agents, cfg.parallel_envs
    n = int(o.task_offsets[-1])
    snap = {'agents': o.agents, 'passengers': o.table(), 'num_moves': o.num_moves, 'env_task_count': o.env_task_count,
            'agent_task_count': o.agent_task_count, 'task_values': o.task_values[:n], 'task_offsets': o.task_offsets,
            'rewards': o.rewards, 'terminations': o.terminations, 'truncations': o.truncations}
    for a in range(A):
        tasks, mapping, off = o.agent_tasks(a)
        snap[f'act_map_values_{a}'], snap[f'act_map_offsets_{a}'] = mapping, off
        snap[f'obs_map_values_{a}'], snap[f'obs_map_offsets_{a}'] = mapping, off
        snap[f'obs_tasks_values_{a}'], snap[f'obs_tasks_offsets_{a}'] = tasks, off
        snap[f'obs_self_{a}'], snap[f'obs_others_{a}'] = o.obs_self[a], o.obs_others[a]
        snap[f'cumulative_rewards_{a}'] = o.cumulative_rewards[a]
    return snap


def compare_rideshare(snap, data, prefix, A, what):
    for name in ('agents', 'passengers', 'num_moves', 'env_task_count', 'agent_task_count', 'task_values', 'task_offsets'):
        want = data[prefix + name]
        G.assert_same(snap[name], want, f'{what} {name}')
        assert np.asarray(snap[name]).dtype == want.dtype, f'{what} {name}: dtype {np.asarray(snap[name]).dtype} != {want.dtype}'
    for a in range(A):
        for name in ('act_map_values', 'act_map_offsets', 'obs_map_values', 'obs_map_offsets', 'obs_self', 'obs_others', 'obs_tasks_values',
                     'obs_tasks_offsets'):
            got, want = snap[f'{name}_{a}'], data[f'{prefix}{name}_{a}']
            G.assert_same(got, want, f'{what} {name}[{a}]')
            assert np.asarray(got).dtype == want.dtype, f'{what} {name}[{a}] dtype {np.asarray(got).dtype} != {want.dtype}'
        G.assert_same(snap[f'cumulative_rewards_{a}'], data[f'{prefix}cumulative_rewards_{a}'], f'{what} cumulative[{a}]', G.REWARD_RTOL)
    if prefix != 'r_':
        G.assert_same(snap['rewards'], data[prefix + 'rewards'], f'{what} rewards', G.REWARD_RTOL)
        G.assert_same(snap['terminations'].astype(bool), data[prefix + 'terminations'], f'{what} terminations')
        G.assert_same(snap['truncations'].astype(bool), data[prefix + 'truncations'], f'{what} truncations')


@pytest.mark.parametrize('name', G.trajectories('rideshare'))
def test_trajectory_matches_reference(oracle, name):
    data = np.load(G.golden_path(name))
    cfg = G.load_cfg(data, _capi.frz_rideshare_cfg)
    o = oracle.RideshareOracle(cfg, data['schedule'])
    o.reset()
    A = cfg.num_agents
    compare_rideshare(oracle_snapshot(o), data, 'r_', A, f'{name} reset')
    for t in range(int(data['steps'])):
        p = f's{t}_'
        o.step(data[p + 'actions'])
        compare_rideshare(oracle_snapshot(o), data, p, A, f'{name} step {t}')
        finished = o.terminations.all(axis=0) | o.truncations.all(axis=0)
        G.assert_same(finished, data[p + 'finished'], f'{name} step {t} finished')
    assert int(o.error_flags[0]) == 0


def test_known_answer_movement(oracle):
    """MovementTransition.forward calls of the reference's test_movement.py: best move + distance cost per agent."""
    lib = oracle.lib()
    seen = 0
    for case in G.known_answers('rideshare'):
        if case['cls'] != 'MovementTransition':
            continue
        seen += 1
        cfg = _capi.frz_rideshare_cfg()
        cfg.use_fast_travel = int(case['attr_fast_travel'])
        cfg.use_diagonal_travel = int(case['buf_directions'].shape[0] == 9)
        vectors, agents_in, agents_out, dist = case['arg_vectors'], case['in_agents'], case['out_agents'], case['ret_0']
        B, A = vectors.shape[:2]
        for b in range(B):
            for a in range(A):
                vec = (ctypes.c_int32 * 4)(*[int(v) for v in vectors[b, a]])
                move = (ctypes.c_int32 * 2)()
                cost = ctypes.c_float()
                lib.frz_oracle_rs_move(ctypes.byref(cfg), vec, move, ctypes.byref(cost))
                what = f"{case['test']} b={b} a={a}"
                assert [agents_in[b, a, 0] + move[0], agents_in[b, a, 1] + move[1]] == agents_out[b, a].tolist(), what
                assert np.float32(cost.value) == dist[b, a], what
    assert seen >= 3


# ------------------------------------------------------------------------------------------------------------
# the other three transitions of the reference's own transition tests (tests/golden/ka_rideshare.npz), one call each
# ------------------------------------------------------------------------------------------------------------
def _oracle_from_table(oracle, case, schedule=None):
    """An oracle whose per-env slot tables hold the reference's global passenger table of a recorded call."""
    agents = np.asarray(case['in_agents'], np.int32)
    B, A = agents.shape[:2]
    table = np.asarray(case.get('in_passengers', np.zeros((0, 11))), np.int64).reshape(-1, 11)
    schedule = np.zeros((0, 7), np.int32) if schedule is None else np.asarray(schedule, np.int32).reshape(-1, 7)
    cfg = _capi.frz_rideshare_cfg()
    cfg.parallel_envs, cfg.num_agents, cfg.max_passengers, cfg.schedule_rows = B, A, 32, schedule.shape[0]
    o = oracle.RideshareOracle(cfg, schedule if schedule.shape[0] else np.zeros((1, 7), np.int32))
    o.agents[:] = agents
    slot_of_row = np.full(table.shape[0], -100, np.int32)
    for r, row in enumerate(table):
        b = int(row[0])
        slot_of_row[r] = o.passenger_count[b]
        o.passengers[b, o.passenger_count[b]] = row[1:]
        o.passenger_count[b] += 1
    return o, cfg, slot_of_row


def _slots(targets, slot_of_row):
    targets = np.asarray(targets, np.int64)
    return np.where(targets == -100, -100, slot_of_row[np.clip(targets, 0, max(len(slot_of_row) - 1, 0))] if len(slot_of_row) else -100).astype(np.int32)


def _cases(cls):
    cases = [c for c in G.known_answers('rideshare') if c['cls'] == cls]
    assert cases, cls
    return cases


def test_known_answer_passenger_entry(oracle):
    """PassengerEntryTransition.forward calls of the reference's test_passenger_entry.py."""
    lib = oracle.lib()
    for case in _cases('PassengerEntryTransition'):
        o, cfg, _ = _oracle_from_table(oracle, case, case['buf_schedule'])
        timesteps = np.ascontiguousarray(case['arg_timesteps'], np.int32)
        assert lib.frz_oracle_rs_passenger_entry(ctypes.byref(cfg), ctypes.byref(o.bufs), o.schedule.ctypes.data_as(ctypes.c_void_p),
                                                 timesteps.ctypes.data_as(ctypes.c_void_p)) == 0
        G.assert_same(o.table(), np.asarray(case['out_passengers'], np.int32).reshape(-1, 11), case['test'])
        assert int(o.error_flags[0]) == 0


def test_known_answer_passenger_state(oracle):
    """PassengerStateTransition.forward calls of the reference's test_passenger_state.py (accept conflicts, picks on the spot)."""
    lib = oracle.lib()
    for case in _cases('PassengerStateTransition'):
        o, cfg, slot_of_row = _oracle_from_table(oracle, case)
        accepts = np.ascontiguousarray(case['arg_accepts'], np.uint8)
        picks = np.ascontiguousarray(case['arg_picks'], np.uint8)
        targets = np.ascontiguousarray(_slots(case['arg_targets'], slot_of_row))
        vectors = np.ascontiguousarray(case['arg_vectors'], np.int32)
        timesteps = np.ascontiguousarray(case['arg_timesteps'], np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.frz_oracle_rs_passenger_state(ctypes.byref(cfg), ctypes.byref(o.bufs), p(accepts), p(picks), p(targets), p(vectors),
                                                 p(timesteps)) == 0
        G.assert_same(o.table(), np.asarray(case['out_passengers'], np.int32).reshape(-1, 11), case['test'])


def test_known_answer_passenger_exit(oracle):
    """PassengerExitTransition.forward calls of the reference's test_passenger_exit.py: rows removed in order, fares returned."""
    lib = oracle.lib()
    for case in _cases('PassengerExitTransition'):
        o, cfg, slot_of_row = _oracle_from_table(oracle, case)
        drops = np.ascontiguousarray(case['arg_drops'], np.uint8)
        targets = np.ascontiguousarray(_slots(case['arg_targets'], slot_of_row))
        vectors = np.ascontiguousarray(case['arg_vectors'], np.int32)
        fares = np.zeros(drops.shape, np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.frz_oracle_rs_passenger_exit(ctypes.byref(cfg), ctypes.byref(o.bufs), p(drops), p(targets), p(vectors), p(fares)) == 0
        G.assert_same(o.table(), np.asarray(case['out_passengers'], np.int32).reshape(-1, 11), case['test'])
        G.assert_same(fares, np.asarray(case['ret_0'], np.int32), case['test'] + ' fares')
