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


# ------------------------------------------------------------------------------------------------------------------
# scripted baselines: the reference's greedy / FIFO agents, answers and tie-break draws recorded
# ------------------------------------------------------------------------------------------------------------------
RIDESHARE_BOTS = ('greedy_focus', 'greedy_global', 'fifo_focus', 'fifo_global')


def _baseline_cases():
    data = np.load(G.golden_path('baselines_rideshare.npz'))
    for i in range(int(data['cases'])):
        p = f'c{i}_'
        case = {k[len(p):]: data[k] for k in data.files if k.startswith(p)}
        counts = case['task_counts']
        case['task_offsets'] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        yield i, case


def check_task_policy_answer(policy, case, kind, what):
    """`policy(forced_pick)` -> (actions, ties).  With the reference's recorded draw replayed, the answer is the reference's,
    tied or not; the number of tied candidates is the bound the reference passed to torch.randint."""
    got, ties = policy(case[kind + '_picks'])
    assert np.array_equal(ties, case[kind + '_ties']), f'{what}: tied candidates'
    assert np.array_equal(got, case[kind]), f'{what}: answers'


def test_scripted_baselines_match_reference_answers(oracle):
    seen = dict(tied=0, empty_env=0, empty_all=0, skipped=0, both_kinds=0)
    for i, case in _baseline_cases():
        for kind in RIDESHARE_BOTS:
            if not int(case[kind + '_valid']):
                seen['skipped'] += 1  # greedy_Tfocus asserted on this observation in the reference
                continue

            def policy(forced, kind=kind):
                return oracle.rideshare_task_policy(case['task_values'], case['task_offsets'], case['task_counts'], case['map_lengths'],
                                                    case['obs_self'], kind, bool(case['diagonal']), forced_pick=forced, return_ties=True)

            check_task_policy_answer(policy, case, kind, f'case {i} {kind}')
            seen['tied'] += int((case[kind + '_ties'] > 1).sum())
        seen['empty_all'] += int(case['task_counts'].sum() == 0)
        seen['empty_env'] += int((case['task_counts'] == 0).any() and case['task_counts'].sum() > 0)
        values, offsets = case['task_values'], case['task_offsets']
        for b in range(len(case['task_counts'])):
            rows = values[offsets[b]:offsets[b + 1]]
            seen['both_kinds'] += int((rows[:, 4] >= 0).any() and (rows[:, 5] >= 0).any())
    assert all(v > 0 for v in seen.values()), seen


def test_task_policy_tie_stream(oracle):
    """Without a forced draw the tie-break is word 0 of Philox(counter (first_env + b, 0, step, 0), key (seed, 0))."""
    B = 64
    values = np.zeros((B * 3, 8), np.int32)
    values[:, 4:6] = -100
    offsets = np.arange(B + 1, dtype=np.int64) * 3
    lengths = np.full(B, 3, np.int64)
    obs_self = np.zeros((B, 4), np.int32)
    got = oracle.rideshare_task_policy(values, offsets, lengths, lengths, obs_self, 'fifo_global', False, seed=11, step=5, first_env=7)
    want = [(int(oracle.philox4x32_10((7 + b, 0, 5, 0), (11, 0))[0]) * 3) >> 32 for b in range(B)]
    assert got[:, 0].tolist() == want and (got[:, 1] == 0).all()
    assert len(set(want)) == 3
