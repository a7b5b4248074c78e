"""CPU: pin the oracle's wildfire restatement against the reference's golden vectors (no GPU needed)."""
import numpy as np
import pytest

import golden_util as G
from free_range_zoo_amd import _capi


def oracle_snapshot(o):
    cfg = o.cfg
    A, show_bad = cfg.num_agents, bool(cfg.show_bad_actions)
    n = o.total_tasks()
    snap = {k: o.arrays[k] for k in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment', 'num_moves',
                                     'num_burnouts', 'env_task_count', 'agent_task_count', 'rewards', 'terminations',
                                     'truncations', 'burnouts', 'putouts')}
    snap['task_values'], snap['task_offsets'] = o.task_values[:n], o.task_offsets
    k = 2 + int(cfg.observe_other_power) + int(cfg.observe_other_suppressant)
    for a in range(A):
        if show_bad:  # wildfire.py:656-658: the action mapping is the full task list, bad ones listed separately
            snap[f'act_map_values_{a}'], snap[f'act_map_offsets_{a}'] = o.obs_map_values[:n], o.task_offsets
            snap[f'bad_map_values_{a}'], snap[f'bad_map_offsets_{a}'] = o.bad_map(a)
        else:
            snap[f'act_map_values_{a}'], snap[f'act_map_offsets_{a}'] = o.action_map(a)
        snap[f'obs_map_values_{a}'], snap[f'obs_map_offsets_{a}'] = o.obs_map_values[:n], o.task_offsets
        snap[f'obs_self_{a}'] = o.obs_self[a]
        snap[f'obs_others_{a}'] = o.obs_others[a].reshape(cfg.parallel_envs, A - 1, k)
        snap[f'cumulative_rewards_{a}'] = o.cumulative_rewards[a]
    return snap


@pytest.mark.parametrize('name', G.trajectories('wildfire'))
def test_trajectory_matches_reference(oracle, name):
    data = np.load(G.golden_path(name))
    cfg = G.load_cfg(data, _capi.frz_wildfire_cfg)
    o = oracle.WildfireOracle(cfg)
    o.reset()
    G.compare_wildfire(oracle_snapshot(o), data, 'r_', cfg.num_agents, f'{name} reset')
    B, HW, A = cfg.parallel_envs, cfg.grid_height * cfg.grid_width, cfg.num_agents
    frozen_steps = 0
    for t in range(int(data['steps'])):
        p = f's{t}_'
        if bool(data[p + 'stepped']):
            fr, ar = data[p + 'field_randomness'], data[p + 'agent_randomness']
        else:  # reference early-out (utils/env.py:211-213): no randomness was drawn
            frozen_steps += 1
            fr, ar = np.zeros((3, B, HW), np.float32), np.zeros((5, B, A), np.float32)
        o.step(data[p + 'actions'], fr, ar)
        G.compare_wildfire(oracle_snapshot(o), data, p, A, f'{name} step {t}')
        finished = (o.terminations.all(axis=0) | o.truncations.all(axis=0))
        G.assert_same(finished, data[p + 'finished'], f'{name} step {t} finished')
    assert int(o.error_flags[0]) == 0
    if 'no_truncation' not in name:
        assert frozen_steps >= 0


def _cfg_for(case, **flags):
    cfg = _capi.frz_wildfire_cfg()
    for k, v in flags.items():
        setattr(cfg, k, v)
    return cfg


def test_known_answer_transitions(oracle):
    """Every forward() call made by the reference's own transition unit tests (tests/.../transitions/test_*.py)."""
    import ctypes
    lib = oracle.lib()
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
    seen = set()
    for case in G.known_answers('wildfire'):
        cls = case['cls']
        seen.add(cls)
        what = f"{cls} / {case['test']}"
        supp = np.ascontiguousarray(case['in_suppressants'], np.float32)
        cap = np.ascontiguousarray(case['in_capacity'], np.float32)
        equip = np.ascontiguousarray(case['in_equipment'], np.int32)
        fires = np.ascontiguousarray(case['in_fires'], np.int32)
        intensity = np.ascontiguousarray(case['in_intensity'], np.int32)
        fuel = np.ascontiguousarray(case['in_fuel'], np.int32)
        B, H, W = fires.shape
        n_agent, n_cell = supp.size, fires.size
        cfg = _capi.frz_wildfire_cfg()
        cfg.parallel_envs, cfg.grid_height, cfg.grid_width, cfg.num_agents = B, H, W, supp.shape[1]
        r = np.ascontiguousarray(np.broadcast_to(case['arg_randomness_source'], case['arg_randomness_source'].shape), np.float32)
        if cls == 'SuppressantDecreaseTransition':
            cfg.stochastic_suppressant_decrease = int(case['buf_stochastic_decrease'])
            cfg.suppressant_decrease_probability = float(case['buf_decrease_probability'])
            users = np.ascontiguousarray(case['arg_used_suppressants'], np.uint8)
            lib.frz_oracle_wf_suppressant_decrease(ctypes.byref(cfg), ptr(supp), ptr(users), ptr(r), ctypes.c_int64(n_agent))
        elif cls == 'EquipmentTransition':
            states = case['buf_equipment_states']
            cfg.num_equipment_states = states.shape[0]
            cfg.stochastic_repair, cfg.repair_probability = int(case['buf_stochastic_repair']), float(case['buf_repair_probability'])
            cfg.stochastic_degrade, cfg.degrade_probability = int(case['buf_stochastic_degrade']), float(case['buf_degrade_probability'])
            cfg.critical_error = int(case['buf_critical_error'])
            cfg.critical_error_probability = float(case['buf_critical_error_probability'])
            lib.frz_oracle_wf_equipment(ctypes.byref(cfg), ptr(equip), ptr(r), ctypes.c_int64(n_agent))
        elif cls == 'SuppressantRefillTransition':
            bonuses = case['buf_equipment_bonuses']
            for s, v in enumerate(bonuses):
                cfg.equipment_states[s][0] = float(v)
            cfg.stochastic_refill, cfg.suppressant_refill_probability = int(case['buf_stochastic_refill']), float(case['buf_refill_probability'])
            refills = np.ascontiguousarray(case['arg_refilled_suppressants'], np.uint8)
            inc = np.zeros(n_agent, np.uint8)
            lib.frz_oracle_wf_suppressant_refill(ctypes.byref(cfg), ptr(supp), ptr(cap), ptr(equip), ptr(refills), ptr(r), ptr(inc),
                                                 ctypes.c_int64(n_agent))
            if 'ret_0' in case:
                G.assert_same(inc.reshape(supp.shape).astype(bool), case['ret_0'], what + ' increase mask')
        elif cls == 'CapacityTransition':
            caps, cum = case['buf_possible_capacities'], case['buf_capacity_probabilities']
            cfg.num_capacities = len(caps)
            for k in range(len(caps)):
                cfg.possible_capacities[k], cfg.capacity_cumprobs[k] = float(caps[k]), float(cum[k])
            cfg.stochastic_switch, cfg.tank_switch_probability = int(case['buf_stochastic_switch']), float(case['buf_tank_switch_probability'])
            targets = np.ascontiguousarray(case['arg_targets'], np.uint8)
            r0, r1 = np.ascontiguousarray(r[0]), np.ascontiguousarray(r[1])
            lib.frz_oracle_wf_capacity(ctypes.byref(cfg), ptr(supp), ptr(cap), ptr(targets), ptr(r0), ptr(r1), ctypes.c_int64(n_agent))
        elif cls in ('FireIncreaseTransition', 'FireDecreaseTransition'):
            attack = np.ascontiguousarray(case['arg_attack_counts'], np.float32)
            mask = np.zeros(n_cell, np.uint8)
            r = np.ascontiguousarray(np.broadcast_to(case['arg_randomness_source'], fires.shape), np.float32)
            if cls == 'FireIncreaseTransition':
                cfg.num_fire_states = int(case['buf_burnout_state']) + 1
                cfg.stochastic_increase = int(case['buf_stochastic_increase'])
                cfg.intensity_increase_probability = float(case['buf_intensity_increase_probability'])
                cfg.stochastic_burnouts, cfg.burnout_probability = int(case['buf_stochastic_burnouts']), float(case['buf_burnout_probability'])
                lib.frz_oracle_wf_fire_increase(ctypes.byref(cfg), ptr(fires), ptr(intensity), ptr(fuel), ptr(attack), ptr(r), ptr(mask),
                                                ctypes.c_int64(n_cell))
            else:
                cfg.stochastic_decrease = int(case['buf_stochastic_decrease'])
                cfg.intensity_decrease_probability = float(case['buf_decrease_probability'])
                cfg.extra_power_decrease_bonus = float(case['buf_extra_power_decrease_bonus'])
                lib.frz_oracle_wf_fire_decrease(ctypes.byref(cfg), ptr(fires), ptr(intensity), ptr(fuel), ptr(attack), ptr(r), ptr(mask),
                                                ctypes.c_int64(n_cell))
            if 'ret_0' in case:
                G.assert_same(mask.reshape(fires.shape).astype(bool), case['ret_0'], what + ' mask')
        elif cls == 'FireSpreadTransition':
            w = case['buf_fire_spread_weights'].reshape(3, 3)
            cfg.spread_n, cfg.spread_w, cfg.spread_e, cfg.spread_s = float(w[0, 1]), float(w[1, 0]), float(w[1, 2]), float(w[2, 1])
            cfg.random_ignition = float(case['attr_fire_random_spread_weight'])
            cfg.use_fire_fuel = int(case['buf_use_fire_fuel'])
            ign = np.broadcast_to(case['buf_ignition_temperatures'], (H, W)).reshape(-1)
            for c in range(H * W):
                cfg.ignition_temp[c] = int(ign[c])
            r = np.ascontiguousarray(np.broadcast_to(case['arg_randomness_source'], fires.shape), np.float32)
            lib.frz_oracle_wf_fire_spread(ctypes.byref(cfg), ptr(fires), ptr(intensity), ptr(fuel), ptr(r), ctypes.c_int64(B))
        else:
            raise AssertionError(f'unhandled transition {cls}')
        for name, arr in (('fires', fires), ('intensity', intensity), ('fuel', fuel), ('suppressants', supp), ('capacity', cap),
                          ('equipment', equip)):
            G.assert_same(arr, case['out_' + name], f'{what} {name}')
    assert seen == {'SuppressantDecreaseTransition', 'EquipmentTransition', 'SuppressantRefillTransition', 'CapacityTransition',
                    'FireIncreaseTransition', 'FireDecreaseTransition', 'FireSpreadTransition'}


def test_conv_accumulation_order(oracle):
    """fire_spreads.py:46: conv2d of a 0/1 map == N, W, E, S accumulated in that order (float32)."""
    import ctypes
    data = np.load(G.golden_path('conv_order.npz'))
    i = 0
    while f'w{i}' in data.files:
        w, lit, out = data[f'w{i}'], data[f'lit{i}'], data[f'out{i}']
        B, H, W = lit.shape
        cfg = _capi.frz_wildfire_cfg()
        cfg.parallel_envs, cfg.grid_height, cfg.grid_width = B, H, W
        cfg.spread_n, cfg.spread_w, cfg.spread_e, cfg.spread_s = float(w[0, 1]), float(w[1, 0]), float(w[1, 2]), float(w[2, 1])
        # probe: every cell unlit-and-ignitable copy next to the lit map => spread iff r < p; bisect p through r
        fires = np.where(lit > 0, 1, -1).astype(np.int32)
        intensity = np.where(lit > 0, 1, 0).astype(np.int32)
        fuel = np.ones_like(fires)
        for c in range(H * W):
            cfg.ignition_temp[c] = 9
        # r exactly equal to the torch probability must NOT spread, the next float below must
        below = np.where(out > 0, np.nextafter(out, np.float32(-1)), np.float32(0.5)).astype(np.float32)
        for r, expect in ((out, False), (below, True)):
            f, it = fires.copy(), intensity.copy()
            oracle.lib().frz_oracle_wf_fire_spread(ctypes.byref(cfg), ctypes.c_void_p(f.ctypes.data), ctypes.c_void_p(it.ctypes.data),
                                                   ctypes.c_void_p(fuel.ctypes.data),
                                                   ctypes.c_void_p(np.ascontiguousarray(r, np.float32).ctypes.data), ctypes.c_int64(B))
            ignitable = (lit == 0) & (out > 0)
            spread = (f > 0) & (lit == 0)
            G.assert_same(spread, ignitable & expect, f'conv case {i} expect={expect}')
        i += 1
    assert i >= 5


def test_in_range_truth_table(oracle):
    """tests/free_range_zoo/envs/wildfire/env/utils/test_in_range_check.py semantics: chebyshev <= range."""
    f = oracle.lib().frz_oracle_in_range_chebyshev
    import ctypes
    f.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_float]
    assert f(0, 0, 1, 1, 1.0) == 1 and f(0, 0, 2, 1, 1.0) == 0 and f(0, 0, 0, 0, 0.0) == 1
    assert f(3, 3, 1, 2, 2.0) == 1 and f(3, 3, 0, 3, 2.0) == 0 and f(0, 0, 1, 0, 0.5) == 0


def _baseline_cases():
    data = np.load(G.golden_path('baselines_wildfire.npz'))
    for i in range(int(data['cases'])):
        p = f'c{i}_'
        yield i, {k[len(p):]: data[k] for k in data.files if k.startswith(p)}


def check_extreme_answer(got, case, kind, what):
    """Exact where the extreme intensity is unique; where the reference broke a tie with torch's global generator, any tied index."""
    want = case[kind]
    counts, lengths, values = case['task_counts'], case['map_lengths'], case['task_values']
    offsets = np.concatenate([[0], np.cumsum(counts)])
    assert np.array_equal(got[:, 1], want[:, 1]), f'{what}: action column'
    for b in range(len(lengths)):
        n = int(lengths[b])
        if lengths.sum() == 0 or n == 0:
            assert got[b, 0] == want[b, 0] == -1, f'{what}: env {b} has no mapped task'
            continue
        intensity = values[offsets[b]:offsets[b] + n, 3]
        best = intensity.min() if kind == 'weakest' else intensity.max()
        assert intensity[want[b, 0]] == best, f'{what}: the recorded answer is not an extreme?!'
        if (intensity == best).sum() == 1:
            assert got[b, 0] == want[b, 0], f'{what}: env {b}'
        else:
            assert 0 <= got[b, 0] < n and intensity[got[b, 0]] == best, f'{what}: env {b} (tie)'


def test_scripted_baselines_match_reference_answers(oracle):
    """Strongest / Weakest baselines of the reference (envs/wildfire/baselines) on 154 recorded (observation, mapping) pairs."""
    seen_tie = seen_empty = 0
    for i, case in _baseline_cases():
        counts, lengths = case['task_counts'], case['map_lengths']
        task_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        map_offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        for kind in ('strongest', 'weakest'):
            got = oracle.wildfire_extreme_policy(case['task_values'], task_offsets, map_offsets, lengths, case['obs_self'], kind == 'weakest',
                                                 seed=5, step=i)
            check_extreme_answer(got, case, kind, f'case {i} {kind}')
        seen_empty += int((lengths == 0).any())
        for b in range(len(lengths)):
            v = case['task_values'][task_offsets[b]:task_offsets[b] + lengths[b], 3]
            seen_tie += int(len(v) > 1 and (v == v.max()).sum() > 1)
    assert seen_tie > 0 and seen_empty > 0  # the fixtures exercise ties and empty mappings

