"""ctypes wrapper of the CPU oracle (oracle/libfrz_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is the checker
(and the reported CPU baseline), never a product path.  Parity status: pinned (see oracle/frz_oracle.h).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from free_range_zoo_amd import _capi  # noqa: E402  (struct definitions only; does not load libfrz_hip.so)
from free_range_zoo_amd._cstruct import parse_header  # noqa: E402

LIB_PATH = os.path.join(_HERE, 'libfrz_oracle.so')
_, _STRUCTS = parse_header(os.path.join(_HERE, 'frz_oracle.h'), known=_capi.STRUCTS)
frz_oracle_wildfire_bufs = _STRUCTS['frz_oracle_wildfire_bufs']
frz_oracle_cybersecurity_bufs = _STRUCTS['frz_oracle_cybersecurity_bufs']
frz_oracle_rideshare_bufs = _STRUCTS['frz_oracle_rideshare_bufs']

_lib = None


def build() -> None:
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.frz_oracle_philox_uniform.restype = ctypes.c_float
        _lib.frz_oracle_philox_uniform.argtypes = [ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    return _lib


def _ptr(a: np.ndarray) -> ctypes.c_void_p:
    assert a.flags['C_CONTIGUOUS']
    return ctypes.c_void_p(a.ctypes.data)


def others_width(cfg) -> int:
    return 2 + int(bool(cfg.observe_other_power)) + int(bool(cfg.observe_other_suppressant))


class WildfireOracle:
    """Holds one wildfire env batch in the reference's batch-major layout and steps it with the C oracle."""

    def __init__(self, cfg):
        self.cfg = cfg
        B, HW, A = cfg.parallel_envs, cfg.grid_height * cfg.grid_width, cfg.num_agents
        cap = B * HW
        k = others_width(cfg)
        z = np.zeros
        self.arrays = dict(
            fires=z((B, HW), np.int32), intensity=z((B, HW), np.int32), fuel=z((B, HW), np.int32),
            suppressants=z((B, A), np.float32), capacity=z((B, A), np.float32), equipment=z((B, A), np.int32),
            num_moves=z(B, np.int32), num_burnouts=z(B, np.int32),
            rewards=z((A, B), np.float32), cumulative_rewards=z((A, B), np.float32),
            terminations=z((A, B), np.uint8), truncations=z((A, B), np.uint8),
            burnouts=z(B, np.int64), putouts=z(B, np.int64),
            obs_self=z((A, B, 4), np.float32), obs_others=z((A, B, max(A - 1, 0) * k), np.float32),
            task_values=z((cap, 4), np.int64), task_offsets=z(B + 1, np.int64), obs_map_values=z(cap, np.int64),
            act_map_values=z((A, cap), np.int64), act_map_offsets=z((A, B + 1), np.int64),
            bad_map_values=z((A, cap), np.int64), bad_map_offsets=z((A, B + 1), np.int64),
            env_task_count=z(B, np.int64), agent_task_count=z((A, B), np.int32),
            error_flags=z(1, np.uint32), frozen=z(2, np.int32),
        )
        self.bufs = frz_oracle_wildfire_bufs()
        for name, arr in self.arrays.items():
            setattr(self.bufs, name, _ptr(arr))

    def __getattr__(self, name):
        arrays = self.__dict__.get('arrays', {})
        if name in arrays:
            return arrays[name]
        raise AttributeError(name)

    def reset(self):
        assert lib().frz_oracle_wildfire_reset(ctypes.byref(self.cfg), ctypes.byref(self.bufs)) == 0

    def rebuild(self):
        assert lib().frz_oracle_wildfire_rebuild(ctypes.byref(self.cfg), ctypes.byref(self.bufs)) == 0

    def step(self, actions: np.ndarray, field_randomness: np.ndarray, agent_randomness: np.ndarray):
        """actions int32 [A,B,2]; field_randomness f32 [3,B,H*W] (or [3,B,H,W]); agent_randomness f32 [5,B,A]."""
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        fr = np.ascontiguousarray(field_randomness, dtype=np.float32)
        ar = np.ascontiguousarray(agent_randomness, dtype=np.float32)
        B, HW, A = self.cfg.parallel_envs, self.cfg.grid_height * self.cfg.grid_width, self.cfg.num_agents
        assert actions.shape == (A, B, 2) and fr.size == 3 * B * HW and ar.size == 5 * B * A
        assert lib().frz_oracle_wildfire_step(ctypes.byref(self.cfg), ctypes.byref(self.bufs), _ptr(actions), _ptr(fr), _ptr(ar)) == 0

    def batch_totals(self) -> np.ndarray:
        """int64 [A + 3]: (lit fires, fires agent a can attack ..., envs not terminated, envs not truncated) of this batch, as the next step's
        batch-global tests read them."""
        A = self.cfg.num_agents
        out = np.zeros(A + 3, np.int64)
        out[0] = self.env_task_count.sum()
        out[1:1 + A] = self.agent_task_count.sum(axis=1)
        out[A + 1] = (~self.terminations[0].astype(bool)).sum()
        out[A + 2] = (~self.truncations[0].astype(bool)).sum()
        return out

    def set_global_totals(self, totals) -> None:
        """The totals of the batch this one is a shard of (None: it is the whole batch) for the next step."""
        self._global_totals = None if totals is None else np.ascontiguousarray(totals, np.int64)
        self.bufs.global_totals = None if totals is None else _ptr(self._global_totals)

    def reset_masked(self, mask, seeds: np.ndarray = None, seed_increment: int = 0):
        """reset_batches with the selection as a mask (None: the finished envs); ``seeds`` (int32 [B]) moves on in place for the selected envs."""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        assert seeds is None or (seeds.dtype == np.int32 and seeds.flags['C_CONTIGUOUS'])
        assert lib().frz_oracle_wildfire_reset_masked(ctypes.byref(self.cfg), ctypes.byref(self.bufs), None if m is None else _ptr(m),
                                                      None if seeds is None else _ptr(seeds), ctypes.c_int32(seed_increment)) == 0

    def rollout(self, env_seeds: np.ndarray, policy_seed: int, first_step: int, n_steps: int) -> int:
        """``n_steps`` x (uniform random policy, the step's Philox randomness, step) in ONE C call (the loop smoke() drives from Python);
        the interpreter lock is released for its whole duration, so host threads stepping their own shards run in parallel."""
        B, HW, A = self.cfg.parallel_envs, self.cfg.grid_height * self.cfg.grid_width, self.cfg.num_agents
        if getattr(self, '_scratch', None) is None:
            self._scratch = (np.zeros((A, B, 2), np.int32), np.zeros((3, B, HW), np.float32), np.zeros((5, B, A), np.float32))
        actions, field, agent = self._scratch
        seeds = np.ascontiguousarray(env_seeds, np.int32)
        done = lib().frz_oracle_wildfire_rollout(ctypes.byref(self.cfg), ctypes.byref(self.bufs), _ptr(seeds), ctypes.c_uint64(policy_seed),
                                                 ctypes.c_uint64(first_step), ctypes.c_int32(n_steps), _ptr(actions), _ptr(field), _ptr(agent))
        assert done == n_steps, done
        return done

    # jagged helpers -------------------------------------------------------------------------------------
    def total_tasks(self) -> int:
        return int(self.task_offsets[-1])

    def tasks(self):
        n = self.total_tasks()
        return self.task_values[:n], self.task_offsets

    def action_map(self, a: int):
        off = self.act_map_offsets[a]
        return self.act_map_values[a, :int(off[-1])], off

    def bad_map(self, a: int):
        off = self.bad_map_offsets[a]
        return self.bad_map_values[a, :int(off[-1])], off


def mt19937_seed(seeds: np.ndarray):
    seeds = np.ascontiguousarray(seeds, dtype=np.int32)
    B = seeds.shape[0]
    state = np.zeros((B, 624), np.uint32)
    index = np.zeros(B, np.int32)
    lib().frz_oracle_mt19937_seed(_ptr(state), _ptr(index), _ptr(seeds), ctypes.c_int64(B))
    return state, index


def mt19937_generate(state: np.ndarray, index: np.ndarray, events: int, count: int) -> np.ndarray:
    B = state.shape[0]
    out = np.zeros((events, B, count), np.float32)
    lib().frz_oracle_mt19937_generate(_ptr(state), _ptr(index), _ptr(out), ctypes.c_int64(events), ctypes.c_int64(count),
                                      ctypes.c_int64(B))
    return out


def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    out = (ctypes.c_uint32 * 4)()
    lib().frz_oracle_philox4x32_10(c, k, out)
    return list(out)


def philox_uniform(seed: int, step: int, draw: int, stream: int = 0) -> float:
    return float(lib().frz_oracle_philox_uniform(seed, step, draw, stream))


def wildfire_philox_randomness(cfg, seeds: np.ndarray, num_moves: np.ndarray):
    B, HW, A = cfg.parallel_envs, cfg.grid_height * cfg.grid_width, cfg.num_agents
    field, agent = np.zeros((3, B, HW), np.float32), np.zeros((5, B, A), np.float32)
    seeds, num_moves = np.ascontiguousarray(seeds, np.int32), np.ascontiguousarray(num_moves, np.int32)
    lib().frz_oracle_wildfire_philox_randomness(ctypes.byref(cfg), _ptr(seeds), _ptr(num_moves), _ptr(field), _ptr(agent))
    return field, agent


def wildfire_random_policy(cfg, agent_task_count: np.ndarray, env_task_count: np.ndarray, env_seeds: np.ndarray, seed: int,
                           step: int) -> np.ndarray:
    actions = np.zeros((cfg.num_agents, cfg.parallel_envs, 2), np.int32)
    atc, etc = np.ascontiguousarray(agent_task_count, np.int32), np.ascontiguousarray(env_task_count, np.int64)
    seeds = np.ascontiguousarray(env_seeds, np.int32)
    lib().frz_oracle_wildfire_random_policy(ctypes.byref(cfg), _ptr(atc), _ptr(etc), _ptr(seeds), ctypes.c_uint64(seed),
                                            ctypes.c_uint64(step), _ptr(actions))
    return actions


def wildfire_extreme_policy(task_values, task_offsets, map_offsets, map_lengths, obs_self, weakest: bool, seed: int, step: int,
                            first_env: int = 0) -> np.ndarray:
    """envs/wildfire/baselines/strongest.py / weakest.py on the jagged observation -> int32 [B, 2]."""
    tv = np.ascontiguousarray(task_values, np.int64).reshape(-1, 4)
    to, mo = np.ascontiguousarray(task_offsets, np.int64), np.ascontiguousarray(map_offsets, np.int64)
    ml, ob = np.ascontiguousarray(map_lengths, np.int64), np.ascontiguousarray(obs_self, np.float32)
    B = ml.shape[0]
    if tv.shape[0] == 0:
        tv = np.zeros((1, 4), np.int64)
    actions = np.zeros((B, 2), np.int32)
    lib().frz_oracle_wildfire_extreme_policy(_ptr(tv), _ptr(to), _ptr(mo), _ptr(ml), _ptr(ob), ctypes.c_int64(B), ctypes.c_int(int(weakest)),
                                             ctypes.c_uint64(seed), ctypes.c_uint64(step), ctypes.c_int64(first_env), _ptr(actions))
    return actions


RIDESHARE_TASK_POLICIES = ('greedy_focus', 'greedy_global', 'fifo_focus', 'fifo_global')


def rideshare_task_policy(task_values, task_offsets, task_lengths, map_lengths, obs_self, kind: str, diagonal: bool, seed: int = 0, step: int = 0,
                          first_env: int = 0, forced_pick=None, return_ties: bool = False):
    """envs/rideshare/baselines/{greedy,fifo}_T{focus,global}.py on the jagged observation -> int32 [B, 2] (and the tie counts)."""
    tv = np.ascontiguousarray(task_values, np.int32).reshape(-1, 8)
    to, tl = np.ascontiguousarray(task_offsets, np.int64), np.ascontiguousarray(task_lengths, np.int64)
    ml, ob = np.ascontiguousarray(map_lengths, np.int64), np.ascontiguousarray(obs_self, np.int32)
    B = ml.shape[0]
    if tv.shape[0] == 0:
        tv = np.zeros((1, 8), np.int32)
    actions, ties = np.zeros((B, 2), np.int32), np.zeros(B, np.int64)
    forced = None if forced_pick is None else np.ascontiguousarray(forced_pick, np.int64)
    lib().frz_oracle_rideshare_task_policy(_ptr(tv), _ptr(to), _ptr(tl), _ptr(ml), _ptr(ob), ctypes.c_int64(B),
                                           ctypes.c_int(RIDESHARE_TASK_POLICIES.index(kind)), ctypes.c_int(int(diagonal)), ctypes.c_uint64(seed),
                                           ctypes.c_uint64(step), ctypes.c_int64(first_env), None if forced is None else _ptr(forced), _ptr(ties),
                                           _ptr(actions))
    return (actions, ties) if return_ties else actions


CYBER_FOCUS_POLICIES = ('patched_attacker', 'exploited_attacker', 'patched_defender', 'exploited_defender', 'camp_defender')


def cyber_focus_policy(tasks, obs_self, kind: str, target_node, time_focused, actions, subnetwork_states: int = 0, camp_target: int = 0,
                       mapping_numel: int = 1, seed: int = 0, step: int = 0, first_env: int = 0, forced_pick=None, across_nodes: bool = False):
    """envs/cybersecurity/baselines/{patched,exploited,camp}.py observe() on the [B, N, F] task observation; the state arrays
    (int32 target_node [B], time_focused [B], actions [B, 2]) are updated in place; returns the tie counts."""
    tasks = np.ascontiguousarray(tasks, np.int64)
    B, N, F = tasks.shape
    ob = np.ascontiguousarray(obs_self, np.float32)
    for arr in (target_node, time_focused, actions):
        assert arr.dtype == np.int32 and arr.flags['C_CONTIGUOUS']
    ties = np.zeros(B, np.int64)
    forced = None if forced_pick is None else np.ascontiguousarray(forced_pick, np.int64)
    row = (N * F, F, N) if across_nodes else (N * F, 1, F)
    lib().frz_oracle_cyber_focus_policy(_ptr(tasks), ctypes.c_int64(row[0]), ctypes.c_int64(row[1]), ctypes.c_int32(row[2]), _ptr(ob),
                                        ctypes.c_int32(ob.shape[1]), ctypes.c_int64(B), ctypes.c_int(CYBER_FOCUS_POLICIES.index(kind)),
                                        ctypes.c_int32(subnetwork_states), ctypes.c_int32(camp_target), ctypes.c_int64(mapping_numel),
                                        ctypes.c_uint64(seed), ctypes.c_uint64(step), ctypes.c_int64(first_env),
                                        None if forced is None else _ptr(forced), _ptr(ties), _ptr(target_node), _ptr(time_focused),
                                        _ptr(actions))
    return ties


class _ArrayOracle:
    """Common plumbing: named numpy arrays bound to a ctypes bufs struct."""

    def _bind(self, struct_cls, arrays):
        self.arrays = arrays
        self.bufs = struct_cls()
        for name, arr in arrays.items():
            setattr(self.bufs, name, _ptr(arr))

    def __getattr__(self, name):
        arrays = self.__dict__.get('arrays', {})
        if name in arrays:
            return arrays[name]
        raise AttributeError(name)


def cyber_others_cols(cfg):
    ka = int(bool(cfg.observe_other_power)) + int(bool(cfg.observe_other_presence))
    return ka, ka + int(bool(cfg.observe_other_location))


class CybersecurityOracle(_ArrayOracle):
    """One cybersecurity env batch in the reference's batch-major layout, stepped by the C oracle."""

    def __init__(self, cfg):
        self.cfg = cfg
        B, N, Att, D = cfg.parallel_envs, cfg.num_nodes, cfg.num_attackers, cfg.num_defenders
        A = Att + D
        ka, kd = cyber_others_cols(cfg)
        z = np.zeros
        self._bind(frz_oracle_cybersecurity_bufs, dict(
            network_state=z((B, N), np.int32), location=z((B, D), np.int32), presence=z((B, A), np.uint8),
            last_action=z((B, D), np.int32), num_moves=z(B, np.int32), rewards=z((A, B), np.float32),
            cumulative_rewards=z((A, B), np.float32), terminations=z((A, B), np.uint8), truncations=z((A, B), np.uint8),
            obs_self_attackers=z((Att, B, 2), np.float32), obs_self_defenders=z((D, B, 3), np.float32),
            obs_others_attackers=z((Att, B, max(Att - 1, 0) * ka), np.float32),
            obs_others_defenders=z((D, B, max(D - 1, 0) * kd), np.float32), obs_tasks=z((A, B, N, 2), np.int64),
            act_map_values=z((A, B * N), np.int32), act_map_offsets=z((A, B + 1), np.int64), obs_map_values=z((B, N), np.int32),
            obs_map_offsets=z(B + 1, np.int64), env_task_count=z(B, np.int32), agent_task_count=z((A, B), np.int32),
            error_flags=z(1, np.uint32), frozen=z(2, np.int32)))

    def reset(self):
        assert lib().frz_oracle_cybersecurity_reset(ctypes.byref(self.cfg), ctypes.byref(self.bufs)) == 0

    def rebuild(self):
        assert lib().frz_oracle_cybersecurity_rebuild(ctypes.byref(self.cfg), ctypes.byref(self.bufs)) == 0

    def step(self, actions, network_randomness, agent_randomness):
        """actions int32 [A,B,2]; network_randomness f32 [1,B,N]; agent_randomness f32 [1,B,A]."""
        actions = np.ascontiguousarray(actions, np.int32)
        nr, ar = np.ascontiguousarray(network_randomness, np.float32), np.ascontiguousarray(agent_randomness, np.float32)
        B, N, A = self.cfg.parallel_envs, self.cfg.num_nodes, self.cfg.num_attackers + self.cfg.num_defenders
        assert actions.shape == (A, B, 2) and nr.size == B * N and ar.size == B * A
        assert lib().frz_oracle_cybersecurity_step(ctypes.byref(self.cfg), ctypes.byref(self.bufs), _ptr(actions), _ptr(nr), _ptr(ar)) == 0

    def rollout(self, env_seeds: np.ndarray, policy_seed: int, first_step: int, n_steps: int) -> int:
        """``n_steps`` x (uniform random policy, the step's Philox randomness, step) in one C call."""
        B, N, A = self.cfg.parallel_envs, self.cfg.num_nodes, self.cfg.num_attackers + self.cfg.num_defenders
        actions, network, agent = np.zeros((A, B, 2), np.int32), np.zeros((1, B, N), np.float32), np.zeros((1, B, A), np.float32)
        seeds = np.ascontiguousarray(env_seeds, np.int32)
        done = lib().frz_oracle_cybersecurity_rollout(ctypes.byref(self.cfg), ctypes.byref(self.bufs), _ptr(seeds), ctypes.c_uint64(policy_seed),
                                                      ctypes.c_uint64(first_step), ctypes.c_int32(n_steps), _ptr(actions), _ptr(network), _ptr(agent))
        assert done == n_steps, done
        return done

    def action_map(self, a):
        off = self.act_map_offsets[a]
        return self.act_map_values[a, :int(off[-1])], off


def cybersecurity_philox_randomness(cfg, seeds, num_moves):
    B, N, A = cfg.parallel_envs, cfg.num_nodes, cfg.num_attackers + cfg.num_defenders
    network, agent = np.zeros((1, B, N), np.float32), np.zeros((1, B, A), np.float32)
    seeds, num_moves = np.ascontiguousarray(seeds, np.int32), np.ascontiguousarray(num_moves, np.int32)
    lib().frz_oracle_cybersecurity_philox_randomness(ctypes.byref(cfg), _ptr(seeds), _ptr(num_moves), _ptr(network), _ptr(agent))
    return network, agent


def cybersecurity_random_policy(cfg, agent_task_count, location, env_seeds, seed, step):
    A = cfg.num_attackers + cfg.num_defenders
    actions = np.zeros((A, cfg.parallel_envs, 2), np.int32)
    atc, loc = np.ascontiguousarray(agent_task_count, np.int32), np.ascontiguousarray(location, np.int32)
    seeds = np.ascontiguousarray(env_seeds, np.int32)
    lib().frz_oracle_cybersecurity_random_policy(ctypes.byref(cfg), _ptr(atc), _ptr(loc), _ptr(seeds), ctypes.c_uint64(seed),
                                                 ctypes.c_uint64(step), _ptr(actions))
    return actions


class RideshareOracle(_ArrayOracle):
    """One rideshare env batch (per-env ordered passenger slots), stepped by the C oracle."""

    def __init__(self, cfg, schedule):
        self.cfg = cfg
        self.schedule = np.ascontiguousarray(schedule, np.int32)
        B, A, P = cfg.parallel_envs, cfg.num_agents, cfg.max_passengers
        cap = B * P
        z = np.zeros
        self._bind(frz_oracle_rideshare_bufs, dict(
            agents=z((B, A, 2), np.int32), passengers=z((B, P, 10), np.int32), passenger_count=z(B, np.int32), num_moves=z(B, np.int32),
            rewards=z((A, B), np.float32), cumulative_rewards=z((A, B), np.float32), terminations=z((A, B), np.uint8),
            truncations=z((A, B), np.uint8), obs_self=z((A, B, 4), np.int32), obs_others=z((A, B, max(A - 1, 0), 4), np.int32),
            task_values=z((cap, 8), np.int32), task_offsets=z(B + 1, np.int64), agent_task_values=z((A, cap, 8), np.int32),
            agent_map_values=z((A, cap), np.int64), agent_offsets=z((A, B + 1), np.int64), agent_task_states=z((A, cap), np.int32),
            env_task_count=z(B, np.int64), agent_task_count=z((A, B), np.int32), error_flags=z(1, np.uint32), frozen=z(2, np.int32)))

    def reset(self):
        assert lib().frz_oracle_rideshare_reset(ctypes.byref(self.cfg), ctypes.byref(self.bufs), _ptr(self.schedule)) == 0

    def rebuild(self):
        assert lib().frz_oracle_rideshare_rebuild(ctypes.byref(self.cfg), ctypes.byref(self.bufs)) == 0

    def step(self, actions):
        actions = np.ascontiguousarray(actions, np.int32)
        assert actions.shape == (self.cfg.num_agents, self.cfg.parallel_envs, 2)
        assert lib().frz_oracle_rideshare_step(ctypes.byref(self.cfg), ctypes.byref(self.bufs), _ptr(self.schedule), _ptr(actions)) == 0

    def random_policy(self, seed, step):
        actions = np.zeros((self.cfg.num_agents, self.cfg.parallel_envs, 2), np.int32)
        lib().frz_oracle_rideshare_random_policy(ctypes.byref(self.cfg), ctypes.byref(self.bufs), ctypes.c_uint64(seed), ctypes.c_uint64(step),
                                                 _ptr(actions))
        return actions

    def table(self):
        """The reference's global passenger table [P, 11] (env id first), rebuilt from the per-env slots."""
        rows = []
        for b in range(self.cfg.parallel_envs):
            n = int(self.passenger_count[b])
            rows.append(np.concatenate([np.full((n, 1), b, np.int32), self.passengers[b, :n]], axis=1))
        return np.concatenate(rows, axis=0) if rows else np.zeros((0, 11), np.int32)

    def agent_tasks(self, a):
        off = self.agent_offsets[a]
        n = int(off[-1])
        return self.agent_task_values[a, :n], self.agent_map_values[a, :n], off
