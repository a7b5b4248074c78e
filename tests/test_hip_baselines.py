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


# ------------------------------------------------------------------------------------------------------------------
# rideshare: greedy / FIFO, task-focused / task-global (frz_rideshare_task_policy)
# ------------------------------------------------------------------------------------------------------------------
from test_oracle_rideshare import RIDESHARE_BOTS, check_task_policy_answer  # noqa: E402
from test_oracle_rideshare import _baseline_cases as _rideshare_cases  # noqa: E402


def hip_task_policy(task_values, task_offsets, task_lengths, map_lengths, obs_self, kind, diagonal, seed=0, step=0, first_env=0, tie_draws=None):
    from free_range_zoo_amd import _capi
    dev = torch.device('cuda')
    tv = torch.as_tensor(np.ascontiguousarray(task_values, np.int32).reshape(-1, 8), device=dev)
    if tv.shape[0] == 0:
        tv = torch.zeros((1, 8), dtype=torch.int32, device=dev)
    to, tl = torch.as_tensor(np.ascontiguousarray(task_offsets, np.int64), device=dev), torch.as_tensor(np.ascontiguousarray(task_lengths, np.int64), device=dev)
    ml, ob = torch.as_tensor(np.ascontiguousarray(map_lengths, np.int64), device=dev), torch.as_tensor(np.ascontiguousarray(obs_self, np.int32), device=dev)
    draws = None if tie_draws is None else torch.as_tensor(np.ascontiguousarray(tie_draws, np.int64), device=dev)
    B = ml.shape[0]
    out = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    _capi.check(_capi.lib().frz_rideshare_task_policy(tv.data_ptr(), to.data_ptr(), tl.data_ptr(), ml.data_ptr(), ob.data_ptr(), B,
                                                      RIDESHARE_BOTS.index(kind), int(diagonal), seed, step, first_env,
                                                      None if draws is None else draws.data_ptr(), out.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), 'frz_rideshare_task_policy')
    return out.cpu().numpy()


def test_rideshare_recorded_reference_answers_and_oracle(oracle):
    """The reference's four agents on 420 recorded observations: with the recorded torch.randint draws replayed the HIP answers are
    the reference's (tied or not); with the build's own Philox tie stream they are the oracle's."""
    for i, case in _rideshare_cases():
        args = (case['task_values'], case['task_offsets'], case['task_counts'], case['map_lengths'], case['obs_self'])
        diagonal = bool(case['diagonal'])
        for kind in RIDESHARE_BOTS:
            if int(case[kind + '_valid']):
                def policy(forced, kind=kind):
                    return hip_task_policy(*args, kind, diagonal, tie_draws=forced), case[kind + '_ties']  # tie counts: oracle-side check
                check_task_policy_answer(policy, case, kind, f'case {i} {kind}')
            got = hip_task_policy(*args, kind, diagonal, seed=5, step=i, first_env=3)
            want = oracle.rideshare_task_policy(*args, kind, diagonal, seed=5, step=i, first_env=3)
            assert np.array_equal(got, want), f'case {i} {kind}: HIP vs oracle'


@pytest.mark.parametrize('diagonal', [False, True])
def test_rideshare_task_policy_large_batch_vs_oracle(oracle, diagonal):
    rng = np.random.default_rng(8 + int(diagonal))
    B = 50021
    counts = rng.integers(0, 9, B).astype(np.int64)
    counts[rng.random(B) < 0.1] = 0
    total = int(counts.sum())
    values = np.zeros((total, 8), np.int32)
    values[:, :4] = rng.integers(0, 6, (total, 4))  # small grid: many equal distances
    state = rng.integers(0, 3, total)
    values[:, 4] = np.where(state == 1, rng.integers(0, 4, total), -100)
    values[:, 5] = np.where(state == 2, rng.integers(0, 4, total), -100)
    values[:, 6] = rng.integers(1, 11, total)
    values[:, 7] = rng.integers(0, 5, total)
    obs_self = rng.integers(0, 6, (B, 4)).astype(np.int32)
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    # rows of an env need not be packed back to back (the env's persistent buffers reserve max_passengers rows per env)
    for offsets, vals in ((starts, values), (None, None)):
        if offsets is None:
            stride = 9
            vals = np.full((B * stride, 8), 77, np.int32)
            offsets = np.arange(B + 1, dtype=np.int64) * stride
            for b in np.flatnonzero(counts):
                vals[offsets[b]:offsets[b] + counts[b]] = values[starts[b]:starts[b + 1]]
        for kind in RIDESHARE_BOTS:
            got = hip_task_policy(vals, offsets, counts, counts, obs_self, kind, diagonal, seed=(1 << 40) + 17, step=(1 << 33) + 2, first_env=1000)
            want, ties = oracle.rideshare_task_policy(vals, offsets, counts, counts, obs_self, kind, diagonal, seed=(1 << 40) + 17,
                                                      step=(1 << 33) + 2, first_env=1000, return_ties=True)
            assert np.array_equal(got, want), kind
            assert (ties > 1).sum() > B // 20 and ((got[:, 0] == -1) == (counts == 0)).all()
    shorter = np.maximum(counts - 1, 0)  # a mapping shorter than the task list: only its first rows are candidates
    got = hip_task_policy(values, starts, counts, shorter, obs_self, 'fifo_focus', diagonal, seed=3, step=4)
    assert np.array_equal(got, oracle.rideshare_task_policy(values, starts, counts, shorter, obs_self, 'fifo_focus', diagonal, seed=3, step=4))
    assert (got[:, 0] < np.maximum(shorter, 1)).all()


@pytest.mark.parametrize('diagonal', [False, True])
def test_rideshare_baselines_drive_a_rollout_through_the_action_task_wrapper(oracle, diagonal):
    """The reference's usage on the rideshare env: agents observe() the wrapped observation and act(); every decision is checked
    against the oracle on the tensors the agent saw, the env accepts every action, and the focused agents deliver passengers."""
    from free_range_zoo_amd.envs import rideshare_v0
    from free_range_zoo_amd.envs.rideshare.baselines import (FirstInFirstOutTfocusBaseline, FirstInFirstOutTglobalBaseline, GreedyTaskFocus,
                                                             GreedyTaskGlobal, NoopBaseline, RandomBaseline)
    from free_range_zoo_amd.envs.rideshare.baselines._task_policy import jagged_parts
    from free_range_zoo_amd.wrappers import action_mapping_wrapper_v0
    B = 2000
    configuration = configs.rideshare_small(diagonal, False) if diagonal else configs.rideshare_busy(A=6, steps=24, per_step=2, grid=8, seed=2)
    env = rideshare_v0.parallel_env(configuration=configuration, parallel_envs=B, max_steps=30, device=torch.device('cuda'))
    env = action_mapping_wrapper_v0(env)
    observations, _ = env.reset(seed=torch.arange(B, dtype=torch.int32))
    makers = [GreedyTaskFocus, FirstInFirstOutTfocusBaseline, GreedyTaskGlobal, FirstInFirstOutTglobalBaseline, RandomBaseline, NoopBaseline]
    agents = {}
    for i, name in enumerate(env.agents):
        cls = makers[i % len(makers)]
        kwargs = dict(agent_configuration=configuration.agent_config, seed=i) if cls.__name__.startswith('Greedy') else {}
        agents[name] = cls(name, B, **kwargs)
    first, delivered = list(agents)[0], 0
    for t in range(20):
        actions = {}
        for name, agent in agents.items():
            agent.observe(observations[name])
            actions[name] = agent.act(env.action_space(name))
            if hasattr(agent, 'kind'):
                obs, mapping = observations[name]
                tv, to, tl = (x.cpu().numpy() for x in jagged_parts(obs['tasks']))
                _, _, ml = (x.cpu().numpy() for x in jagged_parts(mapping['agent_action_mapping']))
                want = oracle.rideshare_task_policy(tv, to, tl, ml, obs['self'].cpu().numpy(), agent.kind, diagonal, seed=agent.seed, step=t)
                assert np.array_equal(actions[name].cpu().numpy(), want), f'{name} step {t}'
        observations, rewards, terminations, truncations, infos = env.step(actions)
        delivered += int((rewards[first] > 0).sum())
    env.check()
    assert delivered > 0  # the focused greedy agent completes trips (a fare is the only positive reward)


# ------------------------------------------------------------------------------------------------------------------
# cybersecurity: stateful patched / exploited / camp agents (frz_cybersecurity_focus_policy)
# ------------------------------------------------------------------------------------------------------------------
from test_oracle_cybersecurity import CYBER_BOTS, check_focus_policy  # noqa: E402
from test_oracle_cybersecurity import _baseline_cases as _cyber_cases  # noqa: E402

CYBER_KINDS = ('patched_attacker', 'exploited_attacker', 'patched_defender', 'exploited_defender', 'camp_defender')


def hip_focus_policy(tasks, obs_self, kind, target, focused, actions, subnetwork_states=0, camp_target=0, mapping_numel=1, seed=0, step=0,
                     first_env=0, tie_draws=None, across_nodes=False):
    """Same contract as oracle.cyber_focus_policy: the three numpy state arrays are updated in place."""
    from free_range_zoo_amd import _capi
    dev = torch.device('cuda')
    tk = torch.as_tensor(np.ascontiguousarray(tasks, np.int64), device=dev)
    ob = torch.as_tensor(np.ascontiguousarray(obs_self, np.float32), device=dev)
    B, N, F = tk.shape
    row = (N * F, F, N) if across_nodes else (N * F, 1, F)
    state = [torch.as_tensor(x.copy(), device=dev) for x in (target, focused, actions)]
    draws = None if tie_draws is None else torch.as_tensor(np.ascontiguousarray(tie_draws, np.int64), device=dev)
    _capi.check(_capi.lib().frz_cybersecurity_focus_policy(tk.data_ptr(), row[0], row[1], row[2], ob.data_ptr(), ob.shape[1], B,
                                                           CYBER_KINDS.index(kind), subnetwork_states, camp_target, mapping_numel, seed, step,
                                                           first_env, None if draws is None else draws.data_ptr(), state[0].data_ptr(),
                                                           state[1].data_ptr(), state[2].data_ptr(), torch.cuda.current_stream().cuda_stream),
                'frz_cybersecurity_focus_policy')
    for dst, src in zip((target, focused, actions), state):
        dst[...] = src.cpu().numpy()


def test_cyber_recorded_reference_answers(oracle):
    """528 recorded observe() calls of the reference's agents: from the recorded agent state, with the recorded draws replayed, the
    HIP policy lands on the reference's answer and state."""
    for i, case in _cyber_cases():
        for kind in CYBER_BOTS[str(case['role'])]:
            def policy(kind, target, focused, actions, forced):
                hip_focus_policy(case['tasks'], case['obs_self'], kind, target, focused, actions, subnetwork_states=int(case['subnetwork_states']),
                                 camp_target=int(case.get('camp_target', 0)), mapping_numel=int(case['mapping_numel']), tie_draws=forced)

            check_focus_policy(policy, case, kind, f'case {i} {kind}')


@pytest.mark.parametrize('across_nodes', [False, True])
def test_cyber_focus_policy_trajectory_vs_oracle(oracle, across_nodes):
    """Large batch, eight consecutive decisions per kind with the build's own tie stream: state and answers bit-equal to the oracle."""
    rng = np.random.default_rng(12 + int(across_nodes))
    B, N, F, S = 40003, 5, 2, 6
    for kind in CYBER_KINDS:
        width = 2 if 'attacker' in kind else 3
        state_h = [np.full(B, -1, np.int32), np.zeros(B, np.int32), np.zeros((B, 2), np.int32)]
        state_o = [x.copy() for x in state_h]
        for step in range(8):
            tasks = rng.integers(0, S, (B, N, F)).astype(np.int64)
            tasks[rng.random((B, N)) < 0.3] = -100  # unobserved nodes
            tasks[rng.random(B) < 0.3] = rng.integers(0, S, (N, F))  # fully monitored envs
            obs_self = rng.random((B, width)).astype(np.float32)
            obs_self[:, 1] = rng.random(B) < 0.8
            if width == 3:
                obs_self[:, 2] = rng.integers(-1, N, B)
            kwargs = dict(subnetwork_states=S, camp_target=3, seed=(7 << 32) + 5, step=(1 << 32) + step, first_env=17, across_nodes=across_nodes)
            hip_focus_policy(tasks, obs_self, kind, *state_h, **kwargs)
            oracle.cyber_focus_policy(tasks, obs_self, kind, *state_o, **kwargs)
            for h, o in zip(state_h, state_o):
                assert np.array_equal(h, o), f'{kind} step {step}'
        if kind != 'camp_defender':
            assert (state_h[1] > 0).any() and (state_h[0] >= 0).any()
        before = [x.copy() for x in state_h]
        hip_focus_policy(tasks, obs_self, kind, *state_h, mapping_numel=0)
        assert (state_h[2] == np.array([-100, -1])).all() and np.array_equal(state_h[0], before[0]) and np.array_equal(state_h[1], before[1])


def test_cyber_baselines_drive_a_rollout_through_the_action_task_wrapper(oracle):
    from free_range_zoo_amd.envs import cybersecurity_v0
    from free_range_zoo_amd.envs.cybersecurity.baselines import (CampDefenderBaseline, ExploitedAttackerBaseline, ExploitedDefenderBaseline,
                                                                 NoopBaseline, PatchedAttackerBaseline, PatchedDefenderBaseline, RandomBaseline)
    from free_range_zoo_amd.wrappers import action_mapping_wrapper_v0
    B = 3000
    configuration = configs.cyber_rich()
    S = int(configuration.network_config.num_states)
    env = cybersecurity_v0.parallel_env(configuration=configuration, parallel_envs=B, max_steps=30, device=torch.device('cuda'),
                                        show_bad_actions=True)
    env = action_mapping_wrapper_v0(env)
    observations, _ = env.reset(seed=torch.arange(B, dtype=torch.int32))
    attackers = [lambda n: PatchedAttackerBaseline(n, B, seed=1), lambda n: ExploitedAttackerBaseline(S, n, B, seed=2), lambda n: RandomBaseline(n, B)]
    defenders = [lambda n: PatchedDefenderBaseline(n, B, seed=3), lambda n: ExploitedDefenderBaseline(S, n, B, seed=4),
                 lambda n: CampDefenderBaseline(n, B), lambda n: NoopBaseline(n, B)]
    agents, shadow = {}, {}
    for name in env.agents:
        pool = attackers if name.startswith('attacker') else defenders
        agents[name] = pool[(int(name.split('_')[-1]) - 1) % len(pool)](name)
        shadow[name] = [np.full(B, -1, np.int32), np.zeros(B, np.int32), np.zeros((B, 2), np.int32)]
    for t in range(15):
        actions = {}
        for name, agent in agents.items():
            agent.observe(observations[name])
            actions[name] = agent.act(env.action_space(name))
            if hasattr(agent, 'kind'):
                obs, mapping = observations[name]
                nodes = obs['tasks'].shape[1]
                oracle.cyber_focus_policy(obs['tasks'].cpu().numpy(), obs['self'].cpu().numpy(), agent.kind, *shadow[name], subnetwork_states=S,
                                          camp_target=getattr(agent, 'agent_index', 0) % nodes,
                                          mapping_numel=int(mapping['agent_action_mapping'].numel()), seed=agent.seed, step=t)
                assert np.array_equal(actions[name].cpu().numpy(), shadow[name][2]), f'{name} step {t}'
                assert np.array_equal(agent.target_node.cpu().numpy(), shadow[name][0]), f'{name} step {t}: target'
        observations, rewards, terminations, truncations, infos = env.step(actions)
    env.check()


# ------------------------------------------------------------------------------------------------------------------
# action space validator wrapper (wrappers/space_validator.py)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('domain', ['wildfire', 'rideshare', 'cybersecurity'])
def test_space_validator_wrapper_accepts_members_and_rejects_the_rest(domain):
    from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0
    from free_range_zoo_amd.wrappers import space_validator_wrapper_v0
    module, configuration = {'wildfire': (wildfire_v0, configs.wildfire_rich()), 'rideshare': (rideshare_v0, configs.rideshare_busy(A=4, steps=12)),
                             'cybersecurity': (cybersecurity_v0, configs.cyber_rich())}[domain]
    B = 500
    env = space_validator_wrapper_v0(module.parallel_env(configuration=configuration, parallel_envs=B, max_steps=20, device=torch.device('cuda')))
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(6):  # members of the spaces always pass, and the env agrees (no error flags)
        env.step({agent: env.action_space(agent).sample_nested() for agent in env.agents})
    env.check()
    actions = {agent: env.action_space(agent).sample_nested() for agent in env.agents}
    victim = env.agents[-1]
    actions[victim] = actions[victim].clone()
    actions[victim][B // 2] = torch.tensor([0, 7], dtype=torch.int32)  # no member has the value 7
    with pytest.raises(IndexError, match=f'{victim} in batch {B // 2}'):
        env.step(actions)
    actions[victim][B // 2] = torch.tensor([10 ** 6, 0], dtype=torch.int32)  # no such member
    with pytest.raises(IndexError):
        env.step(actions)


# ------------------------------------------------------------------------------------------------------------------
# AEC protocol (<domain>_v0.env): agents act one after the other, the simulation steps when the last one has acted
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('domain', ['wildfire', 'rideshare', 'cybersecurity'])
def test_aec_cycle_equals_the_parallel_step(domain):
    from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0
    module, configuration = {'wildfire': (wildfire_v0, configs.wildfire_openness()), 'rideshare': (rideshare_v0, configs.rideshare_busy(A=4, steps=12)),
                             'cybersecurity': (cybersecurity_v0, configs.cyber_openness())}[domain]
    B = 400
    kwargs = dict(configuration=configuration, parallel_envs=B, max_steps=6, device=torch.device('cuda'))
    aec, par = module.env(**kwargs), module.parallel_env(**kwargs)
    seed = torch.arange(B, dtype=torch.int32)
    assert aec.reset(seed=seed) is None
    par.reset(seed=seed)
    cycles = 0
    for agent in aec.agent_iter():
        assert agent == aec.agent_selection
        observation, cumulative, terminations, truncations, info = aec.last()
        assert torch.equal(cumulative, par._cumulative_rewards[agent])
        first = agent == aec.agents[0]
        if first:
            actions = {name: par.action_space(name).sample_nested() for name in par.agents}
        aec.step(actions[agent])
        if agent != aec.agents[-1]:
            assert all(not r.any() for r in aec.rewards.values())  # rewards read as zero inside a cycle
        else:
            par.step(actions)
            cycles += 1
            for name in par.agents:
                assert torch.equal(aec.rewards[name], par.rewards[name]) and torch.equal(aec.truncations[name], par.truncations[name])
                got, want = aec.observe(name), par.observe(name)
                assert torch.equal(got['self'], want['self'])
            assert torch.equal(aec.num_moves, par.num_moves)
    assert cycles == 6 and bool(aec.finished.all())
    aec.step(actions[aec.agent_selection])  # finished: a no-op that leaves the selection where it is
    assert aec.agent_selection == aec.agents[0] and torch.equal(aec.num_moves, par.num_moves)
    aec.check()


def test_wildfire_baselines_survive_an_observation_without_any_task():
    """When no env has a lit cell the jagged task observation is an EMPTY tensor (no address to hand to the kernel): the agents
    answer [-1, -1] like the reference's (strongest.py:41-43)."""
    from free_range_zoo_amd.envs import wildfire_v0
    from free_range_zoo_amd.envs.wildfire.baselines import StrongestBaseline, WeakestBaseline
    from free_range_zoo_amd.wrappers import action_mapping_wrapper_v0
    B = 3
    env = action_mapping_wrapper_v0(wildfire_v0.parallel_env(configuration=configs.wildfire_rich(), parallel_envs=B, max_steps=12,
                                                             device=torch.device('cuda')))
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    state = env.state()
    state.fires.copy_(-state.fires.abs())  # every fire out
    env.update_observations()
    for agent_cls in (StrongestBaseline, WeakestBaseline):
        for name in env.agents:
            agent = agent_cls(name, B)
            observation = env.observe(name)
            assert observation[0]['tasks'].values().numel() == 0
            agent.observe(observation)
            assert (agent.act(None) == -1).all()
    env.step({name: torch.tensor([[0, -1]] * B, dtype=torch.int32, device='cuda') for name in env.agents})
    env.check()


@pytest.mark.parametrize('domain', ['wildfire', 'rideshare', 'cybersecurity'])
@pytest.mark.parametrize('B', [1, 2])
def test_tiny_batches_run_whole_episodes_through_every_wrapper(domain, B, tmp_path):
    """B = 1 and 2 with exact shapes: empty task lists, empty jagged tensors and zero-length mappings all occur; the validator and
    action-task wrappers, the logging tap and the materialised per-env spaces keep working for a whole episode."""
    from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0
    from free_range_zoo_amd.wrappers import action_mapping_wrapper_v0, space_validator_wrapper_v0
    module, configuration = {'wildfire': (wildfire_v0, configs.wildfire_non_stochastic()), 'rideshare': (rideshare_v0, configs.rideshare_non_stochastic()),
                             'cybersecurity': (cybersecurity_v0, configs.cyber_openness())}[domain]
    env = module.parallel_env(configuration=configuration, parallel_envs=B, max_steps=25, device=torch.device('cuda'),
                              log_directory=str(tmp_path / 'logs'))
    env = action_mapping_wrapper_v0(space_validator_wrapper_v0(env))
    observations, infos = env.reset(seed=torch.arange(B, dtype=torch.int32))
    steps = 0
    while not torch.all(env.finished):
        for name in env.agents:
            space = env.action_space(name)
            assert len(space.spaces) == B  # materialise the per-env OneOf objects
            observation, mapping = observations[name]
            assert mapping['agent_action_mapping'].size(0) == B
        observations, rewards, terminations, truncations, infos = env.step({name: env.action_space(name).sample_nested() for name in env.agents})
        steps += 1
    env.check()
    env.close()
    assert 0 < steps <= 25
    rows = open(tmp_path / 'logs' / '0.csv').read().strip().splitlines()
    assert len(rows) == 1 + 1 + steps
