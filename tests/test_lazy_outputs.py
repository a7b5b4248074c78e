"""GPU: with the default exact_shapes=True a step does not read the device — the jagged outputs take their exact shape when they are first
looked at — and observation_space(agent) describes the env's current step."""
import pytest
import torch

import configs

pytestmark = pytest.mark.gpu


def _domains():
    from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0
    return {'wildfire': (wildfire_v0, configs.wildfire_openness, dict(rng='philox')),
            'cybersecurity': (cybersecurity_v0, configs.cyber_openness, dict(rng='philox')),
            'rideshare': (rideshare_v0, lambda: configs.rideshare_busy(A=4, steps=10, per_step=2, seed=2), {})}


@pytest.mark.parametrize('domain', ['wildfire', 'cybersecurity', 'rideshare'])
def test_default_step_is_sync_free_until_an_output_is_looked_at(domain, monkeypatch):
    module, build, kwargs = _domains()[domain]
    B = 512
    env = module.parallel_env(configuration=build(), parallel_envs=B, max_steps=12, device=torch.device('cuda'), **kwargs)
    assert env.exact_shapes
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    reads = []
    real = torch.Tensor.tolist
    monkeypatch.setattr(torch.Tensor, 'tolist', lambda self: (reads.append(1), real(self))[1])
    real_item = torch.Tensor.item
    monkeypatch.setattr(torch.Tensor, 'item', lambda self: (reads.append(1), real_item(self))[1])
    for t in range(5):
        out = env.step_random_policy(policy_seed=3, policy_step=t)
    obs, rewards, terminations, truncations, infos = out
    assert reads == [], 'a device-side-policy rollout read the device'
    assert set(obs) == set(env.agents) and len(obs) == len(env.agents)  # keys are known without touching the device
    assert reads == []
    first = obs[env.agents[0]]  # now the lists take their exact shapes: one read
    assert len(reads) == 1
    tasks = first['tasks']
    total = int(tasks.offsets()[-1]) if tasks.is_nested else tasks.shape[0]
    if tasks.is_nested:
        assert tasks.values().shape[0] == total
    assert obs[env.agents[-1]] is env.observations[env.agents[-1]] and len(reads) <= 2
    before = len(reads)
    env.step_random_policy(policy_seed=3, policy_step=5)
    assert len(reads) == before
    mapping = env.agent_action_mapping[env.agents[0]]  # stale outputs are rebuilt on demand
    assert mapping.offsets().shape[0] == B + 1
    env.check()


@pytest.mark.parametrize('domain', ['wildfire', 'cybersecurity', 'rideshare'])
def test_observation_space_objects(domain):
    from free_range_zoo_amd.utils.spaces import Box, Dict, Tuple
    module, build, kwargs = _domains()[domain]
    B = 64
    env = module.parallel_env(configuration=build(), parallel_envs=B, max_steps=12, device=torch.device('cuda'), **kwargs)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    for t in range(3):
        env.step_random_policy(policy_seed=1, policy_step=t)
    for a, agent in enumerate(env.agents):
        space = env.observation_space(agent)
        assert len(space) == B
        obs = env.observe(agent)
        counts = {'wildfire': env.environment_task_count, 'rideshare': env.agent_task_count[a],
                  'cybersecurity': torch.full((B, ), 3)}[domain].tolist()
        for b in (0, 17, B - 1):
            single = space[b]
            assert isinstance(single, Dict) and set(single.keys()) == {'self', 'others', 'tasks'}
            assert isinstance(single['self'], Box) and len(single['self']) == obs['self'].shape[-1]
            assert isinstance(single['others'], Tuple) and len(single['others']) == obs['others'].shape[1]
            assert isinstance(single['tasks'], Tuple) and len(single['tasks']) == counts[b]
            if len(single['others']) and obs['others'].shape[-1]:
                assert len(single['others'][0]) == obs['others'].shape[-1]


def test_reference_shaped_random_rollout_draws_inside_the_step_launch():
    """The reference's loop (docs/source/events/moasei-2026/evaluation.md test(): per agent `env.action_space(agent).sample_nested()`, then
    `env.step(actions)`): the samples are LazySample tensors; handed to step() untouched they are drawn inside the step launch — one launch
    per step — with exactly the values the policy launch produces when a sample is looked at first, and they stay readable afterwards."""
    from free_range_zoo_amd.envs import wildfire_v0
    from free_range_zoo_amd.utils.env import LazySample
    B = 3001
    fused, looked, plain = [wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=30, device=torch.device('cuda'),
                                                     rng='philox') for _ in range(3)]
    for env in (fused, looked, plain):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 5)
    for t in range(34):  # past the horizon: the last steps are frozen
        a = {agent: fused.action_space(agent).sample_nested() for agent in fused.agents}
        assert all(type(v) is LazySample for v in a.values()) and fused._pending_samples[3] is False
        live = not bool(fused.finished.all())
        fused.step(a)
        assert fused._pending_samples[3] is True
        b = {agent: looked.action_space(agent).sample_nested() for agent in looked.agents}
        first = b[looked.agents[0]].clone()  # looking at a sample launches the policy kernel for the whole draw
        assert type(first) is torch.Tensor and looked._pending_samples[3] is True
        looked.step(b)
        acts = plain.random_policy_actions(plain.policy_seed, t).clone()
        plain.step(acts)
        if live:
            for i, agent in enumerate(fused.agents):
                assert torch.equal(a[agent].as_subclass(torch.Tensor), acts[i]), f'step {t}: samples drawn inside the step launch, {agent}'
                assert torch.equal(b[agent].as_subclass(torch.Tensor), acts[i]), f'step {t}: samples drawn by the policy launch, {agent}'
        for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards', '_task_offsets', '_act_map_offsets'):
            assert torch.equal(getattr(fused, name), getattr(plain, name)) and torch.equal(getattr(looked, name), getattr(plain, name)), f'{name} at step {t}'
    assert bool(fused.finished.all())
    fused.check()


@pytest.mark.parametrize('rng,shape', [('philox', None), ('mt19937', None), ('philox', (1, 7, 3)), ('mt19937', (3, 4, 5))],
                         ids=['philox', 'mt19937', 'philox_runtime_1x7a3', 'mt19937_runtime_3x4a5'])
def test_deferred_steps_of_the_reference_shaped_loop_equal_step_by_step(rng, shape, monkeypatch):
    """With the device declared exclusive the reference-shaped random loop only COUNTS its steps and runs them in chunks — one multi-step
    launch per chunk (utils/env.py: deferred steps).  Whatever is looked at, whenever, must be what a step-by-step execution leaves: a twin env
    that launches every step is stepped alongside, and at seeded-random points one of the things a caller can look at is compared — the
    returned dicts, public attributes, the state object taken BEFORE the steps, spaces, an old sample — then everything at every episode end."""
    import random
    monkeypatch.setenv('FRZ_WF_MULTI_STEP', 'all')
    from free_range_zoo_amd.envs import wildfire_v0
    from free_range_zoo_amd.utils.env import EnvTensor
    from test_hip_wildfire import compare_snapshots, hip_snapshot
    B, horizon = 3001, 30
    # (shape: a grid without an exact kernel instantiation — round 4: those count their steps too, <8, 4> and <16, 8> here)
    build = configs.wildfire_openness if shape is None else (lambda: configs.wildfire_grid(*shape))
    lazy, eager = [wildfire_v0.parallel_env(configuration=build(), parallel_envs=B, max_steps=horizon, device=torch.device('cuda'), rng=rng) for _ in range(2)]
    assert lazy.set_exclusive_device(True) and lazy._defer_chunk > 0 and eager._defer_chunk == 0
    lazy._deferred_log = log = []
    picker = random.Random(7)
    held_state = None
    steps_taken = 0
    for episode in range(3):
        seeds = torch.arange(B, dtype=torch.int32) * 3 + episode
        for env in (lazy, eager):
            env.reset(seed=seeds)
        if held_state is None:
            held_state = (lazy.state(), eager.state())  # views: they show the current state whenever they are looked at
        for t in range(horizon + 4):  # past the horizon: frozen steps are counted like any other
            a = {agent: lazy.action_space(agent).sample_nested() for agent in lazy.agents}
            out_l = lazy.step(a)
            out_e = eager.step({agent: eager.action_space(agent).sample_nested() for agent in eager.agents})
            steps_taken += 1
            agent = lazy.agents[picker.randrange(len(lazy.agents))]
            peek = picker.randrange(14)
            if peek == 0:
                assert type(out_l[1][agent]) is EnvTensor and torch.equal(out_l[1][agent], out_e[1][agent]), f'rewards at {episode}/{t}'
            elif peek == 1:
                assert torch.equal(lazy.num_moves, eager.num_moves) and int(lazy.num_moves.max()) == min(t + 1, horizon)
            elif peek == 2:
                assert bool(lazy.finished.all()) == bool(eager.finished.all())
            elif peek == 3:
                assert torch.equal(out_l[0][agent]['self'], out_e[0][agent]['self']) and torch.equal(out_l[0][agent]['tasks'].values(), out_e[0][agent]['tasks'].values())
            elif peek == 4:
                assert torch.equal(held_state[0].fires, held_state[1].fires) and torch.equal(held_state[0].suppressants, held_state[1].suppressants)
            elif peek == 5:
                assert lazy.environment_task_count.tolist() == eager.environment_task_count.tolist()
            elif peek == 6:
                assert torch.equal(a[agent].clone(), eager._sampled_actions[lazy.agents.index(agent)]), 'the sample a counted step drew'
            elif peek == 7:
                assert torch.equal(out_l[4]['burnouts'], out_e[4]['burnouts']) and torch.equal(out_l[2][agent], out_e[2][agent])
            elif peek == 8:
                assert lazy.action_space(agent).spaces[:40] == eager.action_space(agent).spaces[:40]
            elif peek == 9:
                assert torch.equal(lazy.agent_action_mapping[agent].values(), eager.agent_action_mapping[agent].values())
            elif peek == 10:
                assert float(lazy._cumulative_rewards[agent].sum()) == float(eager._cumulative_rewards[agent].sum())
            # (11-13: nobody looks: the chunk grows)
        compare_snapshots(hip_snapshot(lazy), hip_snapshot(eager), f'{rng}: end of episode {episode}')
        assert torch.equal(lazy.seeds, eager.seeds)
    lazy.check()
    assert sum(log) == steps_taken and max(log) > 2, f'chunks launched: {log}'


def test_reference_shaped_loop_with_mt19937_streams_across_resets():
    """The default RNG mode: the step launch of the untouched-samples path advances the env's own MT19937 streams, which a reset with new
    seeds restarts — also on the second and third episode (the streams used to be expanded only once per env object)."""
    from free_range_zoo_amd.envs import wildfire_v0
    B = 777
    fused, plain = [wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=12, device=torch.device('cuda'))
                    for _ in range(2)]
    for episode in range(3):
        seeds = torch.arange(B, dtype=torch.int32) + 100 * episode
        fused.reset(seed=seeds), plain.reset(seed=seeds)
        for t in range(12):
            fused.step({agent: fused.action_space(agent).sample_nested() for agent in fused.agents})
            plain.step(plain.random_policy_actions(plain.policy_seed, 12 * episode + t).clone())
            for name in ('_fires', '_intensity', '_fuel', '_suppressants', '_rewards', '_task_offsets'):
                assert torch.equal(getattr(fused, name), getattr(plain, name)), f'{name} at episode {episode} step {t}'
    fused.check()


@pytest.mark.parametrize('domain', ['cybersecurity', 'rideshare'])
def test_reference_shaped_random_rollout_in_the_other_domains(domain):
    """The same loop in cybersecurity and rideshare: untouched `sample_nested()` results are drawn inside the step launch (one launch per
    step instead of a policy launch + a step launch) with the values the policy launch would have produced."""
    from free_range_zoo_amd.utils.env import LazySample
    module, build, kwargs = _domains()[domain]
    B = 2049
    fused, plain = [module.parallel_env(configuration=build(), parallel_envs=B, max_steps=9, device=torch.device('cuda'), **kwargs) for _ in range(2)]
    launches = fused._deferred_log = []
    for env in (fused, plain):
        env.reset(seed=torch.arange(B, dtype=torch.int32) + 3)
    names = {'cybersecurity': ('_network_state', '_location', '_presence', '_rewards', '_act_map_offsets', '_tasks'),
             'rideshare': ('_rewards', '_task_offsets', '_agent_offsets', '_obs_self')}[domain]
    for t in range(11):  # past the horizon: the last steps are frozen
        a = {agent: fused.action_space(agent).sample_nested() for agent in fused.agents}
        assert all(type(v) is LazySample for v in a.values())
        live = not bool(fused.finished.all())
        fused.step(a)
        acts = plain.random_policy_actions(plain.policy_seed, t).clone()
        plain.step(acts)
        if live:
            for i, agent in enumerate(fused.agents):
                assert torch.equal(a[agent].as_subclass(torch.Tensor), acts[i]), f'{domain} step {t}: samples drawn inside the step launch, {agent}'
        for name in names:
            assert torch.equal(getattr(fused, name), getattr(plain, name)), f'{domain}: {name} at step {t}'
    assert launches == [1] * 11, 'one fused launch per step'
    fused.check()


def test_deferred_steps_in_cybersecurity():
    """Counted steps (utils/env.py) for cybersecurity: the reference-shaped loop on an exclusive device runs in multi-step launches and
    whatever is looked at equals the step-by-step twin."""
    import random
    from free_range_zoo_amd.envs import cybersecurity_v0
    import test_hip_cybersecurity as C
    B, horizon = 2500, 20
    lazy, eager = [cybersecurity_v0.parallel_env(configuration=configs.cyber_openness(), parallel_envs=B, max_steps=horizon, device=torch.device('cuda'),
                                                 rng='philox') for _ in range(2)]
    assert lazy.set_exclusive_device(True) and lazy._defer_chunk > 0 and eager._defer_chunk == 0
    lazy._deferred_log = log = []
    picker = random.Random(3)
    steps_taken = 0
    for episode in range(2):
        seeds = torch.arange(B, dtype=torch.int32) * 5 + episode
        lazy.reset(seed=seeds), eager.reset(seed=seeds)
        for t in range(horizon + 3):
            out_l = lazy.step({agent: lazy.action_space(agent).sample_nested() for agent in lazy.agents})
            out_e = eager.step({agent: eager.action_space(agent).sample_nested() for agent in eager.agents})
            steps_taken += 1
            agent = lazy.agents[picker.randrange(len(lazy.agents))]
            peek = picker.randrange(9)
            if peek == 0:
                assert torch.equal(out_l[1][agent], out_e[1][agent]), f'rewards at {episode}/{t}'
            elif peek == 1:
                assert torch.equal(lazy.num_moves, eager.num_moves)
            elif peek == 2:
                assert torch.equal(out_l[0][agent]['tasks'], out_e[0][agent]['tasks']) and torch.equal(out_l[0][agent]['self'], out_e[0][agent]['self'])
            elif peek == 3:
                assert torch.equal(lazy.state().network_state, eager.state().network_state)
            elif peek == 4:
                assert lazy.action_space(agent).spaces[:30] == eager.action_space(agent).spaces[:30]
        C.compare_snapshots(C.hip_snapshot(lazy), C.hip_snapshot(eager), f'end of episode {episode}')
    lazy.check()
    assert sum(log) == steps_taken and max(log) > 2, f'chunks launched: {log}'


def test_observations_of_an_earlier_step_refuse_to_fill_late():
    """The dict step() returns stands for THAT step: filled after the next step it would hold the next step's observations — it
    raises instead; looked at in time (or copied) it keeps working."""
    from free_range_zoo_amd.envs import wildfire_v0
    B = 33
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=10, device=torch.device('cuda'), rng='philox')
    obs0, _ = env.reset(seed=torch.arange(B, dtype=torch.int32))
    obs1 = env.step_random_policy(policy_seed=1, policy_step=0)[0]
    held = obs1.copy()  # filled in time
    obs2 = env.step_random_policy(policy_seed=1, policy_step=1)[0]
    with pytest.raises(RuntimeError, match='earlier step'):
        obs0[env.agents[0]]  # never looked at before the env moved on
    assert set(held) == set(env.agents) and set(obs2.keys()) == set(env.agents)
    assert obs1[env.agents[0]]['self'].shape == (B, 4) and obs2[env.agents[0]]['self'].shape == (B, 4)


def test_seed_increments_wrap_modulo_2_to_the_32():
    """frz_wildfire_reset_reseed / the reset folded into a rollout add the episode stride in unsigned arithmetic: a graph replayed
    for days passes 2^31."""
    from free_range_zoo_amd.envs import wildfire_v0
    B = 300
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=10, device=torch.device('cuda'), rng='philox')
    env.set_exclusive_device(True)
    start = torch.full((B, ), 2**31 - 5, dtype=torch.int64)
    env.reset(seed=start.to(torch.int32))
    env.rollout(3, policy_seed=1, reset_first=True, seed_increment=1000003)
    want = ((start + 1000003) % 2**32).to(torch.int64)
    assert torch.equal(env.seeds.cpu().to(torch.int64) % 2**32, want)
