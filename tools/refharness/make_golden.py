"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference in /root/reference.

Runs only in the build container (the reference cannot travel).  Usage:  python tools/refharness/make_golden.py [part ...]
Parts:
  ka        known-answer transition cases: the reference's own unit tests
            (tests/free_range_zoo/envs/*/env/transitions/test_*.py) are executed with a recording hook on every
            transition module's forward(); inputs and outputs of the calls made by PASSING tests are stored as data.
  traj      trajectories of the reference envs (reset + N steps) with recorded actions and injected randomness
            (the reference's own cross-device convention: randomness is an input).
  misc      torch-only vectors: conv2d accumulation order, torch CPU generator (MT19937) float stream.
  partial   reference runs of reset_batches(batch_indices, seed) in the middle of an episode and of
            reset(options={'initial_state': ...}) (wildfire, cybersecurity), with the steps that follow.
  rng       the reference's RandomGenerator: buffered (buffer_size > 0, keyed) and single_seeding draws.
  spaces    the objects raw_env.action_space(agent) / observation_space(agent) of the reference hand out at the reset and after every
            step of the recorded trajectories (the same runs, replayed from the same seeds), as structure: kind, n, start, low, high,
            members — what the reference's builders (envs/*/env/spaces/*.py) pass to free_range_rust's constructors.

Fixtures hold DATA only (inputs / expected outputs / the plain-C configuration fields), never reference source.
"""
import inspect
import json
import os
import sys
import unittest

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import shim  # noqa: E402

shim.install()
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLDEN, exist_ok=True)


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().copy()
    return np.asarray(x)


def _state_arrays(state, prefix):
    out = {}
    for name, value in vars(state).items():
        if isinstance(value, torch.Tensor):
            out[f'{prefix}{name}'] = _np(value)
    return out


# ----------------------------------------------------------------------------------------------------------
# known-answer cases recorded from the reference's own transition tests
# ----------------------------------------------------------------------------------------------------------
def record_known_answers():
    import importlib
    from torch import nn

    suites = {
        'wildfire': ('free_range_zoo.envs.wildfire.env.transitions',
                     ['capacity', 'equipment', 'fire_decrease', 'fire_increase', 'fire_spreads', 'suppressant_decrease',
                      'suppressant_refill']),
        'rideshare': ('free_range_zoo.envs.rideshare.env.transitions',
                      ['movement', 'passenger_entry', 'passenger_exit', 'passenger_state']),
        'cybersecurity': ('free_range_zoo.envs.cybersecurity.env.transitions', ['movement', 'presence', 'subnetwork']),
    }
    for domain, (package, modules) in suites.items():
        records = []
        current = {'test': None}
        patched = []
        for modname in modules:
            mod = importlib.import_module(f'{package}.{modname}')
            for attr in dir(mod):
                cls = getattr(mod, attr)
                if isinstance(cls, type) and issubclass(cls, nn.Module) and cls is not nn.Module and cls.__module__ == mod.__name__:
                    original = cls.forward

                    def make(cls=cls, original=original):
                        def forward(self, *args, **kwargs):
                            rec = {'cls': cls.__name__, 'test': current['test']}
                            for name, buf in self.named_buffers():
                                rec[f'buf_{name}'] = _np(buf)
                            for name in ('fast_travel', 'fire_random_spread_weight'):
                                if hasattr(self, name):
                                    rec[f'attr_{name}'] = np.asarray(getattr(self, name))
                            if hasattr(self, 'fire_spread_filter'):
                                rec['buf_fire_spread_weights'] = _np(self.fire_spread_filter.weight.data)
                            names = list(inspect.signature(inspect.unwrap(original)).parameters)[1:]
                            bound = dict(zip(names, args))
                            bound.update(kwargs)
                            for name, value in bound.items():
                                if isinstance(value, torch.Tensor):
                                    rec[f'arg_{name}'] = _np(value)
                                elif hasattr(value, 'clone') and hasattr(value, '__dict__'):
                                    rec.update(_state_arrays(value, 'in_'))
                                elif isinstance(value, (bool, int, float)):
                                    rec[f'arg_{name}'] = np.asarray(value)
                            result = original(self, *args, **kwargs)
                            outs = result if isinstance(result, tuple) else (result, )
                            extra = 0
                            for value in outs:
                                if isinstance(value, torch.Tensor):
                                    rec[f'ret_{extra}'] = _np(value)
                                    extra += 1
                                elif hasattr(value, '__dict__'):
                                    rec.update(_state_arrays(value, 'out_'))
                            records.append(rec)
                            return result
                        return forward

                    cls.forward = make()
                    patched.append((cls, original))

        class Recorder(unittest.TextTestResult):
            def startTest(self, test):
                current['test'] = test.id()
                self._mark = len(records)
                super().startTest(test)

            def _drop(self):
                del records[self._mark:]

            def addFailure(self, test, err):
                self._drop()
                super().addFailure(test, err)

            def addError(self, test, err):
                self._drop()
                super().addError(test, err)

        loader = unittest.TestLoader()
        test_dir = os.path.join(shim.REFERENCE_ROOT, 'tests', 'free_range_zoo', 'envs', domain, 'env', 'transitions')
        suite = loader.discover(test_dir, pattern='test_*.py', top_level_dir=shim.REFERENCE_ROOT)
        runner = unittest.TextTestRunner(resultclass=Recorder, verbosity=0, stream=open(os.devnull, 'w'))
        result = runner.run(suite)
        for cls, original in patched:
            cls.forward = original
        flat = {}
        meta = []
        for i, rec in enumerate(records):
            meta.append({'cls': rec['cls'], 'test': rec['test'].split('tests.free_range_zoo.')[-1]})
            for key, value in rec.items():
                if key in ('cls', 'test'):
                    continue
                flat[f'c{i}_{key}'] = value
        flat['meta'] = np.asarray(json.dumps(meta))
        path = os.path.join(GOLDEN, f'ka_{domain}.npz')
        np.savez_compressed(path, **flat)
        print(f'{path}: {len(records)} recorded calls from {result.testsRun} tests '
              f'({len(result.failures)} failures, {len(result.errors)} errors, {len(result.skipped)} skipped)')


# ----------------------------------------------------------------------------------------------------------
# trajectories
# ----------------------------------------------------------------------------------------------------------
class InjectedRandomness:
    """Replaces RandomGenerator.generate (utils/random_generator.py:86) with a recorded torch.rand source."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.log = []

    def __call__(self, parallel_envs, events, shape, key=None):
        out = torch.rand((events, parallel_envs, *shape), generator=self.gen)
        self.log.append(out.clone())
        return out


def _jagged(nt):
    return _np(nt.values()), _np(nt.offsets())


def wildfire_snapshot(env, prefix, out):
    aec = env.aec_env
    st = aec.state()
    for name in ('fires', 'intensity', 'fuel', 'suppressants', 'capacity', 'equipment'):
        out[f'{prefix}{name}'] = _np(getattr(st, name))
    out[f'{prefix}num_moves'] = _np(aec.num_moves)
    out[f'{prefix}num_burnouts'] = _np(aec.num_burnouts)
    out[f'{prefix}env_task_count'] = _np(aec.environment_task_count)
    out[f'{prefix}agent_task_count'] = _np(aec.agent_task_count)
    tv, to = _jagged(aec.task_store)
    out[f'{prefix}task_values'], out[f'{prefix}task_offsets'] = tv, to
    for a, agent in enumerate(aec.agents):
        v, o = _jagged(aec.agent_action_mapping[agent])
        out[f'{prefix}act_map_values_{a}'], out[f'{prefix}act_map_offsets_{a}'] = v, o
        v, o = _jagged(aec.agent_observation_mapping[agent])
        out[f'{prefix}obs_map_values_{a}'], out[f'{prefix}obs_map_offsets_{a}'] = v, o
        if aec.show_bad_actions:
            v, o = _jagged(aec.agent_bad_actions[agent])
            out[f'{prefix}bad_map_values_{a}'], out[f'{prefix}bad_map_offsets_{a}'] = v, o
        obs = aec.observe(agent)
        out[f'{prefix}obs_self_{a}'] = _np(obs['self'])
        out[f'{prefix}obs_others_{a}'] = _np(obs['others'])
        out[f'{prefix}cumulative_rewards_{a}'] = _np(aec._cumulative_rewards[agent])


def wildfire_policy(aec, rng, p_noop=0.25):
    """Uniform valid random actions from the recorded counts: [task index, 0] or noop [n, -1]."""
    A, B = len(aec.agents), aec.parallel_envs
    actions = np.zeros((A, B, 2), np.int32)
    for a in range(A):
        counts = _np(aec.environment_task_count if aec.show_bad_actions else aec.agent_task_count[a]).astype(np.int64)
        for b in range(B):
            n = int(counts[b])
            if n == 0 or rng.random() < p_noop:
                actions[a, b] = (n, -1)
            else:
                actions[a, b] = (rng.integers(0, n), 0)
    return actions


def wildfire_variants():
    from dataclasses import replace
    from tests.utils import wildfire_configs
    from free_range_zoo.envs.wildfire.configs.aaai_2024 import aaai_2025_ol_config
    from free_range_zoo.envs.wildfire.env.structures import configuration as C

    def openness(base_level=2, **stoch_over):
        """cfg2 of SURVEY.md §8d: non_stochastic() layout (3 agents, 2x3) + the AAAI openness stochastic block."""
        base = wildfire_configs.non_stochastic()
        aaai = aaai_2025_ol_config(base_level)
        fire = replace(aaai.fire_config, lit=base.fire_config.lit.clone())
        agent = replace(base.agent_config,
                        suppressant_decrease_probability=aaai.agent_config.suppressant_decrease_probability,
                        suppressant_refill_probability=aaai.agent_config.suppressant_refill_probability)
        stoch = replace(aaai.stochastic_config, **stoch_over)
        return C.WildfireConfiguration(grid_width=3, grid_height=2, fire_config=fire, agent_config=agent,
                                       reward_config=base.reward_config, stochastic_config=stoch)

    def rich():
        """4x5 grid, 4 agents, every stochastic switch on, non-trivial equipment/capacity tables, shaped rewards."""
        g = torch.Generator().manual_seed(7)
        H, W = 4, 5
        types = torch.randint(0, 4, (H, W), generator=g, dtype=torch.int32)
        lit = (torch.rand((H, W), generator=g) < 0.35) & (types > 0)
        fire = C.FireConfiguration(
            fire_types=types, num_fire_states=6, lit=lit, intensity_increase_probability=0.7,
            intensity_decrease_probability=0.6, extra_power_decrease_bonus=0.17, burnout_probability=0.3,
            base_spread_rate=30.0, max_spread_rate=67.0, random_ignition_probability=0.02, cell_size=200.0,
            wind_direction=1.1, ignition_temp=torch.randint(1, 4, (H, W), generator=g, dtype=torch.int32), initial_fuel=2)
        agent = C.AgentConfiguration(
            agents=torch.tensor([[0, 0], [1, 3], [3, 4], [2, 1]], dtype=torch.int32),
            fire_reduction_power=torch.tensor([1.0, 1.5, 0.75, 2.0], dtype=torch.float32),
            attack_range=torch.tensor([1, 2, 1, 1], dtype=torch.int32), suppressant_states=4, initial_suppressant=2,
            suppressant_decrease_probability=0.6, suppressant_refill_probability=0.5, initial_equipment_state=2,
            equipment_states=torch.tensor([[-1.0, -0.5, -1.0], [0.0, 0.0, 0.0], [1.0, 0.25, 1.0]], dtype=torch.float32),
            repair_probability=0.4, degrade_probability=0.3, critical_error_probability=0.1, initial_capacity=2,
            tank_switch_probability=0.5, possible_capacities=torch.tensor([1, 2, 3], dtype=torch.float32),
            capacity_probabilities=torch.tensor([0.25, 0.5, 0.25], dtype=torch.float32))
        reward = C.RewardConfiguration(fire_rewards=torch.rand((H, W), generator=g) * 40 + 5, bad_attack_penalty=-3.5,
                                       burnout_penalty=0.0, burnout_penalty_scaled=True, termination_reward=25.0,
                                       termination_kappa=4.0, localize_putouts=True)
        stoch = C.StochasticConfiguration(special_burnout_probability=True, suppressant_refill=True, suppressant_decrease=True,
                                          tank_switch=True, critical_error=True, degrade=True, repair=True, fire_increase=True,
                                          fire_decrease=True, fire_spread=True, realistic_fire_spread=True,
                                          random_fire_ignition=True, fire_fuel=True)
        return C.WildfireConfiguration(grid_width=W, grid_height=H, fire_config=fire, agent_config=agent, reward_config=reward,
                                       stochastic_config=stoch)

    def rich_plain():
        cfg = rich()
        cfg.reward_config = replace(cfg.reward_config, localize_putouts=False, burnout_penalty_scaled=False, burnout_penalty=-2.5)
        cfg.stochastic_config = replace(cfg.stochastic_config, fire_fuel=False, special_burnout_probability=False)
        return cfg

    def grid(H, Wd, A, seed=11):
        """tests/configs.py:wildfire_grid built with the reference's classes: a fully stochastic H x Wd grid with A agents — the shapes the
        cells-across-lanes kernels serve (8 x 8: one cell per lane, 16 x 16: four), recorded from the reference itself."""
        g = torch.Generator().manual_seed(seed)
        types = torch.randint(0, 4, (H, Wd), generator=g, dtype=torch.int32)
        lit = (torch.rand((H, Wd), generator=g) < 0.4) & (types > 0)
        fire = C.FireConfiguration(
            fire_types=types, num_fire_states=5, lit=lit, intensity_increase_probability=0.6, intensity_decrease_probability=0.7,
            extra_power_decrease_bonus=0.1, burnout_probability=0.4, base_spread_rate=30.0, max_spread_rate=67.0,
            random_ignition_probability=0.01, cell_size=200.0, wind_direction=0.7,
            ignition_temp=torch.randint(1, 3, (H, Wd), generator=g, dtype=torch.int32), initial_fuel=2)
        agents = torch.stack([torch.randint(0, H, (A, ), generator=g), torch.randint(0, Wd, (A, ), generator=g)], dim=1).to(torch.int32)
        agent = C.AgentConfiguration(
            agents=agents, fire_reduction_power=torch.rand((A, ), generator=g) + 0.5,
            attack_range=torch.randint(1, 3, (A, ), generator=g, dtype=torch.int32), suppressant_states=4, initial_suppressant=2,
            suppressant_decrease_probability=0.7, suppressant_refill_probability=0.6, initial_equipment_state=1,
            equipment_states=torch.tensor([[-1.0, -0.5, -1.0], [0.0, 0.0, 0.0], [1.0, 0.25, 1.0]], dtype=torch.float32),
            repair_probability=0.5, degrade_probability=0.2, critical_error_probability=0.05, initial_capacity=2,
            tank_switch_probability=0.5, possible_capacities=torch.tensor([1, 2, 3], dtype=torch.float32),
            capacity_probabilities=torch.tensor([0.3, 0.4, 0.3], dtype=torch.float32))
        reward = C.RewardConfiguration(fire_rewards=torch.rand((H, Wd), generator=g) * 30 + 5, bad_attack_penalty=-2.0, burnout_penalty=-1.5,
                                       burnout_penalty_scaled=False, termination_reward=10.0, termination_kappa=2.0, localize_putouts=False)
        stoch = C.StochasticConfiguration(special_burnout_probability=True, suppressant_refill=True, suppressant_decrease=True,
                                          tank_switch=True, critical_error=True, degrade=True, repair=True, fire_increase=True,
                                          fire_decrease=True, fire_spread=True, realistic_fire_spread=True, random_fire_ignition=True,
                                          fire_fuel=False)
        return C.WildfireConfiguration(grid_width=Wd, grid_height=H, fire_config=fire, agent_config=agent, reward_config=reward,
                                       stochastic_config=stoch)

    return [
        # name, configuration, env kwargs, B, max_steps, steps, seed
        ('grid8x8_12agents', grid(8, 8, 12), dict(observe_other_suppressant=True), 6, 24, 26, 19),
        ('grid16x16_6agents', grid(16, 16, 6), dict(show_bad_actions=True, observe_other_power=True), 5, 16, 16, 20),
        ('grid32x32_16agents', grid(32, 32, 16), dict(observe_other_power=True, observe_other_suppressant=True), 3, 9, 10, 21),
        ('cfg1_nonstochastic', wildfire_configs.non_stochastic(), {}, 4, 15, 18, 11),
        ('cfg2_openness', openness(), {}, 16, 50, 52, 12),
        ('aaai_ol3_2agents', aaai_2025_ol_config(3), {}, 8, 30, 30, 13),
        ('openness_bad_actions', openness(), dict(show_bad_actions=True, observe_other_power=True), 8, 40, 40, 14),
        ('openness_observe_all', openness(1), dict(observe_other_power=True, observe_other_suppressant=True), 6, 25, 25, 15),
        ('rich_localized', rich(), dict(observe_other_suppressant=True), 12, 40, 44, 16),
        ('rich_plain_bad_actions', rich_plain(), dict(show_bad_actions=True), 10, 40, 40, 17),
        ('rich_no_truncation', rich(), {}, 6, None, 30, 18),
    ]


def record_wildfire_trajectories():
    from free_range_zoo.envs import wildfire_v0
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct
    from free_range_zoo_amd._capi import struct_to_dict

    for name, configuration, kwargs, B, max_steps, steps, seed in wildfire_variants():
        flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
        flags.update(kwargs)
        cstruct = to_cstruct(configuration, B, max_steps, **flags)
        env = wildfire_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration,
                                       device=torch.device('cpu'), **flags)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        source = InjectedRandomness(seed)
        env.aec_env.generator.generate = source
        rng = np.random.default_rng(seed)
        out = {'cfg': np.asarray(json.dumps(struct_to_dict(cstruct))), 'steps': np.asarray(steps)}
        wildfire_snapshot(env, 'r_', out)
        agents = list(env.aec_env.agents)
        for t in range(steps):
            actions = wildfire_policy(env.aec_env, rng)
            mark = len(source.log)
            _, rewards, terminations, truncations, infos = env.step(
                {agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
            drawn = source.log[mark:]
            p = f's{t}_'
            out[p + 'actions'] = actions
            out[p + 'stepped'] = np.asarray(len(drawn) == 2)
            if len(drawn) == 2:
                out[p + 'field_randomness'], out[p + 'agent_randomness'] = _np(drawn[0]), _np(drawn[1])
            out[p + 'rewards'] = np.stack([_np(rewards[agent]) for agent in agents])
            out[p + 'terminations'] = np.stack([_np(terminations[agent]) for agent in agents])
            out[p + 'truncations'] = np.stack([_np(truncations[agent]) for agent in agents])
            if 'burnouts' in infos:
                out[p + 'burnouts'], out[p + 'putouts'] = _np(infos['burnouts']), _np(infos['putouts'])
            out[p + 'finished'] = _np(env.finished)
            wildfire_snapshot(env, p, out)
        path = os.path.join(GOLDEN, f'traj_wildfire_{name}.npz')
        np.savez_compressed(path, **out)
        print(f'{path}: B={B} steps={steps} finished={int(_np(env.finished).sum())}/{B} '
              f'tasks_mean={float(out[f"s{steps - 1}_env_task_count"].mean()):.2f}')


# ----------------------------------------------------------------------------------------------------------
# cybersecurity trajectories
# ----------------------------------------------------------------------------------------------------------
def cyber_snapshot(env, prefix, out):
    aec = env.aec_env
    st = aec.state()
    for name in ('network_state', 'location', 'presence'):
        out[f'{prefix}{name}'] = _np(getattr(st, name))
    out[f'{prefix}num_moves'] = _np(aec.num_moves)
    out[f'{prefix}env_task_count'] = _np(aec.environment_task_count)
    out[f'{prefix}agent_task_count'] = _np(aec.agent_task_count)
    for a, agent in enumerate(aec.agents):
        v, o = _jagged(aec.agent_action_mapping[agent])
        out[f'{prefix}act_map_values_{a}'], out[f'{prefix}act_map_offsets_{a}'] = v, o
        v, o = _jagged(aec.agent_observation_mapping[agent])
        out[f'{prefix}obs_map_values_{a}'], out[f'{prefix}obs_map_offsets_{a}'] = v, o
        obs = aec.observe(agent)
        out[f'{prefix}obs_self_{a}'] = _np(obs['self'])
        out[f'{prefix}obs_others_{a}'] = _np(obs['others'])
        out[f'{prefix}obs_tasks_{a}'] = _np(obs['tasks'])
        out[f'{prefix}cumulative_rewards_{a}'] = _np(aec._cumulative_rewards[agent])


def cyber_policy(aec, rng):
    """Uniform valid actions from the same member lists the reference's action spaces hold (spaces/actions.py:11-99)."""
    A, B = len(aec.agents), aec.parallel_envs
    N = aec.network_config.num_nodes
    Att = aec.attacker_config.num_attackers
    loc = _np(aec.state().location)
    atc = _np(aec.agent_task_count)
    actions = np.zeros((A, B, 2), np.int32)
    for a in range(A):
        for b in range(B):
            n = N if aec.show_bad_actions else int(atc[a, b])
            tail = [-1]
            if a >= Att and n > 0:
                if aec.show_bad_actions or loc[b, a - Att] != -1:
                    tail.append(-2)
                tail.append(-3)
            j = int(rng.integers(0, n + len(tail)))
            actions[a, b] = (j, 0) if j < n else (j, tail[j - n])
    return actions


def cyber_variants():
    from dataclasses import replace
    from tests.utils import cybersecurity_configs
    from free_range_zoo.envs.cybersecurity.env.structures import configuration as C

    def openness(**stoch):
        """cfg4 of SURVEY.md §8d: non_stochastic() with agent openness on (persist 0.9 / return 0.5) + stochastic states."""
        base = cybersecurity_configs.non_stochastic()
        att = replace(base.attacker_config, persist_probs=torch.tensor([0.9, 0.9]), return_probs=torch.tensor([0.5, 0.5]))
        dfn = replace(base.defender_config, persist_probs=torch.tensor([0.9, 0.9]), return_probs=torch.tensor([0.5, 0.5]))
        return C.CybersecurityConfiguration(attacker_config=att, defender_config=dfn, network_config=base.network_config,
                                            reward_config=base.reward_config,
                                            stochastic_config=C.StochasticConfiguration(network_state=stoch.get('network_state', True)))

    def rich():
        g = torch.Generator().manual_seed(5)
        N, Att, D = 6, 3, 4
        adj = torch.rand((N, N), generator=g) < 0.5
        adj = (adj | adj.T) & ~torch.eye(N, dtype=torch.bool)
        att = C.AttackerConfiguration(initial_presence=torch.tensor([True, False, True]), threat=torch.tensor([1.0, 0.5, 1.75]),
                                      persist_probs=torch.tensor([0.8, 0.95, 0.7]), return_probs=torch.tensor([0.3, 0.6, 0.45]))
        dfn = C.DefenderConfiguration(initial_location=torch.tensor([0, -1, 3, 5], dtype=torch.int32),
                                      initial_presence=torch.tensor([True, True, False, True]),
                                      mitigation=torch.tensor([1.0, 0.75, 1.25, 0.6]), persist_probs=torch.tensor([0.85, 0.9, 0.75, 0.95]),
                                      return_probs=torch.tensor([0.4, 0.5, 0.6, 0.35]))
        net = C.NetworkConfiguration(patched_states=2, vulnerable_states=2, exploited_states=3, temperature=1.7,
                                     initial_state=torch.randint(0, 7, (N, ), generator=g, dtype=torch.int32), adj_matrix=adj)
        rew = C.RewardConfiguration(bad_action_penalty=-7.5, patch_reward=-0.25,
                                    network_state_rewards=torch.tensor([4.0, 2.5, 0.0, -1.0, -2.0, -4.5, -8.0]))
        return C.CybersecurityConfiguration(attacker_config=att, defender_config=dfn, network_config=net, reward_config=rew,
                                            stochastic_config=C.StochasticConfiguration(network_state=True))

    def grid(N, Att, D, seed=13):
        """tests/configs.py:cyber_grid built with the reference's classes: N subnetworks, Att attackers, D defenders, stochastic — many
        distinct (patches - attacks) sums, i.e. many arguments of the tanh the subnetwork transition compares its draw with."""
        g = torch.Generator().manual_seed(seed)
        adj = torch.rand((N, N), generator=g) < 0.5
        adj = (adj | adj.T) & ~torch.eye(N, dtype=torch.bool)
        att = C.AttackerConfiguration(initial_presence=torch.rand((Att, ), generator=g) < 0.6, threat=torch.rand((Att, ), generator=g) + 0.5,
                                      persist_probs=torch.rand((Att, ), generator=g) * 0.3 + 0.65, return_probs=torch.rand((Att, ), generator=g) * 0.4 + 0.3)
        dfn = C.DefenderConfiguration(initial_location=torch.randint(-1, N, (D, ), generator=g, dtype=torch.int32),
                                      initial_presence=torch.rand((D, ), generator=g) < 0.7, mitigation=torch.rand((D, ), generator=g) + 0.5,
                                      persist_probs=torch.rand((D, ), generator=g) * 0.3 + 0.65, return_probs=torch.rand((D, ), generator=g) * 0.4 + 0.3)
        net = C.NetworkConfiguration(patched_states=1, vulnerable_states=2, exploited_states=2, temperature=2.3,
                                     initial_state=torch.randint(0, 5, (N, ), generator=g, dtype=torch.int32), adj_matrix=adj)
        rew = C.RewardConfiguration(bad_action_penalty=-5.0, patch_reward=-0.5, network_state_rewards=torch.tensor([3.0, 1.0, 0.0, -2.0, -5.0]))
        return C.CybersecurityConfiguration(attacker_config=att, defender_config=dfn, network_config=net, reward_config=rew,
                                            stochastic_config=C.StochasticConfiguration(network_state=True))

    return [
        ('grid8_4x4', grid(8, 4, 4), dict(observe_other_presence=True), 24, 40, 40, 26),
        ('nonstochastic', cybersecurity_configs.non_stochastic(), {}, 4, 15, 18, 21),
        ('cfg4_openness', openness(), {}, 16, 50, 52, 22),
        ('openness_no_bad_actions', openness(), dict(show_bad_actions=False, observe_other_presence=True, observe_other_location=True), 8,
         30, 30, 23),
        ('rich', rich(), dict(partially_observable=True, observe_other_location=True), 12, 40, 42, 24),
        ('rich_fully_observable', rich(), dict(partially_observable=False, observe_other_power=False, observe_other_presence=True,
                                               show_bad_actions=False), 10, None, 30, 25),
    ]


def record_cyber_trajectories():
    from free_range_zoo.envs import cybersecurity_v0
    from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct
    from free_range_zoo_amd._capi import struct_to_dict

    for name, configuration, kwargs, B, max_steps, steps, seed in cyber_variants():
        flags = dict(observe_other_location=False, observe_other_presence=False, observe_other_power=True, partially_observable=True,
                     show_bad_actions=True)
        flags.update(kwargs)
        cstruct = to_cstruct(configuration, B, max_steps, **flags)
        env = cybersecurity_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'),
                                            **flags)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        source = InjectedRandomness(seed)
        env.aec_env.generator.generate = source
        rng = np.random.default_rng(seed)
        out = {'cfg': np.asarray(json.dumps(struct_to_dict(cstruct))), 'steps': np.asarray(steps)}
        cyber_snapshot(env, 'r_', out)
        agents = list(env.aec_env.agents)
        for t in range(steps):
            actions = cyber_policy(env.aec_env, rng)
            mark = len(source.log)
            _, rewards, terminations, truncations, infos = env.step(
                {agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
            drawn = source.log[mark:]
            p = f's{t}_'
            out[p + 'actions'] = actions
            out[p + 'stepped'] = np.asarray(len(drawn) == 2)
            if len(drawn) == 2:
                out[p + 'network_randomness'], out[p + 'agent_randomness'] = _np(drawn[0]), _np(drawn[1])
            out[p + 'rewards'] = np.stack([_np(rewards[agent]) for agent in agents])
            out[p + 'terminations'] = np.stack([_np(terminations[agent]) for agent in agents])
            out[p + 'truncations'] = np.stack([_np(truncations[agent]) for agent in agents])
            out[p + 'finished'] = _np(env.finished)
            cyber_snapshot(env, p, out)
        path = os.path.join(GOLDEN, f'traj_cybersecurity_{name}.npz')
        np.savez_compressed(path, **out)
        print(f'{path}: B={B} steps={steps} finished={int(_np(env.finished).sum())}/{B} '
              f'present_mean={float(out[f"s{steps - 1}_presence"].mean()):.2f}')


# ----------------------------------------------------------------------------------------------------------
# rideshare trajectories
# ----------------------------------------------------------------------------------------------------------
def rideshare_snapshot(env, prefix, out):
    aec = env.aec_env
    st = aec.state()
    out[f'{prefix}agents'] = _np(st.agents)
    out[f'{prefix}passengers'] = _np(st.passengers)
    out[f'{prefix}num_moves'] = _np(aec.num_moves)
    out[f'{prefix}env_task_count'] = _np(aec.environment_task_count)
    out[f'{prefix}agent_task_count'] = _np(aec.agent_task_count)
    tv, to = _jagged(aec.task_store)
    out[f'{prefix}task_values'], out[f'{prefix}task_offsets'] = tv, to
    for a, agent in enumerate(aec.agents):
        v, o = _jagged(aec.agent_action_mapping[agent])
        out[f'{prefix}act_map_values_{a}'], out[f'{prefix}act_map_offsets_{a}'] = v, o
        v, o = _jagged(aec.agent_observation_mapping[agent])
        out[f'{prefix}obs_map_values_{a}'], out[f'{prefix}obs_map_offsets_{a}'] = v, o
        obs = aec.observe(agent)
        out[f'{prefix}obs_self_{a}'] = _np(obs['self'])
        out[f'{prefix}obs_others_{a}'] = _np(obs['others'])
        v, o = _jagged(obs['tasks'])
        out[f'{prefix}obs_tasks_values_{a}'], out[f'{prefix}obs_tasks_offsets_{a}'] = v, o
        out[f'{prefix}cumulative_rewards_{a}'] = _np(aec._cumulative_rewards[agent])


def rideshare_policy(aec, rng, p_noop=0.15, p_contest=0.35):
    """Valid actions from the members the reference's action space holds: per visible task the action id = the passenger's
    state (0 accept / 1 pick / 2 drop), or noop.  Agents are nudged towards the same unaccepted passenger now and then so
    that the accept-conflict resolution is exercised."""
    A, B = len(aec.agents), aec.parallel_envs
    passengers = _np(aec.state().passengers)
    actions = np.zeros((A, B, 2), np.int32)
    starts = np.concatenate([[0], np.cumsum(_np(aec.environment_task_count))])
    for b in range(B):
        rows = passengers[starts[b]:starts[b + 1]]
        for a in range(A):
            visible = [k for k in range(len(rows)) if rows[k, 6] == 0 or rows[k, 7] == a]
            n = len(visible)
            if n == 0 or rng.random() < p_noop:
                actions[a, b] = (n, -1)
                continue
            j = int(rng.integers(0, n))
            unaccepted = [i for i, k in enumerate(visible) if rows[k, 6] == 0]
            if unaccepted and rng.random() < p_contest:
                j = unaccepted[0]
            actions[a, b] = (j, rows[visible[j], 6])
    return actions


def rideshare_variants():
    from dataclasses import replace
    from tests.utils import rideshare_configs
    from free_range_zoo.envs.rideshare.env.structures import configuration as C

    def busy(A=8, steps=32, per_step=2, grid=10, seed=0, env_specific=0, B=1, **reward_over):
        """cfg3 of SURVEY.md §8d: `per_step` wildcard passengers per step for `steps` steps, fares 1..10, 8 agents, pool 4."""
        g = torch.Generator().manual_seed(seed)
        rows = []
        for t in range(steps):
            for _ in range(per_step):
                y, x, yd, xd = torch.randint(0, grid, (4, ), generator=g).tolist()
                rows.append([t, -1, y, x, yd, xd, int(torch.randint(1, 11, (1, ), generator=g))])
            for _ in range(env_specific):
                y, x, yd, xd = torch.randint(0, grid, (4, ), generator=g).tolist()
                rows.append([t, int(torch.randint(0, B, (1, ), generator=g)), y, x, yd, xd, int(torch.randint(1, 11, (1, ), generator=g))])
        base = rideshare_configs.non_stochastic()
        positions = [(0, 0), (grid - 1, grid - 1), (0, grid - 1), (grid - 1, 0), (0, 4), (grid - 1, 5), (4, 0), (5, grid - 1)][:A]
        agent = C.AgentConfiguration(start_positions=torch.tensor(positions), pool_limit=4, use_fast_travel=False, use_diagonal_travel=False)
        reward = replace(base.reward_config, **reward_over)
        return C.RideshareConfiguration(grid_height=grid, grid_width=grid, agent_config=agent, reward_config=reward,
                                        passenger_config=C.PassengerConfiguration(schedule=torch.tensor(rows, dtype=torch.int)))

    def small_fast(diagonal, fast):
        cfg = busy(A=3, steps=12, per_step=1, grid=5, seed=3, env_specific=1, B=6, use_waiting_costs=True, use_variable_move_cost=False,
                   wait_limit=torch.tensor([1, 2, 2]), long_wait_time=3, drop_cost=0.5, accept_cost=-0.2)
        cfg.agent_config = replace(cfg.agent_config, use_diagonal_travel=diagonal, use_fast_travel=fast, pool_limit=1)
        return cfg

    def fast_diagonal():
        """Fast travel AND diagonal moves on a 12 x 12 grid with the variable move cost: an agent jumps to its goal in one move of any
        (dy, dx), so the move cost is the square root of an arbitrary sum of squares (and is divided by the passengers on board + 1) —
        the float path a unit-move trajectory never exercises (tests/test_hip_fuzz.py found a last-bit difference there)."""
        cfg = busy(A=4, steps=14, per_step=2, grid=12, seed=5, env_specific=1, B=8)
        cfg.agent_config = replace(cfg.agent_config, use_diagonal_travel=True, use_fast_travel=True, pool_limit=2)
        return cfg

    return [
        ('nonstochastic', rideshare_configs.non_stochastic(), 4, 15, 18, 31),
        ('cfg3_busy', busy(), 12, 50, 52, 32),
        ('busy_waiting_costs', busy(A=4, steps=20, per_step=2, seed=1, use_waiting_costs=True, wait_limit=torch.tensor([2, 3, 4]),
                                    long_wait_time=6), 10, 30, 32, 33),
        ('small_diagonal', small_fast(True, False), 6, 20, 22, 34),
        ('small_fast_travel', small_fast(False, True), 6, None, 18, 35),
        ('fast_diagonal', fast_diagonal(), 8, 16, 18, 36),
    ]


def record_rideshare_trajectories():
    from free_range_zoo.envs import rideshare_v0
    from free_range_zoo_amd.envs.rideshare.env.structures.configuration import to_cstruct
    from free_range_zoo_amd._capi import struct_to_dict

    for name, configuration, B, max_steps, steps, seed in rideshare_variants():
        cstruct, schedule = to_cstruct(configuration, B, max_steps)
        env = rideshare_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'))
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        rng = np.random.default_rng(seed)
        out = {'cfg': np.asarray(json.dumps(struct_to_dict(cstruct))), 'schedule': schedule, 'steps': np.asarray(steps)}
        rideshare_snapshot(env, 'r_', out)
        agents = list(env.aec_env.agents)
        peak = 0
        for t in range(steps):
            actions = rideshare_policy(env.aec_env, rng)
            moves_before = _np(env.aec_env.num_moves).copy()
            _, rewards, terminations, truncations, infos = env.step(
                {agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
            p = f's{t}_'
            out[p + 'actions'] = actions
            out[p + 'stepped'] = np.asarray(bool((_np(env.aec_env.num_moves) != moves_before).any()))
            out[p + 'rewards'] = np.stack([_np(rewards[agent]) for agent in agents])
            out[p + 'terminations'] = np.stack([_np(terminations[agent]) for agent in agents])
            out[p + 'truncations'] = np.stack([_np(truncations[agent]) for agent in agents])
            out[p + 'finished'] = _np(env.finished)
            rideshare_snapshot(env, p, out)
            peak = max(peak, int(out[p + 'env_task_count'].max()))
        path = os.path.join(GOLDEN, f'traj_rideshare_{name}.npz')
        np.savez_compressed(path, **out)
        print(f'{path}: B={B} steps={steps} peak passengers/env={peak} slots={cstruct.max_passengers}')


# ----------------------------------------------------------------------------------------------------------
# torch-only vectors
# ----------------------------------------------------------------------------------------------------------
def record_misc():
    # torch CPU generator float32 stream (what RandomGenerator draws per env), seeds as the reference uses them
    seeds = np.array([0, 1, 7, 12345, 99999999, 4357, 5489], np.int32)
    n = 2000
    draws = np.zeros((len(seeds), n), np.float32)
    for i, s in enumerate(seeds):
        g = torch.Generator().manual_seed(int(s))
        # two consecutive generate()-style draws from one stream, like wildfire.py:409-410
        draws[i, :18] = torch.rand((3, 2, 3), generator=g).reshape(-1).numpy()
        draws[i, 18:33] = torch.rand((5, 3), generator=g).reshape(-1).numpy()
        draws[i, 33:] = torch.rand(n - 33, generator=g).numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'mt19937_torch.npz'), seeds=seeds, draws=draws)

    # conv2d(1->1, 3x3, pad 1) accumulation order on 0/1 inputs with cross filters
    gen = torch.Generator().manual_seed(3)
    cases = {}
    for i, (H, W, B) in enumerate([(2, 3, 5), (4, 5, 7), (8, 8, 3), (1, 6, 4), (6, 1, 4)]):
        w = torch.zeros(3, 3)
        w[0, 1], w[1, 0], w[1, 2], w[2, 1] = torch.rand(4, generator=gen)
        lit = (torch.rand((B, 1, H, W), generator=gen) < 0.55).float()
        out = torch.nn.functional.conv2d(lit, w.reshape(1, 1, 3, 3), padding=1)[:, 0]
        cases[f'w{i}'], cases[f'lit{i}'], cases[f'out{i}'] = w.numpy(), lit[:, 0].numpy(), out.numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'conv_order.npz'), **cases)
    print('misc vectors written')


# ----------------------------------------------------------------------------------------------------------
# scripted wildfire baselines: the reference's agents on observations of the reference env
# ----------------------------------------------------------------------------------------------------------
def record_wildfire_baselines():
    """Runs free_range_zoo.envs.wildfire.baselines.{Strongest,Weakest}Baseline.observe() on (observation, action mapping) pairs of
    the unmodified reference env (the pair is what wrappers/action_task.py:45 hands to an agent) and records inputs and answers."""
    from free_range_zoo.envs import wildfire_v0
    from free_range_zoo.envs.wildfire.baselines.strongest import StrongestBaseline
    from free_range_zoo.envs.wildfire.baselines.weakest import WeakestBaseline

    variants = {v[0]: v for v in wildfire_variants()}
    out, cases = {}, 0
    for name in ('rich_localized', 'rich_plain_bad_actions', 'cfg2_openness'):
        _, configuration, kwargs, B, max_steps, steps, seed = variants[name]
        flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
        flags.update(kwargs)
        env = wildfire_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'), **flags)
        observations, _ = env.reset(seed=torch.arange(B, dtype=torch.int32))
        env.aec_env.generator.generate = InjectedRandomness(seed + 100)
        rng = np.random.default_rng(seed + 100)
        agents = list(env.aec_env.agents)
        bots = {agent: (StrongestBaseline(agent, B), WeakestBaseline(agent, B)) for agent in agents}
        for t in range(min(steps, 14)):
            for agent in agents:
                mapping = env.aec_env.agent_action_mapping[agent]
                pair = (observations[agent], {'agent_action_mapping': mapping})
                tasks = observations[agent]['tasks']
                rows = [r for r in tasks.unbind()]
                p = f'c{cases}_'
                out[p + 'task_values'] = np.concatenate([_np(r).reshape(-1, 4) for r in rows]).astype(np.int64) if rows else np.zeros((0, 4), np.int64)
                out[p + 'task_counts'] = np.asarray([r.shape[0] for r in rows], np.int64)
                out[p + 'map_lengths'] = np.asarray([m.shape[0] for m in mapping.unbind()], np.int64)
                out[p + 'obs_self'] = _np(observations[agent]['self']).astype(np.float32)
                for kind, bot in zip(('strongest', 'weakest'), bots[agent]):
                    bot.observe(pair)
                    out[p + kind] = _np(bot.act(None)).astype(np.int32).copy()
                cases += 1
            actions = wildfire_policy(env.aec_env, rng)
            observations, *_ = env.step({agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
    out['cases'] = np.asarray(cases)
    path = os.path.join(GOLDEN, 'baselines_wildfire.npz')
    np.savez_compressed(path, **out)
    print(f'{path}: {cases} observe() calls')



# ----------------------------------------------------------------------------------------------------------
# scripted rideshare baselines: greedy / FIFO, task-focused / task-global
# ----------------------------------------------------------------------------------------------------------
RIDESHARE_BOTS = ('greedy_focus', 'greedy_global', 'fifo_focus', 'fifo_global')


def record_rideshare_baselines():
    """Runs the reference's four scripted rideshare agents (envs/rideshare/baselines/{greedy,fifo}_T{focus,global}.py) on
    (observation, action mapping) pairs of the unmodified reference env.  Their tie-breaks come from torch.randint on the global
    generator; the call is wrapped so that each decision's number of tied candidates and the member drawn are recorded next to
    the answer, which makes the tied answers checkable too.  States come from two drivers: the recorded random policy (agents
    hold several passengers at once) and the focused agents driving every car themselves (one passenger at a time)."""
    from free_range_zoo.envs import rideshare_v0
    from free_range_zoo.envs.rideshare.baselines.greedy_Tfocus import GreedyTaskFocus
    from free_range_zoo.envs.rideshare.baselines.greedy_Tglobal import GreedyTaskGlobal
    from free_range_zoo.envs.rideshare.baselines.fifo_Tfocus import FirstInFirstOutTfocusBaseline
    from free_range_zoo.envs.rideshare.baselines.fifo_Tglobal import FirstInFirstOutTglobalBaseline

    classes = dict(zip(RIDESHARE_BOTS, (GreedyTaskFocus, GreedyTaskGlobal, FirstInFirstOutTfocusBaseline, FirstInFirstOutTglobalBaseline)))
    variants = {v[0]: v for v in rideshare_variants()}
    draws = []
    original_randint = torch.randint

    def recording_randint(low, high, size, **kwargs):
        value = original_randint(low, high, size, **kwargs)
        draws.append((int(high), int(value)))
        return value

    def make(kind, agent, B, configuration):
        kwargs = dict(agent_configuration=configuration.agent_config) if kind.startswith('greedy') else {}
        return classes[kind](agent, B, **kwargs)

    out, cases = {}, 0
    for name in ('nonstochastic', 'cfg3_busy', 'busy_waiting_costs', 'small_diagonal', 'small_fast_travel'):
        _, configuration, B, max_steps, steps, seed = variants[name]
        diagonal = int(bool(configuration.agent_config.use_diagonal_travel))
        for driver in ('random', 'focused'):
            env = rideshare_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'))
            observations, _ = env.reset(seed=torch.arange(B, dtype=torch.int32))
            aec = env.aec_env
            agents = list(aec.agents)
            rng = np.random.default_rng(seed + 200)
            torch.manual_seed(seed)
            drivers = {agent: make(RIDESHARE_BOTS[(2 * i) % 4], agent, B, configuration) for i, agent in enumerate(agents)}
            for t in range(min(steps, 14)):
                driven = np.zeros((len(agents), B, 2), np.int32)
                for a, agent in enumerate(agents):
                    mapping = aec.agent_action_mapping[agent]
                    pair = (observations[agent], {'agent_action_mapping': mapping})
                    rows = list(observations[agent]['tasks'].unbind())
                    counts = np.asarray([r.shape[0] for r in rows], np.int64)
                    if a < 3:
                        p = f'c{cases}_'
                        out[p + 'task_values'] = (np.concatenate([_np(r).reshape(-1, 8) for r in rows]).astype(np.int32)
                                                  if counts.sum() else np.zeros((0, 8), np.int32))
                        out[p + 'task_counts'] = counts
                        out[p + 'map_lengths'] = np.asarray([m.shape[0] for m in mapping.unbind()], np.int64)
                        out[p + 'obs_self'] = _np(observations[agent]['self']).astype(np.int32)
                        out[p + 'diagonal'] = np.asarray(diagonal)
                        for kind in RIDESHARE_BOTS:
                            bot = make(kind, agent, B, configuration)
                            del draws[:]
                            torch.randint = recording_randint
                            try:
                                bot.observe(pair)
                                valid = 1
                            except AssertionError:  # greedy_Tfocus refuses observations holding several accepted / riding passengers
                                valid = 0
                            finally:
                                torch.randint = original_randint
                            out[p + kind + '_valid'] = np.asarray(valid)
                            if valid:
                                answer = _np(bot.act(None)).astype(np.int32).copy()
                                decided = np.flatnonzero(answer[:, 0] >= 0)
                                assert len(decided) == len(draws), (kind, len(decided), len(draws))
                                ties, picks = np.zeros(B, np.int64), np.zeros(B, np.int64)
                                ties[decided] = [d[0] for d in draws]
                                picks[decided] = [d[1] for d in draws]
                                out[p + kind], out[p + kind + '_ties'], out[p + kind + '_picks'] = answer, ties, picks
                        cases += 1
                    if driver == 'focused':
                        try:
                            drivers[agent].observe(pair)
                            answer = _np(drivers[agent].act(None)).astype(np.int32).copy()
                        except AssertionError:
                            answer = np.full((B, 2), -1, np.int32)
                        noop = answer[:, 1] < 0
                        answer[noop, 0] = counts[noop]
                        driven[a] = answer
                actions = rideshare_policy(aec, rng) if driver == 'random' else driven
                observations, *_ = env.step({agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
    out['cases'] = np.asarray(cases)
    path = os.path.join(GOLDEN, 'baselines_rideshare.npz')
    np.savez_compressed(path, **out)
    print(f'{path}: {cases} observe() calls x {len(RIDESHARE_BOTS)} agents')



# ----------------------------------------------------------------------------------------------------------
# scripted cybersecurity baselines: patched / exploited attackers and defenders
# ----------------------------------------------------------------------------------------------------------
CYBER_BOTS = {'attacker': ('patched_attacker', 'exploited_attacker'), 'defender': ('patched_defender', 'exploited_defender')}


def record_cyber_baselines():
    """Runs the reference's stateful agents (envs/cybersecurity/baselines/{patched,exploited}.py) along trajectories of the
    unmodified reference env and records, per observe() call, the observation, the agent's state before (target_node,
    time_focused, actions), the torch.randint tie-break draws, and state + answer after.  Two drivers: the recorded random
    policy and the agents driving the env themselves."""
    from free_range_zoo.envs import cybersecurity_v0
    from free_range_zoo.envs.cybersecurity.baselines.patched import PatchedAttackerBaseline, PatchedDefenderBaseline
    from free_range_zoo.envs.cybersecurity.baselines.exploited import ExploitedAttackerBaseline, ExploitedDefenderBaseline
    from free_range_zoo.envs.cybersecurity.baselines.camp import CampDefenderBaseline

    variants = {v[0]: v for v in cyber_variants()}
    draws = []
    original_randint = torch.randint

    def recording_randint(low, high, size, **kwargs):
        value = original_randint(low, high, size, **kwargs)
        draws.append((int(high), int(value)))
        return value

    out, cases = {}, 0
    for name in ('nonstochastic', 'cfg4_openness', 'rich', 'rich_fully_observable'):
        _, configuration, kwargs, B, max_steps, steps, seed = variants[name]
        flags = dict(observe_other_location=False, observe_other_presence=False, observe_other_power=True, partially_observable=True,
                     show_bad_actions=True)
        flags.update(kwargs)
        states = int(configuration.network_config.num_states) if hasattr(configuration.network_config, 'num_states') else None
        for driver in ('random', 'bots'):
            env = cybersecurity_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'),
                                                **flags)
            observations, _ = env.reset(seed=torch.arange(B, dtype=torch.int32))
            aec = env.aec_env
            aec.generator.generate = InjectedRandomness(seed + 300)
            rng = np.random.default_rng(seed + 300)
            torch.manual_seed(seed)
            agents = list(aec.agents)
            S = states if states is not None else int(aec.network_config.num_states)
            bots = {}
            camps = {agent: CampDefenderBaseline(agent, B) for agent in agents if agent.startswith('defender')}
            for agent in agents:
                if agent.startswith('attacker'):
                    bots[agent] = (PatchedAttackerBaseline(agent, B), ExploitedAttackerBaseline(S, agent, B))
                else:
                    bots[agent] = (PatchedDefenderBaseline(agent, B), ExploitedDefenderBaseline(S, agent, B))
            for t in range(min(steps, 12)):
                driven = np.zeros((len(agents), B, 2), np.int32)
                for a, agent in enumerate(agents):
                    mapping = aec.agent_action_mapping[agent]
                    pair = (observations[agent], {'agent_action_mapping': mapping})
                    p = f'c{cases}_'
                    out[p + 'tasks'] = _np(observations[agent]['tasks']).astype(np.int64)
                    out[p + 'obs_self'] = _np(observations[agent]['self']).astype(np.float32)
                    out[p + 'mapping_numel'] = np.asarray(int(mapping.numel()))
                    out[p + 'subnetwork_states'] = np.asarray(S)
                    role = 'attacker' if agent.startswith('attacker') else 'defender'
                    out[p + 'role'] = np.asarray(role)
                    for kind, bot in zip(CYBER_BOTS[role], bots[agent]):
                        out[p + kind + '_pre'] = np.stack([_np(bot.target_node), _np(bot.time_focused), *_np(bot.actions).T]).astype(np.int32)
                        del draws[:]
                        torch.randint = recording_randint
                        try:
                            bot.observe(pair)
                        finally:
                            torch.randint = original_randint
                        out[p + kind + '_post'] = np.stack([_np(bot.target_node), _np(bot.time_focused), *_np(bot.actions).T]).astype(np.int32)
                        assert len(draws) in (0, B)
                        out[p + kind + '_draws'] = np.asarray(draws, np.int64).reshape(-1, 2)
                    if role == 'defender':  # camp.py reads the mapping under another key than the wrapper's; hand it that key
                        bot = camps[agent]
                        out[p + 'camp_defender_pre'] = _np(bot.actions).astype(np.int32).copy()
                        bot.observe((observations[agent], {'action_task_mappings': mapping}))
                        out[p + 'camp_defender_post'] = _np(bot.actions).astype(np.int32).copy()
                        out[p + 'camp_target'] = np.asarray(int(bot.agent_index % observations[agent]['tasks'].size(1)))
                    which = (a + t // 4) % 2
                    driven[a] = _np(bots[agent][which].actions)
                    cases += 1
                actions = cyber_policy(aec, rng) if driver == 'random' else driven
                observations, *_ = env.step({agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
    out['cases'] = np.asarray(cases)
    path = os.path.join(GOLDEN, 'baselines_cybersecurity.npz')
    np.savez_compressed(path, **out)
    print(f'{path}: {cases} observe() calls x 2 agents')



# ----------------------------------------------------------------------------------------------------------
# CSV logs (logging tap): what utils/logging_handlers.py:CSVLogger writes along the golden trajectories
# ----------------------------------------------------------------------------------------------------------
def record_logs():
    """Re-runs one golden trajectory per domain (same seeds, hence same actions and injected randomness as traj_<domain>_<name>.npz)
    with log_directory set and stores the text of the per-env CSV files the reference wrote — output data of the reference."""
    import shutil
    import tempfile
    from free_range_zoo.envs import wildfire_v0, rideshare_v0, cybersecurity_v0

    def run(module, name, B, max_steps, steps, seed, configuration, flags, policy, inject):
        directory = tempfile.mkdtemp(prefix='frz_reflog_')
        shutil.rmtree(directory)
        env = module.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'),
                                  log_directory=directory, **flags)
        env.reset(seed=torch.arange(B, dtype=torch.int32), options={'log_description': f'golden {name}'})
        if inject:
            env.aec_env.generator.generate = InjectedRandomness(seed)
        rng = np.random.default_rng(seed)
        agents = list(env.aec_env.agents)
        for t in range(steps):
            actions = policy(env.aec_env, rng)
            env.step({agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
        texts = [open(os.path.join(directory, f'{i}.csv')).read() for i in range(B)]
        shutil.rmtree(directory)
        return np.asarray(texts)

    out = {}
    v = {x[0]: x for x in wildfire_variants()}['rich_localized']
    flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
    flags.update(v[2])
    out['wildfire_name'] = np.asarray(v[0])
    out['wildfire'] = run(wildfire_v0, v[0], v[3], v[4], v[5], v[6], v[1], flags, wildfire_policy, True)
    v = {x[0]: x for x in cyber_variants()}['rich']
    flags = dict(observe_other_location=False, observe_other_presence=False, observe_other_power=True, partially_observable=True,
                 show_bad_actions=True)
    flags.update(v[2])
    out['cybersecurity_name'] = np.asarray(v[0])
    out['cybersecurity'] = run(cybersecurity_v0, v[0], v[3], v[4], v[5], v[6], v[1], flags, cyber_policy, True)
    v = {x[0]: x for x in rideshare_variants()}['busy_waiting_costs']
    out['rideshare_name'] = np.asarray(v[0])
    out['rideshare'] = run(rideshare_v0, v[0], v[2], v[3], v[4], v[5], v[1], {}, rideshare_policy, False)
    path = os.path.join(GOLDEN, 'logs_csv.npz')
    np.savez_compressed(path, **out)
    print(path, {k: (len(out[k]), sum(len(t) for t in out[k])) for k in ('wildfire', 'cybersecurity', 'rideshare')})



# ----------------------------------------------------------------------------------------------------------
# SQL logs: what utils/logging_handlers.py:SQLLogger leaves in a sqlite database along the same trajectories
# ----------------------------------------------------------------------------------------------------------
SQL_TABLES = ('simulation', 'environment', 'agent', 'environment_timestep', 'wildfire_environment_log', 'rideshare_environment_log',
              'cybersecurity_environment_log', 'agent_log')


def dump_sqlite(path):
    """Every table of the logging schema as {'table': {'columns': [...], 'rows': [[...], ...]}} (rows in id order; the simulation
    table's date column is left out: it is the day of the run)."""
    import sqlite3
    con = sqlite3.connect(path)
    out = {}
    for table in SQL_TABLES:
        cur = con.execute(f'SELECT * FROM {table} ORDER BY id')
        columns = [c[0] for c in cur.description]
        rows = [list(r) for r in cur.fetchall()]
        if table == 'simulation':
            keep = [i for i, c in enumerate(columns) if c != 'timestamp']
            columns, rows = [columns[i] for i in keep], [[r[i] for i in keep] for r in rows]
        out[table] = {'columns': columns, 'rows': rows}
    con.close()
    return out


def record_sql_logs():
    """The golden trajectories of record_logs() re-run with log_directory='sqlite:///...' (same seeds, hence same actions and injected
    randomness) and the database the reference's SQLLogger wrote dumped table by table — output data of the reference."""
    import json
    import tempfile
    from free_range_zoo.envs import wildfire_v0, rideshare_v0, cybersecurity_v0

    def run(module, name, B, max_steps, steps, seed, configuration, flags, policy, inject):
        handle, path = tempfile.mkstemp(prefix='frz_refsql_', suffix='.db')
        os.close(handle)
        os.remove(path)
        env = module.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'),
                                  log_directory=f'sqlite:///{path}', **flags)
        env.reset(seed=torch.arange(B, dtype=torch.int32), options={'log_description': f'golden {name}', 'log_label': f'run {name}'})
        if inject:
            env.aec_env.generator.generate = InjectedRandomness(seed)
        rng = np.random.default_rng(seed)
        agents = list(env.aec_env.agents)
        for t in range(steps):
            actions = policy(env.aec_env, rng)
            env.step({agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
        env.aec_env.logger.session.close()
        tables = dump_sqlite(path)
        os.remove(path)
        return np.asarray(json.dumps(tables))

    out = {}
    v = {x[0]: x for x in wildfire_variants()}['rich_localized']
    flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
    flags.update(v[2])
    out['wildfire_name'] = np.asarray(v[0])
    out['wildfire'] = run(wildfire_v0, v[0], v[3], v[4], v[5], v[6], v[1], flags, wildfire_policy, True)
    v = {x[0]: x for x in cyber_variants()}['rich']
    flags = dict(observe_other_location=False, observe_other_presence=False, observe_other_power=True, partially_observable=True,
                 show_bad_actions=True)
    flags.update(v[2])
    out['cybersecurity_name'] = np.asarray(v[0])
    out['cybersecurity'] = run(cybersecurity_v0, v[0], v[3], v[4], v[5], v[6], v[1], flags, cyber_policy, True)
    v = {x[0]: x for x in rideshare_variants()}['busy_waiting_costs']
    out['rideshare_name'] = np.asarray(v[0])
    try:
        out['rideshare'] = run(rideshare_v0, v[0], v[2], v[3], v[4], v[5], v[1], {}, rideshare_policy, False)
    except Exception as failure:  # noqa: BLE001 - recorded: what the reference does with this configuration
        out['rideshare'] = np.asarray(json.dumps({'raises': type(failure).__name__, 'message': str(failure)[:200]}))
    path = os.path.join(GOLDEN, 'logs_sql.npz')
    np.savez_compressed(path, **out)
    print(path, {k: len(str(out[k])) for k in ('wildfire', 'cybersecurity', 'rideshare')})


# ----------------------------------------------------------------------------------------------------------
# configurations pickled by the reference (how the reference distributes its competition configurations)
# ----------------------------------------------------------------------------------------------------------
def record_pickles():
    """pickle.dumps of reference Configuration objects (one per domain): the bytes a user's `<configuration>.pkl` holds."""
    import pickle
    out = {
        'wildfire': {x[0]: x for x in wildfire_variants()}['rich_localized'][1],
        'cybersecurity': {x[0]: x for x in cyber_variants()}['rich'][1],
        'rideshare': {x[0]: x for x in rideshare_variants()}['busy_waiting_costs'][1],
    }
    for domain, configuration in out.items():
        path = os.path.join(GOLDEN, f'reference_configuration_{domain}.pkl')
        with open(path, 'wb') as handle:
            pickle.dump(configuration.to(torch.device('cpu')), handle)
        print(path, os.path.getsize(path), 'bytes', type(configuration).__module__)



# ----------------------------------------------------------------------------------------------------------
# partial resets / restarts from a saved state, recorded from the unmodified reference envs
# ----------------------------------------------------------------------------------------------------------
def _record_steps(env, out, prefix, steps, policy, source, rng, random_names, snapshot):
    agents = list(env.aec_env.agents)
    for t in range(steps):
        actions = policy(env.aec_env, rng)
        mark = len(source.log)
        # the env keeps the tensors it is given (and reset_batches later scribbles torch.empty() into them): hand it copies
        _, rewards, terminations, truncations, infos = env.step({agent: torch.from_numpy(actions[a].copy()) for a, agent in enumerate(agents)})
        drawn = source.log[mark:]
        p = f'{prefix}{t}_'
        out[p + 'actions'] = actions.copy()
        out[p + 'stepped'] = np.asarray(len(drawn) == 2)
        if len(drawn) == 2:
            out[p + random_names[0]], out[p + random_names[1]] = _np(drawn[0]), _np(drawn[1])
        out[p + 'rewards'] = np.stack([_np(rewards[agent]) for agent in agents])
        out[p + 'terminations'] = np.stack([_np(terminations[agent]) for agent in agents])
        out[p + 'truncations'] = np.stack([_np(truncations[agent]) for agent in agents])
        out[p + 'finished'] = _np(env.finished)
        snapshot(env, p, out)


def record_partial_resets():
    """utils/env.py:162-189 + wildfire.py:376-397 / cybersecurity.py:274-293 (reset_batches) and the `initial_state` reset option
    (wildfire.py:341-345, cybersecurity.py:241-246): the reference is stepped, partially reset, stepped on; then a second env is
    started from the state the first one was left in.  Randomness is injected (recorded), so seeding plays no part in the values."""
    from free_range_zoo.envs import cybersecurity_v0, wildfire_v0
    from free_range_zoo_amd._capi import struct_to_dict
    from free_range_zoo_amd.envs.wildfire.env.structures.configuration import to_cstruct as wf_cstruct
    from free_range_zoo_amd.envs.cybersecurity.env.structures.configuration import to_cstruct as cy_cstruct

    wf = {v[0]: v for v in wildfire_variants()}
    cy = {v[0]: v for v in cyber_variants()}
    jobs = [('wildfire', wildfire_v0, wf['cfg2_openness'], dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False),
             wf_cstruct, wildfire_policy, wildfire_snapshot, ('field_randomness', 'agent_randomness')),
            ('wildfire_bad_actions', wildfire_v0, wf['openness_bad_actions'],
             dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False), wf_cstruct, wildfire_policy, wildfire_snapshot,
             ('field_randomness', 'agent_randomness')),
            ('wildfire_grid8x8', wildfire_v0, wf['grid8x8_12agents'],
             dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False), wf_cstruct, wildfire_policy, wildfire_snapshot,
             ('field_randomness', 'agent_randomness')),
            ('cybersecurity', cybersecurity_v0, cy[sorted(cy)[0]],
             dict(observe_other_location=False, observe_other_presence=False, observe_other_power=True, partially_observable=True, show_bad_actions=True),
             cy_cstruct, cyber_policy, cyber_snapshot, ('network_randomness', 'agent_randomness'))]
    for label, module, variant, defaults, cstruct_of, policy, snapshot, random_names in jobs:
        name, configuration, kwargs, B, max_steps, steps, seed = variant
        flags = dict(defaults)
        flags.update(kwargs)
        B = 8
        env = module.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'), **flags)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        source = InjectedRandomness(seed + 100)
        env.aec_env.generator.generate = source
        rng = np.random.default_rng(seed + 100)
        out = {'cfg': np.asarray(json.dumps(struct_to_dict(cstruct_of(configuration, B, max_steps, **flags)))), 'variant': np.asarray(name)}
        snapshot(env, 'r_', out)
        _record_steps(env, out, 'a', 6, policy, source, rng, random_names, snapshot)
        # the reference's indexed restore indexes EVERY tensor of the state with the env indices, the [agents, 2] position table of the
        # wildfire state included (utils/state.py:81): env indices below the agent count are the ones it can run on
        batch_indices = torch.tensor([1, 2, 0]) if label.startswith('wildfire') else torch.tensor([1, 4, 6])
        out['batch_indices'] = _np(batch_indices)
        out['batch_seeds'] = np.asarray([111, 222, 333], np.int32)
        aec = env.aec_env
        try:
            aec.reset_batches(batch_indices=batch_indices, seed=torch.tensor([111, 222, 333], dtype=torch.int32))
            out['reset_batches_reference_error'] = np.asarray('')
        except AttributeError as error:
            # The unmodified reference cannot finish this call: State.restore_initial (utils/state.py:47) tests `self.initial`, an attribute
            # that does not exist (save_initial stores `initial_state`).  By then the base class has zeroed the bookkeeping of the chosen envs
            # (utils/env.py:177-189 ran first).  The rest of the method (wildfire.py:390-397 / cybersecurity.py:288-293) is carried out here
            # with the reference's own working pieces: the indexed restore is State.restore_from_checkpoint (the same loop as the one
            # restore_initial would run, utils/state.py:66-83) on the saved initial state.
            out['reset_batches_reference_error'] = np.asarray(f'{type(error).__name__}: {error}')
            state = aec._state
            held = state.checkpoint
            state.checkpoint = state.initial_state
            state.restore_from_checkpoint(batch_indices)
            state.checkpoint = held
            if hasattr(aec, 'num_burnouts'):
                aec.num_burnouts[batch_indices] = 0
            aec.update_observations()
            aec.update_actions()
        snapshot(env, 'p_', out)
        agents = list(env.aec_env.agents)
        out['p_rewards'] = np.stack([_np(env.aec_env.rewards[agent]) for agent in agents])
        out['p_terminations'] = np.stack([_np(env.aec_env.terminations[agent]) for agent in agents])
        out['p_truncations'] = np.stack([_np(env.aec_env.truncations[agent]) for agent in agents])
        _record_steps(env, out, 'b', 5, policy, source, rng, random_names, snapshot)
        # a second env restarted from the state the first one is in now
        saved = env.aec_env.state().clone()
        env2 = module.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'), **flags)
        env2.reset(seed=torch.arange(B, dtype=torch.int32), options={'initial_state': saved})
        source2 = InjectedRandomness(seed + 200)
        env2.aec_env.generator.generate = source2
        snapshot(env2, 'i_', out)
        _record_steps(env2, out, 'c', 4, policy, source2, rng, random_names, snapshot)
        # ... and reset again without reseeding (utils/env.py:122-131: the generator keeps its streams)
        env2.reset(options={'skip_seeding': True})
        snapshot(env2, 'k_', out)
        path = os.path.join(GOLDEN, f'partial_{label}.npz')
        np.savez_compressed(path, **out)
        print(f'{path}: variant {name}, B={B}')


def record_rng():
    """utils/random_generator.py:49-146 on the torch CPU generator: buffered draws (one buffer per key and shape, refilled every buffer_size
    calls), single_seeding (ONE default-seeded stream for every env: the seed is ignored, :62-68) with and without a buffer.  The per-env
    cases use one env: the reference hands `Generator.set_state` a row VIEW of its state table, which this torch only accepts for row 0."""
    from free_range_zoo.utils.random_generator import RandomGenerator

    out = {}
    calls = [('field', 3, (2, 3)), ('agent', 5, (3, )), ('field', 3, (2, 3)), ('agent', 5, (3, )), ('field', 3, (2, 3)), ('other', 1, (4, )),
             ('field', 3, (2, 3)), ('agent', 5, (3, )), (None, 2, (2, )), ('field', 3, (2, 3)), ('agent', 5, (3, ))]
    out['calls'] = np.asarray(json.dumps([[k, e, list(sh)] for k, e, sh in calls]))
    cases = {'buffered_b1': dict(parallel_envs=1, buffer_size=3, single_seeding=False, seed=[1234]),
             'unbuffered_b1': dict(parallel_envs=1, buffer_size=0, single_seeding=False, seed=[77]),
             'single_b4': dict(parallel_envs=4, buffer_size=0, single_seeding=True, seed=[5]),
             'single_buffered_b4': dict(parallel_envs=4, buffer_size=2, single_seeding=True, seed=[5])}
    out['cases'] = np.asarray(json.dumps(cases))
    for name, case in cases.items():
        generator = RandomGenerator(parallel_envs=case['parallel_envs'], buffer_size=case['buffer_size'], single_seeding=case['single_seeding'],
                                    device=torch.device('cpu'))
        generator.seed(torch.tensor(case['seed'], dtype=torch.int32))
        for i, (key, events, shape) in enumerate(calls):
            out[f'{name}_{i}'] = _np(generator.generate(case['parallel_envs'], events, shape, key=key))
    np.savez_compressed(os.path.join(GOLDEN, 'rng_modes.npz'), **out)
    print('rng_modes.npz written')

# ----------------------------------------------------------------------------------------------------------
# action / observation spaces along the recorded trajectories
# ----------------------------------------------------------------------------------------------------------
def _canon_space(space):
    """Structure of a space as the reference's builders construct it (the stand-in for free_range_rust.Space in shim.py keeps the
    constructor name and arguments): plain JSON."""
    def number(v):
        v = v.item() if hasattr(v, 'item') else v
        if v is None:  # (a bound the configuration leaves open, e.g. AgentConfiguration.suppressant_states)
            return None
        return int(v) if float(v) == int(v) else float(v)

    if isinstance(space, (list, tuple)):
        return {'kind': 'list', 'spaces': [_canon_space(s) for s in space]}
    kind, args, kwargs = space.kind, space.args, space.kwargs
    if kind == 'Discrete':
        return {'kind': 'Discrete', 'n': number(args[0]), 'start': number(kwargs.get('start', args[1] if len(args) > 1 else 0))}
    if kind == 'Box':
        low = kwargs['low'] if 'low' in kwargs else args[0]
        high = kwargs['high'] if 'high' in kwargs else args[1]
        return {'kind': 'Box', 'low': [number(v) for v in low], 'high': [number(v) for v in high]}
    if kind in ('OneOf', 'Tuple', 'Vector'):
        return {'kind': kind, 'spaces': [_canon_space(s) for s in args[0]]}
    if kind == 'Dict':
        return {'kind': 'Dict', 'spaces': {key: _canon_space(value) for key, value in args[0].items()}}
    raise ValueError(f'unknown space constructor {kind}')


class _SpaceTable:
    """Per-env spaces deduplicated: `table` holds each distinct structure once (JSON text), the per-step arrays index into it."""

    def __init__(self):
        self.table, self.index = [], {}

    def ids(self, batch):
        """batch: a Vector / list of per-env spaces -> (container kind, int32 [B] table indices)."""
        canon = _canon_space(batch)
        out = []
        for entry in canon['spaces']:
            text = json.dumps(entry, sort_keys=True)
            if text not in self.index:
                self.index[text] = len(self.table)
                self.table.append(text)
            out.append(self.index[text])
        return canon['kind'], np.asarray(out, np.int32)


def _record_spaces_of(env, table, out, prefix):
    aec = env.aec_env
    for a, agent in enumerate(aec.agents):
        kind, ids = table.ids(aec.action_space(agent))
        out[f'{prefix}action_{a}'] = ids
        out.setdefault('action_container', np.asarray(kind))
        assert str(out['action_container']) == kind
        kind, ids = table.ids(aec.observation_space(agent))
        out[f'{prefix}observation_{a}'] = ids
        out.setdefault('observation_container', np.asarray(kind))
        assert str(out['observation_container']) == kind


def record_spaces():
    """Replays the trajectory runs (same variants, seeds, policies and injected randomness as the traj_* parts: the recorded task counts
    are asserted to agree step by step) and records every agent's action / observation space at the reset and after each step."""
    from free_range_zoo.envs import cybersecurity_v0, rideshare_v0, wildfire_v0

    def run(domain, name, env, steps, policy, rng, count_of):
        reference = np.load(os.path.join(GOLDEN, f'traj_{domain}_{name}.npz'))
        table, out = _SpaceTable(), {'steps': np.asarray(steps)}
        _record_spaces_of(env, table, out, 'r_')
        agents = list(env.aec_env.agents)
        for t in range(steps):
            actions = policy(env.aec_env, rng)
            assert np.array_equal(actions, reference[f's{t}_actions']), f'{domain} {name}: the replay left the recorded trajectory at step {t}'
            env.step({agent: torch.from_numpy(actions[a]) for a, agent in enumerate(agents)})
            assert np.array_equal(_np(count_of(env.aec_env)), reference[f's{t}_env_task_count']), (domain, name, t)
            _record_spaces_of(env, table, out, f's{t}_')
        out['table'] = np.asarray(json.dumps(table.table))
        path = os.path.join(GOLDEN, f'spaces_{domain}_{name}.npz')
        np.savez_compressed(path, **out)
        print(f'{path}: {len(table.table)} distinct per-env spaces over {steps + 1} snapshots')

    for name, configuration, kwargs, B, max_steps, steps, seed in wildfire_variants():
        flags = dict(show_bad_actions=False, observe_other_power=False, observe_other_suppressant=False)
        flags.update(kwargs)
        env = wildfire_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'), **flags)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        env.aec_env.generator.generate = InjectedRandomness(seed)
        run('wildfire', name, env, steps, wildfire_policy, np.random.default_rng(seed), lambda aec: aec.environment_task_count)
    for name, configuration, kwargs, B, max_steps, steps, seed in cyber_variants():
        flags = dict(observe_other_location=False, observe_other_presence=False, observe_other_power=True, partially_observable=True,
                     show_bad_actions=True)
        flags.update(kwargs)
        env = cybersecurity_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'), **flags)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        env.aec_env.generator.generate = InjectedRandomness(seed)
        run('cybersecurity', name, env, steps, cyber_policy, np.random.default_rng(seed), lambda aec: aec.environment_task_count)
    for name, configuration, B, max_steps, steps, seed in rideshare_variants():
        env = rideshare_v0.parallel_env(parallel_envs=B, max_steps=max_steps, configuration=configuration, device=torch.device('cpu'))
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        run('rideshare', name, env, steps, rideshare_policy, np.random.default_rng(seed), lambda aec: aec.environment_task_count)


PARTS = {'spaces': record_spaces, 'partial': record_partial_resets, 'rng': record_rng, 'ka': record_known_answers, 'baselines_wildfire': record_wildfire_baselines, 'baselines_rideshare': record_rideshare_baselines, 'baselines_cybersecurity': record_cyber_baselines, 'logs': record_logs, 'logs_sql': record_sql_logs, 'pickles': record_pickles, 'traj_wildfire': record_wildfire_trajectories, 'traj_cybersecurity': record_cyber_trajectories,
         'traj_rideshare': record_rideshare_trajectories,
         'misc': record_misc}

if __name__ == '__main__':
    torch.set_num_threads(4)
    for part in (sys.argv[1:] or list(PARTS)):
        PARTS[part]()
