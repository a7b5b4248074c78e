"""CPU: the observation / action space builders of the three domains (envs/*/env/spaces) — structure, the reference's flag handling and
its builder caches (cases modelled on the reference's tests/free_range_zoo/envs/*/env/spaces/test_{observation,action}_space.py), and the
count-based batched spaces the envs hand out."""
import pytest
import torch

from free_range_zoo_amd.utils.spaces import BatchedOneOfSpace, BatchedSpace, Space
from free_range_zoo_amd.envs.wildfire.env.spaces import actions as wf_actions, observations as wf_obs
from free_range_zoo_amd.envs.rideshare.env.spaces import actions as rs_actions, observations as rs_obs
from free_range_zoo_amd.envs.cybersecurity.env.spaces import actions as cy_actions, observations as cy_obs


def box(high):
    return Space.Box(low=[0] * len(high), high=high)


# ------------------------------------------------------------------------------------------------------------------ caches
CACHED = [
    (wf_obs.build_single_observation_space, ((10, 10, 5, 8), (10, 10, 3, 4), 4, 3, True, True), ((10, 10, 5, 4), (10, 10, 3, 4), 3, 3, True, True)),
    (wf_obs.build_single_agent_observation_space, ((10, 10, 5, 5), ), ((10, 10, 5, 4), )),
    (wf_obs.build_single_fire_observation_space, ((10, 10, 5, 5), 3), ((10, 10, 5, 5), 4)),
    (wf_actions.build_single_action_space, (3, ), (4, )),
    (rs_obs.build_single_observation_space, ((10, 10, 4, 4), (10, 10, 10, 10, 8, 8, 10, 50), 4, 3), ((10, 10, 4, 4), (10, 10, 10, 10, 8, 8, 10, 50), 2, 3)),
    (rs_actions.build_single_action_space, ((0, 1, 2), ), ((0, 0), )),
    (cy_obs.build_observation_space, ('attacker', 3, 10, 5, 5, (10, 1), (3, 2, 1), (5, ), True, True, True),
     ('defender', 3, 10, 5, 5, (10, 1), (3, 2, 1), (5, ), True, True, True)),
    (cy_obs.build_single_subnetwork_observation_space, ((5, ), 3), ((5, ), 4)),
    (cy_actions.build_single_defender_action_space, (3, 0, False), (3, -1, False)),
    (cy_actions.build_single_attacker_action_space, (3, ), (2, )),
]


@pytest.mark.parametrize('func,first,second', CACHED, ids=lambda v: getattr(v, '__name__', None))
def test_builders_are_cached_like_the_reference(func, first, second):
    func.cache_clear()
    func(*first)
    assert (func.cache_info().hits, func.cache_info().misses) == (0, 1)
    func(*first)
    assert (func.cache_info().hits, func.cache_info().misses) == (1, 1)
    func(*second)
    assert (func.cache_info().hits, func.cache_info().misses) == (1, 2)
    assert func(*first) is func(*first)


# ------------------------------------------------------------------------------------------------------------------ wildfire
def test_wildfire_observation_space_structure_and_flags():
    agent_high, fire_high = (10, 10, 5, 8), (10, 10, 3, 4)
    full = wf_obs.build_single_observation_space(agent_high, fire_high, 4, 3, True, True)
    assert full == Space.Dict({'self': box(agent_high), 'others': Space.Tuple([box(agent_high)] * 2),
                               'tasks': Space.Tuple([box(fire_high)] * 4)})
    assert wf_obs.build_single_observation_space(agent_high, fire_high, 4, 3, True, False)['others'] == Space.Tuple([box(agent_high[:3])] * 2)
    assert wf_obs.build_single_observation_space(agent_high, fire_high, 4, 3, False, True)['others'] == Space.Tuple([box((10, 10, 8))] * 2)
    assert wf_obs.build_single_observation_space(agent_high, fire_high, 4, 3, False, False)['others'] == Space.Tuple([box(agent_high[:2])] * 2)
    for agents in range(1, 11):
        assert len(wf_obs.build_single_observation_space(agent_high, fire_high, 4, agents, True, True)['others']) == agents - 1
    for tasks in range(1, 11):
        assert wf_obs.build_single_fire_observation_space(fire_high, tasks) == Space.Tuple([box(fire_high) for _ in range(tasks)])


def test_wildfire_batched_observation_space_is_count_based():
    counts = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
    space = wf_obs.build_observation_space(counts, 3, (10, 10, 5, 5), (10, 10, 3, 4), True, True)
    assert isinstance(space, BatchedSpace) and len(space) == 11
    assert space == [wf_obs.build_single_observation_space((10, 10, 5, 5), (10, 10, 3, 4), i, 3, True, True) for i in range(11)]
    assert len(space[7]['tasks']) == 7 and space[0]['tasks'] == Space.Tuple([])
    # the reference hands its two flags on positionally, suppressant first, to (include_power, include_suppressant): kept as written
    only_suppressant = wf_obs.build_observation_space(counts, 3, (10, 10, 5, 8), (10, 10, 3, 4), True, False)
    assert only_suppressant[2]['others'] == Space.Tuple([box((10, 10, 5))] * 2)


def test_wildfire_action_space():
    assert wf_actions.build_single_action_space(0) == Space.OneOf([Space.Discrete(1, start=-1)])
    assert wf_actions.build_single_action_space(3) == Space.OneOf([Space.Discrete(1, start=0)] * 3 + [Space.Discrete(1, start=-1)])
    batched = wf_actions.build_action_space(torch.tensor([0, 2, 5]))
    assert isinstance(batched, BatchedOneOfSpace)
    assert batched.spaces == [wf_actions.build_single_action_space(n) for n in (0, 2, 5)]
    assert Space.Vector([wf_actions.build_single_action_space(n) for n in (0, 2, 5)]) == batched


# ------------------------------------------------------------------------------------------------------------------ rideshare
def test_rideshare_spaces():
    agent_high, passenger_high = (10, 10, 4, 4), (10, 10, 10, 10, 8, 8, 10, 50)
    single = rs_obs.build_single_observation_space(agent_high, passenger_high, 4, 3)
    assert single == Space.Dict({'self': box(agent_high), 'others': Space.Tuple([box(agent_high)] * 2), 'tasks': Space.Tuple([box(passenger_high)] * 4)})
    space = rs_obs.build_observation_space(torch.tensor([3, 0, 1]), 3, agent_high, passenger_high)
    assert [len(s['tasks']) for s in space] == [3, 0, 1]
    assert rs_actions.build_single_action_space(()) == Space.OneOf([Space.Discrete(1, start=-1)])
    assert rs_actions.build_single_action_space((0, 2, 1)) == Space.OneOf([Space.Discrete(1, start=s) for s in (0, 2, 1, -1)])
    starts = torch.tensor([[0, 2, 1], [1, 0, 0], [0, 0, 0]])
    batched = rs_actions.build_action_space(starts, torch.tensor([3, 1, 0]))
    assert batched.spaces == [rs_actions.build_single_action_space(t) for t in ((0, 2, 1), (1, ), ())]


# ------------------------------------------------------------------------------------------------------------------ cybersecurity
def test_cybersecurity_observation_spaces():
    attacker_high, defender_high, network_high = (10, 1), (3, 2, 1), (5, )
    spaces = cy_obs.build_observation_space('attacker', 3, 10, 5, 4, attacker_high, defender_high, network_high, True, True, True)
    assert len(spaces) == 10 and all(s is spaces[0] for s in spaces)
    assert spaces[0] == Space.Dict({'self': box(attacker_high), 'others': Space.Tuple([box(attacker_high)] * 4), 'tasks': Space.Tuple([box(network_high)] * 3)})
    defender = cy_obs.build_observation_space('defender', 3, 10, 5, 4, attacker_high, defender_high, network_high, True, False, True)[0]
    assert defender['others'] == Space.Tuple([box((3, 1))] * 3) and defender['self'] == box(defender_high)
    nothing = cy_obs.build_single_attacker_observation_space(attacker_high, network_high, 3, 3, False, False)
    assert nothing['others'] == Space.Tuple([Space.Discrete(0, start=0)] * 2)
    with pytest.raises(ValueError):
        cy_obs.build_observation_space('observer', 3, 10, 5, 4, attacker_high, defender_high, network_high, True, True, True)


def test_cybersecurity_action_spaces():
    noop = Space.OneOf([Space.Discrete(1, start=-1)])
    assert cy_actions.build_single_attacker_action_space(0) == noop and cy_actions.build_single_defender_action_space(0, 2, True) == noop
    moves = [Space.Discrete(1, start=0)] * 3
    assert cy_actions.build_single_attacker_action_space(3) == Space.OneOf(moves + [Space.Discrete(1, start=-1)])
    assert cy_actions.build_single_defender_action_space(3, 1, False) == Space.OneOf(moves + [Space.Discrete(1, start=v) for v in (-1, -2, -3)])
    assert cy_actions.build_single_defender_action_space(3, -1, False) == Space.OneOf(moves + [Space.Discrete(1, start=v) for v in (-1, -3)])
    assert cy_actions.build_single_defender_action_space(3, -1, True) == Space.OneOf(moves + [Space.Discrete(1, start=v) for v in (-1, -2, -3)])
    counts, location = torch.tensor([3, 0, 3, 3]), torch.tensor([1, 0, -1, -1])
    batched = cy_actions.build_action_space('defender', False, counts, location)
    assert batched.spaces == [cy_actions.build_single_defender_action_space(int(n), int(l), False) for n, l in zip(counts, location)]
    assert cy_actions.build_action_space('attacker', False, counts).spaces == [cy_actions.build_single_attacker_action_space(int(n)) for n in counts]
    with pytest.raises(ValueError):
        cy_actions.build_action_space('observer', False, counts, location)


# ------------------------------------------------------------------------------------------------------------
# the builders against the spaces the reference handed out along its recorded trajectories (tests/golden/spaces_wildfire_*.npz), from the
# recorded task counts: no GPU (the HIP envs' action_space / observation_space are compared in tests/test_hip_spaces_golden.py)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['cfg2_openness', 'openness_bad_actions', 'openness_observe_all', 'rich_localized'])
def test_wildfire_builders_reproduce_the_recorded_reference_spaces(name):
    import numpy as np
    import torch
    import configs
    import golden_util as G
    from free_range_zoo_amd.envs.wildfire.env.spaces import actions, observations
    from free_range_zoo_amd.utils.spaces import bounds
    build, kwargs = configs.WILDFIRE_GOLDEN[name]
    cfg = build()
    traj = np.load(G.golden_path(f'traj_wildfire_{name}.npz'))
    data, table = G.load_spaces('wildfire', name)
    A = cfg.agent_config.num_agents
    agent_high = bounds((cfg.grid_height, cfg.grid_width, cfg.agent_config.max_fire_reduction_power, cfg.agent_config.suppressant_states))
    fire_high = bounds((cfg.grid_height, cfg.grid_width, cfg.fire_config.max_fire_type, cfg.fire_config.num_fire_states))
    for prefix in ['r_'] + [f's{t}_' for t in range(int(traj['steps']))]:
        env_counts = torch.from_numpy(traj[prefix + 'env_task_count'])
        for a in range(A):
            counts = env_counts if kwargs.get('show_bad_actions') else torch.from_numpy(traj[prefix + 'agent_task_count'][a])
            got = G.canon_space(actions.build_action_space(counts))
            assert got['kind'] == str(data['action_container'])
            assert got['spaces'] == [table[i] for i in data[f'{prefix}action_{a}']], f'{name} {prefix} action space of agent {a}'
            got = G.canon_space(observations.build_observation_space(env_counts, A, agent_high, fire_high,
                                                                     kwargs.get('observe_other_suppressant', False), kwargs.get('observe_other_power', False)))
            assert got['kind'] == str(data['observation_container'])
            assert got['spaces'] == [table[i] for i in data[f'{prefix}observation_{a}']], f'{name} {prefix} observation space of agent {a}'


def test_open_bounds_given_as_infinity_stay_floats():
    """`int(inf)` raised OverflowError inside Box."""
    from free_range_zoo_amd.utils import spaces
    box = spaces.Box([0, float('-inf')], [float('inf'), 2.5])
    assert box.low == [0, float('-inf')] and box.high == [float('inf'), 2.5]
    assert spaces.bounds((3.0, float('inf'), None)) == (3, float('inf'), None)
