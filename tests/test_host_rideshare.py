"""CPU: host-side logic of the rideshare boundary (configuration lowering and validation)."""
import json

import numpy as np
import pytest
import torch

import configs
import golden_util as G
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs.rideshare.env.structures import configuration as R


@pytest.mark.parametrize('name', sorted(configs.RIDESHARE_GOLDEN))
def test_configuration_lowers_to_the_reference_struct(name):
    data = np.load(G.golden_path(f'traj_rideshare_{name}.npz'))
    want = json.loads(str(data['cfg']))
    max_steps = None if want['max_steps'] < 0 else want['max_steps']
    cfg, schedule = R.to_cstruct(configs.RIDESHARE_GOLDEN[name](), want['parallel_envs'], max_steps)
    assert _capi.struct_to_dict(cfg) == want
    assert np.array_equal(schedule, data['schedule'])


def test_validation_and_slot_bound():
    cfg = configs.rideshare_non_stochastic()
    assert cfg.agent_config.num_agents == 4 and cfg.max_fare == 3
    with pytest.raises(ValueError):
        R.AgentConfiguration(start_positions=torch.zeros((2, 2)), pool_limit=0, use_diagonal_travel=False, use_fast_travel=False)
    with pytest.raises(ValueError):
        R.PassengerConfiguration(schedule=torch.zeros((3, 6), dtype=torch.int))
    with pytest.raises(ValueError):
        configs._rideshare_base_rewards(wait_limit=torch.tensor([1, 0, 3]))
    with pytest.raises(ValueError):
        R.RideshareConfiguration(grid_height=0, grid_width=5, agent_config=cfg.agent_config, passenger_config=cfg.passenger_config,
                                 reward_config=cfg.reward_config)
    schedule = np.array([[0, -1, 0, 0, 0, 0, 1], [3, 2, 0, 0, 0, 0, 1], [4, 2, 0, 0, 0, 0, 1], [4, 0, 0, 0, 0, 0, 1]], np.int32)
    assert R.default_max_passengers(schedule, 4) == 3  # one wildcard + at most two rows of env 2
    with pytest.raises(ValueError):
        R.to_cstruct(configs.rideshare_busy(steps=100, per_step=2), 4, 10)  # 200 slots > FRZ_MAX_PASSENGERS
