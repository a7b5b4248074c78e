"""GPU: the behaviours the reference pins on its base classes with mocks (tests/free_range_zoo/utils/test_env.py:133-279, 414-459 and
test_conversions.py:42-99), replayed on the real envs: partial resets of the bookkeeping, what a step does to it, the calculated
finished / terminated / truncated properties, the adapter's return values."""
import pytest
import torch

import configs

pytestmark = pytest.mark.gpu


def make(domain, B, max_steps=10, **kwargs):
    from free_range_zoo_amd.envs import cybersecurity_v0, wildfire_v0
    module, build = {'wildfire': (wildfire_v0, configs.wildfire_openness), 'cybersecurity': (cybersecurity_v0, configs.cyber_openness)}[domain]
    return module.parallel_env(configuration=build(), parallel_envs=B, max_steps=max_steps, device=torch.device('cuda'), **kwargs)


@pytest.mark.parametrize('domain', ['wildfire', 'cybersecurity'])
def test_reset_batches_resets_the_bookkeeping_of_the_selected_envs_only(domain):
    """test_env.py:133-185 (TestResetBatches): rewards, cumulative rewards, terminations, truncations and num_moves are zeroed at
    batch_indices and left alone elsewhere."""
    env = make(domain, 4, track_cumulative_rewards=True) if domain == 'wildfire' else make(domain, 4)
    env.reset(seed=torch.arange(4, dtype=torch.int32))
    env._rewards.fill_(1), env._cumulative.fill_(1), env._terminations.fill_(True), env._truncations.fill_(True), env.num_moves.fill_(1)
    env.reset_batches(batch_indices=torch.tensor([1, 3], dtype=torch.int32), seed=torch.tensor([12345, 67890], dtype=torch.int32))
    ones = torch.tensor([1, 0, 1, 0], device='cuda')
    for agent in env.possible_agents:
        assert torch.equal(env.rewards[agent], ones.float()), agent
        assert torch.equal(env._cumulative_rewards[agent], ones.float()), agent
        assert torch.equal(env.terminations[agent], ones.bool()) and torch.equal(env.truncations[agent], ones.bool()), agent
    assert torch.equal(env.num_moves, ones.int())


@pytest.mark.parametrize('domain', ['wildfire', 'cybersecurity'])
def test_finished_terminated_truncated_are_conjunctions_over_the_agents(domain):
    """test_env.py:414-459 (TestCalculatedProperties): an env is terminated / truncated when every agent is; finished = either."""
    env = make(domain, 3)
    env.reset(seed=torch.arange(3, dtype=torch.int32))
    first = torch.tensor([False, True, True], device='cuda')
    rest = torch.tensor([False, True, False], device='cuda')
    env._terminations[0].copy_(first), env._truncations[0].copy_(first)
    env._terminations[1:].copy_(rest.expand_as(env._terminations[1:])), env._truncations[1:].copy_(rest.expand_as(env._truncations[1:]))
    for name in ('finished', 'terminated', 'truncated'):
        assert torch.equal(getattr(env, name), rest), name
    env._truncations.fill_(False)
    assert torch.equal(env.finished, rest) and not bool(env.truncated.any())


@pytest.mark.parametrize('domain', ['wildfire', 'cybersecurity'])
def test_step_bookkeeping_and_adapter_return_values(domain):
    """test_env.py:187-279 (TestStep) and test_conversions.py:42-99: a step of the parallel adapter counts one move, adds the step's
    rewards to the cumulative rewards without clearing them, returns five dicts keyed by the agents whose tensors are the env's own,
    truncates at max_steps, and leaves the agent list as it was."""
    B = 5
    env = make(domain, B, max_steps=3, track_cumulative_rewards=True) if domain == 'wildfire' else make(domain, B, max_steps=3)
    observations, infos = env.reset(seed=torch.arange(B, dtype=torch.int32))
    assert tuple(observations) == tuple(env.agents) == tuple(env.possible_agents) and set(env.agents) <= set(infos)
    running = {agent: torch.zeros(B, device='cuda') for agent in env.agents}
    for t in range(3):
        actions = {agent: env.action_space(agent).sample_nested() for agent in env.agents}
        observations, rewards, terminations, truncations, infos = env.step(actions)
        assert [tuple(d) for d in (observations, rewards, terminations, truncations)] == [tuple(env.agents)] * 4
        assert torch.equal(env.num_moves, torch.full((B, ), t + 1, dtype=torch.int32, device='cuda'))
        for agent in env.agents:
            running[agent] += rewards[agent]
            assert rewards[agent].dtype == torch.float32 and rewards[agent].shape == (B, )
            assert torch.equal(env.rewards[agent], rewards[agent])  # not cleared after the update (test_env.py:268-271)
            assert torch.allclose(env._cumulative_rewards[agent], running[agent], rtol=1e-5, atol=1e-5), agent
            assert terminations[agent].dtype == torch.bool and truncations[agent].dtype == torch.bool
            assert bool(truncations[agent].all()) == (t == 2)  # utils/env.py:228-233
        for agent in env.agents:
            assert set(observations[agent].keys()) == {'self', 'others', 'tasks'}
    assert bool(env.finished.all())
    env.check()
