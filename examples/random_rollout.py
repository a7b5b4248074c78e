"""Random-policy rollouts of the three domains on one MI355X: the reference's rollout loop, then the same episode as ONE HIP-graph
replay.  usage: python examples/random_rollout.py [parallel_envs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import configs  # noqa: E402  (the configurations of the reference's own tests, rebuilt with this package's classes)
from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
STEPS = 50
device = torch.device('cuda')

for name, module, configuration in (('wildfire', wildfire_v0, configs.wildfire_openness()), ('cybersecurity', cybersecurity_v0, configs.cyber_openness()),
                                    ('rideshare', rideshare_v0, configs.rideshare_busy())):
    env = module.parallel_env(configuration=configuration, parallel_envs=B, max_steps=STEPS, device=device, rng='philox', exact_shapes=False)

    # 1. the loop a user of the reference already has: sample every agent's action space, step, repeat (third episode timed: the first ones warm the allocator up)
    for episode in range(3):
        observations, infos = env.reset(seed=torch.arange(B, dtype=torch.int32))
        torch.cuda.synchronize()
        start = time.perf_counter()
        while not torch.all(env.finished):
            actions = {agent: env.action_space(agent).sample_nested() for agent in env.agents}
            observations, rewards, terminations, truncations, infos = env.step(actions)
        torch.cuda.synchronize()
        loop = time.perf_counter() - start

    # 2. the same kind of episode with the policy sampled on the device, captured once and replayed as one graph
    graph = env.capture_random_rollout(STEPS, policy_seed=1, include_reset=True)
    graph.replay()
    torch.cuda.synchronize()
    start = time.perf_counter()
    graph.replay()
    torch.cuda.synchronize()
    replay = time.perf_counter() - start
    env.check()
    total = sum(float(env._cumulative_rewards[agent].sum()) for agent in env.agents)
    print(f'{name:14s} B={B}: python loop {B * STEPS / loop / 1e6:9.1f} M env-steps/s   graph replay {B * STEPS / replay / 1e6:9.1f} M env-steps/s   '
          f'(sum of episode rewards {total:.1f})')
