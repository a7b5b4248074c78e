"""Cybersecurity step time (cfg4 shape, B = 65536): event-bracketed run of graph-replayed random-policy episodes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd.envs import cybersecurity_v0
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = sys.argv[2] if len(sys.argv) > 2 else 'philox'
env = cybersecurity_v0.parallel_env(configuration=configs.cyber_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), exact_shapes=False, rng=rng)
env.reset(seed=torch.arange(B, dtype=torch.int32))
g = env.capture_random_rollout(50, policy_seed=1, include_reset=False)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    torch.cuda.synchronize()
    s.record()
    if g is not None:
        g.replay()
    else:
        for t in range(50):
            env.step_random_policy(1, t)
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 50 * 1e3)
print(os.environ.get('FRZ_CY_KERNEL', 'roles'), rng, 'B', B, f'{best:.2f} us per step', 'graph' if g is not None else 'eager')
