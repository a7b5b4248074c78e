"""Kernel micro-benchmark: time the fused step launch alone (HIP events on the launch stream), for tuning sweeps.

usage: python tools/kbench.py [domain] [B] [steps] [rng]
       domain in {wildfire, cybersecurity, rideshare}; env: FRZ_WF_KERNEL=lane|roles"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import configs
from free_range_zoo_amd.envs import wildfire_v0, cybersecurity_v0, rideshare_v0

domain = sys.argv[1] if len(sys.argv) > 1 else 'wildfire'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rng = sys.argv[4] if len(sys.argv) > 4 else 'philox'
make = {'wildfire': (wildfire_v0, configs.wildfire_openness), 'cybersecurity': (cybersecurity_v0, configs.cyber_openness),
        'rideshare': (rideshare_v0, configs.rideshare_busy)}[domain]
env = make[0].parallel_env(configuration=make[1](), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng=rng, exact_shapes=False)
env.reset(seed=torch.arange(B, dtype=torch.int32))
times, policy_times = [], []
for rep in range(3):
    env.reset(seed=torch.arange(B, dtype=torch.int32) + rep)
    pairs = []
    for t in range(steps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        acts = env.random_policy_actions(policy_seed=1, policy_step=t)
        e[1].record()
        env.step(acts)
        e[2].record()
        pairs.append(e)
    torch.cuda.synchronize()
    times = [a[1].elapsed_time(a[2]) * 1e3 for a in pairs]
    policy_times = [a[0].elapsed_time(a[1]) * 1e3 for a in pairs]
env.check()
extra = ''
if domain == 'rideshare':
    extra = f' passengers/env(last)={float(env._passenger_count.float().mean()):.1f}'
print(f'{domain} B={B} rng={rng} step kernel us: median={np.median(times):.1f} min={np.min(times):.1f} max={np.max(times):.1f} '
      f'last={times[-1]:.1f} policy median={np.median(policy_times):.1f}{extra}')
