"""Kernel micro-benchmark: time the fused wildfire step launch alone (HIP events), for tuning sweeps.

usage: python tools/kbench.py [B] [steps] [rng]     (env: FRZ_WF_LANE_KERNEL, FRZ_WF_BLOCKS_PER_CU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import configs
from free_range_zoo_amd.envs import wildfire_v0

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rng = sys.argv[3] if len(sys.argv) > 3 else 'philox'
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'),
                               rng=rng, exact_shapes=False)
env.reset(seed=torch.arange(B, dtype=torch.int32))
times = []
for rep in range(3):
    env.reset(seed=torch.arange(B, dtype=torch.int32) + rep)
    pairs = []
    for t in range(steps):
        acts = env.random_policy_actions(policy_seed=1, policy_step=t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step(acts)
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    times = [a.elapsed_time(b) * 1e3 for a, b in pairs]
env.check()
print(f'B={B} rng={rng} lane={os.environ.get("FRZ_WF_LANE_KERNEL", "0")} per_cu={os.environ.get("FRZ_WF_BLOCKS_PER_CU", "auto")} '
      f'step kernel us: median={np.median(times):.1f} min={np.min(times):.1f} max={np.max(times):.1f} first5={[round(x,1) for x in times[:5]]}')
