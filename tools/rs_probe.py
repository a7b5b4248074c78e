"""Rideshare step latency along an episode (HIP events on the launch stream); FRZ_HIP_LIB selects the library build.
usage: python tools/rs_probe.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd.envs import rideshare_v0
from free_range_zoo_amd.utils.env import stream_ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = rideshare_v0.parallel_env(configuration=configs.rideshare_busy(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), exact_shapes=False)
lib, h, s = env._lib, env._handle, stream_ptr(env.device)
acts = env._actions.data_ptr()
for rep in range(2):
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    torch.cuda.synchronize(); torch.cuda._sleep(int(2.0e9 * 0.02))
    ev, counts = [], []
    for t in range(45):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); lib.frz_rideshare_random_policy(h, 1, t, acts, s)
        e[1].record(); lib.frz_rideshare_step(h, acts, s)
        e[2].record(); ev.append(e)
        counts.append(env.environment_task_count.float().mean())
    torch.cuda.synchronize()
ts = [a[1].elapsed_time(a[2]) * 1e3 for a in ev]
ps = [a[0].elapsed_time(a[1]) * 1e3 for a in ev]
print(os.environ.get('FRZ_HIP_LIB', 'default'), 'B', B)
for t in (0, 5, 10, 20, 30, 40, 44):
    print(f'  step {t:2d}: passengers/env {float(counts[t]):5.1f}  step {ts[t]:7.1f} us  policy {ps[t]:6.1f} us')
print(f'  episode mean step {np.mean(ts):7.1f} us, sum {np.sum(ts) / 1e3:6.2f} ms')
