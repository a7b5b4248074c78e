"""Rideshare step latency along an episode: frz_rideshare_timed_rollout (HIP events taking the begin of each step's first dispatch and the
end of its last one), with the live passengers / visible tasks per env of the same steps.  FRZ_HIP_LIB selects the library build.
usage: python tools/rs_probe.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import rideshare_v0
from free_range_zoo_amd.utils.env import stream_ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = 50
env = rideshare_v0.parallel_env(configuration=configs.rideshare_busy(), parallel_envs=B, max_steps=N, device=torch.device('cuda'), exact_shapes=False)
lib, h, s = env._lib, env._handle, stream_ptr(env.device)
acts = env._actions.data_ptr()
out = (ctypes.c_float * N)()
for rep in range(3):
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    torch.cuda.synchronize()
    _capi.check(lib.frz_rideshare_timed_rollout(h, 1, 0, N, acts, s, out), 'frz_rideshare_timed_rollout')
ts = [out[i] * 1e3 for i in range(N)]
env.reset(seed=torch.arange(B, dtype=torch.int32))
counts, vis = [], []
for t in range(N):
    env.step_random_policy(1, t)
    counts.append(float(env.environment_task_count.float().mean()))
    vis.append(float(env.agent_task_count.float().sum(dim=0).mean()))
print(os.environ.get('FRZ_HIP_LIB', 'default'), 'B', B)
for t in (0, 5, 10, 20, 30, 40, 49):
    print(f'  step {t:2d}: passengers/env {counts[t]:5.1f}  visible (all agents) {vis[t]:6.1f}  step {ts[t]:7.1f} us')
print(f'  episode mean step {np.mean(ts):7.1f} us, sum {np.sum(ts) / 1e3:6.2f} ms; mean passengers {np.mean(counts):.1f}, mean visible {np.mean(vis):.1f}')
