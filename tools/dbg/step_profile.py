"""Per-step dispatch duration along a wildfire cfg2 episode (frz_wildfire_timed_rollout) and the reset / metrics launches."""
import os, sys, ctypes
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
B, EP = 65536, 50
dev = torch.device('cuda')
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=EP, device=dev, rng='philox', exact_shapes=False)
lib, h, s, acts = env._lib, env._handle, stream_ptr(dev), env._actions.data_ptr()
acc = np.zeros(EP)
for ep in range(6):
    env.reset(seed=torch.arange(B, dtype=torch.int32) + 1000003 * ep)
    torch.cuda.synchronize()
    out = (ctypes.c_float * EP)()
    _capi.check(lib.frz_wildfire_timed_rollout(h, 1, 0, EP, acts, _capi.FRZ_RNG_PHILOX, s, out), 'timed')
    if ep: acc += np.array(out[:]) * 1e3
acc /= 5
print('per-step dispatch us:', ' '.join(f'{v:.1f}' for v in acc))
print(f'mean of steps 0-19: {acc[:20].mean():.2f} us; mean of all 50: {acc.mean():.2f} us')
metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device=dev)
a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for _ in range(5):
    a.record(); lib.frz_wildfire_reset_reseed(h, 1000003, s); b.record(); lib.frz_wildfire_episode_metrics(h, metrics.data_ptr(), s); c.record()
    torch.cuda.synchronize()
print(f'reset_reseed {a.elapsed_time(b) * 1e3:.1f} us, episode metrics {b.elapsed_time(c) * 1e3:.1f} us (event to event, eager)')
