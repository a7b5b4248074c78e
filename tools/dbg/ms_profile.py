"""Cost of the steps of an episode inside the multi-step launch: launch duration for n = 1..50 steps after a reset, differenced."""
import os, sys, ctypes
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
B = 65536
dev = torch.device('cuda')
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=dev, rng='philox', exact_shapes=False)
env.reset(seed=torch.arange(B, dtype=torch.int32)); env.set_exclusive_device(True)
lib, h, s, acts = env._lib, env._handle, stream_ptr(dev), env._actions.data_ptr()
ns = [2, 5, 10, 15, 20, 30, 40, 50]
res = {}
for n in ns:
    best = []
    for rep in range(6):
        lib.frz_wildfire_reset(h, s); torch.cuda.synchronize()
        one = ctypes.c_float()
        _capi.check(lib.frz_wildfire_timed_rollout_launch(h, 1, 0, n, acts, _capi.FRZ_RNG_PHILOX, s, ctypes.byref(one)), 'timed')
        best.append(one.value * 1e3)
    res[n] = float(np.median(best))
prev_n, prev_t = 0, 0.0
for n in ns:
    print(f'n={n:3d}: launch {res[n]:7.1f} us; steps {prev_n:2d}..{n - 1:2d}: {(res[n] - prev_t) / (n - prev_n):6.2f} us per step')
    prev_n, prev_t = n, res[n]
