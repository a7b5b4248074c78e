"""A/B on ONE box: duration of the wildfire multi-step launch (n steps, B = 65536, cfg2, Philox) of the library in the tree given as argv[1]
(events around the dispatch).  usage: python tools/dbg/ab_persist.py <repo root> [n ...]"""
import ctypes, os, sys
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import numpy as np, torch, configs
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.env import stream_ptr
B = 65536
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox', exact_shapes=False)
env.set_exclusive_device(True)
seeds = torch.arange(B, dtype=torch.int32)
stream = stream_ptr(env.device)
for n in [int(a) for a in sys.argv[2:]] or [20, 50]:
    ms = []
    for rep in range(14):
        env.reset(seed=seeds)
        torch.cuda.synchronize()
        one = ctypes.c_float()
        _capi.check(env._lib.frz_wildfire_timed_rollout_launch(env._handle, 7, 0, n, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, stream, ctypes.byref(one)), 'timed')
        ms.append(one.value)
    ms = ms[2:]
    print(f'{root}: n={n} launch {np.mean(ms)*1e3:.1f} us (min {np.min(ms)*1e3:.1f}) = {np.mean(ms)*1e3/n:.2f} us/step', flush=True)
