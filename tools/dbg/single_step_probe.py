import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
B = 65536
for rng, mode in (('philox', _capi.FRZ_RNG_PHILOX),):
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng=rng, exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    lib, h, s = env._lib, env._handle, stream_ptr(env.device)
    ts = []
    for rep in range(5):
        lib.frz_wildfire_reset(h, s); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(50):
            lib.frz_wildfire_step_random_policy(h, 1, t, env._actions.data_ptr(), mode, None, None, s)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 50)
    print(os.environ.get('FRZ_HIP_LIB', 'default')[-12:], rng, 'single-step launches, us per step:', [round(x, 2) for x in ts[1:]], flush=True)
