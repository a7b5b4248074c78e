"""Workload for the HBM-traffic counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass):
one episode of the bench workload (wildfire cfg2, B=65536, uniform random policy inside the step launch) through the C-ABI."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
B = 65536
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'),
                               rng='philox', exact_shapes=False)
env.reset(seed=torch.arange(B, dtype=torch.int32))
lib, h, s = env._lib, env._handle, stream_ptr(env.device)
for t in range(50):
    lib.frz_wildfire_step_random_policy(h, 20260104, t, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, None, None, s)
torch.cuda.synchronize()
print('done', flush=True)
