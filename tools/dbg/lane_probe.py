"""Lane-per-env wildfire family (grids of 17-24 cells, or more than 4 agents on more than 8 cells...): us per step of a 50-step random-policy episode, one launch
per step, B = 65536, with the lane-per-env kernel and with the field/crew kernel (FRZ_WF_KERNEL).  FRZ_HIP_LIB selects the library build.  usage: python tools/dbg/lane_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
B = 65536
for shape, family in [(sh, fam) for sh in (sys.argv[1:] or ['4x5x4', '4x6x8', '3x3x6']) for fam in ('lane', 'roles')]:
    os.environ['FRZ_WF_KERNEL'] = family  # (read when the env object is created)
    H, W, A = (int(v) for v in shape.split('x'))
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_grid(H, W, A), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    lib, h, s = env._lib, env._handle, stream_ptr(env.device)
    ts = []
    for rep in range(3):
        lib.frz_wildfire_reset(h, s); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(50):
            lib.frz_wildfire_step_random_policy(h, 1, t, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, None, None, s)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 50)
    env.check()
    print(os.environ.get('FRZ_HIP_LIB', 'default')[-12:], shape, family, [round(x, 1) for x in ts], flush=True)
