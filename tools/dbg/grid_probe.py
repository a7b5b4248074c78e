"""Random-policy episodes of the wildfire grid family at B = 65536 for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/dbg/grid_probe.py [8x8x12 16x16x6 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
B = int(os.environ.get('FRZ_PROBE_B', 65536))
for shape in sys.argv[1:] or ['8x8x12', '16x16x6']:
    H, W, A = (int(v) for v in shape.split('x'))
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_grid(H, W, A), parallel_envs=B, max_steps=50, device=torch.device('cuda'),
                                   rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    lib, h, s = env._lib, env._handle, stream_ptr(env.device)
    times = []
    for rep in range(3):
        lib.frz_wildfire_reset(h, s)
        torch.cuda.synchronize(); torch.cuda._sleep(int(2.0e9 * 0.02))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(50):
            lib.frz_wildfire_step_random_policy(h, 1, t, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, None, None, s)
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / 50)
    env.check()
    print(f'{shape}: B={B} us per step (3 episodes) {[round(x, 1) for x in times]}', flush=True)
    del env
