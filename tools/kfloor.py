"""Latency floor probe: fused wildfire step launch timed through the C-ABI only (no Python work between the events)."""
import os, sys
if os.environ.get('FRZ_WF_SKIP'):  # timing experiments need the diagnostic build
    os.environ.setdefault('FRZ_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'free-range-zoo_amd', 'csrc', 'libfrz_hip_stamps.so'))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
for B in [int(x) for x in sys.argv[1:]] or [256, 4096, 16384, 65536, 131072, 262144]:
    build = {"openness": configs.wildfire_openness, "rich": configs.wildfire_rich, "grid8x8": lambda: configs.wildfire_grid(8, 8, 12), "grid3x3": lambda: configs.wildfire_grid(3, 3, 3), "grid4x4": lambda: configs.wildfire_grid(4, 4, 6), "grid4x4a4": lambda: configs.wildfire_grid(4, 4, 4), "grid3x4a4": lambda: configs.wildfire_grid(3, 4, 4), "grid5x5": lambda: configs.wildfire_grid(5, 5, 3), "grid16x16": lambda: configs.wildfire_grid(16, 16, 6), "grid32x32": lambda: configs.wildfire_grid(32, 32, 12), "grid4x5": lambda: configs.wildfire_grid(4, 5, 4)}[os.environ.get('FRZ_KFLOOR_CONFIG', 'openness')]
    env = wildfire_v0.parallel_env(configuration=build(), parallel_envs=B, max_steps=50, device=torch.device('cuda'),
                                   rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32))
    lib, h, s = env._lib, env._handle, stream_ptr(env.device)
    ts, tp = [], []
    for rep in range(3):
        lib.frz_wildfire_reset(h, s)
        torch.cuda.synchronize(); torch.cuda._sleep(int(2.0e9 * 0.02))  # let the host run ahead: device-side timing
        ev = []
        for t in range(45):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); lib.frz_wildfire_random_policy(h, 1, t, env._actions.data_ptr(), s)
            e[1].record(); lib.frz_wildfire_step(h, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, None, None, s)
            e[2].record(); ev.append(e)
        torch.cuda.synchronize()
        ts = [a[1].elapsed_time(a[2]) * 1e3 for a in ev]; tp = [a[0].elapsed_time(a[1]) * 1e3 for a in ev]
    tf = []
    for rep in range(3):
        lib.frz_wildfire_reset(h, s)
        torch.cuda.synchronize(); torch.cuda._sleep(int(2.0e9 * 0.02))
        ev = []
        for t in range(45):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record(); lib.frz_wildfire_step_random_policy(h, 1, t, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, None, None, s)
            e[1].record(); ev.append(e)
        torch.cuda.synchronize()
        tf = [a[0].elapsed_time(a[1]) * 1e3 for a in ev]
    print(f'B={B:7d} step us median={np.median(ts):6.1f} min={np.min(ts):6.1f}  policy us median={np.median(tp):5.1f}  '
          f'fused policy+step us median={np.median(tf):6.1f} min={np.min(tf):6.1f}', flush=True)
    del env
