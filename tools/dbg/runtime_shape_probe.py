"""Runtime-shape wildfire grids (no exact kernel instantiation: the <8, 4> / <16, 4> field/crew variants): a 50-step random-policy rollout as ONE
multi-step launch (exclusive device) against one launch sequence per step, B = 65536.  usage: python tools/dbg/runtime_shape_probe.py [HxWxA ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
B, N = 65536, 50
RNG = os.environ.get('FRZ_PROBE_RNG', 'philox')  # or mt19937
MODE = _capi.FRZ_RNG_PHILOX if RNG == 'philox' else _capi.FRZ_RNG_MT19937
for shape in sys.argv[1:] or ['1x7x3', '3x5x4', '2x5x2']:
    H, W, A = (int(v) for v in shape.split('x'))
    out = {}
    for exclusive in (False, True):
        env = wildfire_v0.parallel_env(configuration=configs.wildfire_grid(H, W, A), parallel_envs=B, max_steps=N, device=torch.device('cuda'), rng=RNG,
                                       exact_shapes=False)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        if exclusive:
            env.set_exclusive_device(True)
        launches = env._lib.frz_wildfire_rollout_launches(env._handle, N, MODE)
        ts = []
        for rep in range(4):
            env.reset(seed=torch.arange(B, dtype=torch.int32) + rep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            env.rollout(N, policy_seed=3)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6 / N)
        env.check()
        out[exclusive] = (launches, min(ts[1:]))
        del env
    print(f'{shape} {RNG}: one launch per step ({out[False][0]} launches) {out[False][1]:.1f} us per step; multi-step ({out[True][0]} launch) {out[True][1]:.1f} us per step', flush=True)
