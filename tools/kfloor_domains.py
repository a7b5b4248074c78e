"""Latency probe for the cybersecurity / rideshare step launches through the C-ABI only (HIP events on the launch stream, host
running ahead of a busy device), like tools/kfloor.py for wildfire.  usage: python tools/kfloor_domains.py [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0
from free_range_zoo_amd.utils.env import stream_ptr

for B in [int(x) for x in sys.argv[1:]] or [65536]:
    for name, mod, build in (('cybersecurity', cybersecurity_v0, configs.cyber_openness), ('rideshare', rideshare_v0, configs.rideshare_busy)):
        env = mod.parallel_env(configuration=build(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox', exact_shapes=False)
        env.reset(seed=torch.arange(B, dtype=torch.int32))
        lib, h, s = env._lib, env._handle, stream_ptr(env.device)
        acts = env._actions.data_ptr()
        policy = getattr(lib, f'frz_{name}_random_policy')
        step = getattr(lib, f'frz_{name}_step')
        ts = []
        for rep in range(2):
            env.reset(seed=torch.arange(B, dtype=torch.int32) + rep)
            torch.cuda.synchronize(); torch.cuda._sleep(int(2.0e9 * 0.02))
            ev = []
            for t in range(45):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); policy(h, 1, t, acts, s)
                e[1].record()
                if name == 'cybersecurity':
                    step(h, acts, _capi.FRZ_RNG_PHILOX, None, None, s)
                else:
                    step(h, acts, s)
                e[2].record(); ev.append(e)
            torch.cuda.synchronize()
            ts = [a[1].elapsed_time(a[2]) * 1e3 for a in ev]
        env.check()
        tp = [a[0].elapsed_time(a[1]) * 1e3 for a in ev]
        line = f'{name:14s} B={B:7d} step us: first={ts[0]:6.1f} median={np.median(ts):6.1f} last={ts[-1]:6.1f} min={np.min(ts):6.1f}  policy us median={np.median(tp):5.1f}'
        if name == 'cybersecurity':  # the step with the policy sampled in the launch
            env.reset(seed=torch.arange(B, dtype=torch.int32))
            torch.cuda.synchronize(); torch.cuda._sleep(int(2.0e9 * 0.02))
            ev = []
            for t in range(45):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                e[0].record(); lib.frz_cybersecurity_step_random_policy(h, 1, t, acts, _capi.FRZ_RNG_PHILOX, None, None, s)
                e[1].record(); ev.append(e)
            torch.cuda.synchronize()
            tf = [a[0].elapsed_time(a[1]) * 1e3 for a in ev]
            line += f'  fused policy+step us median={np.median(tf):6.1f}'
        print(line, flush=True)
        del env
