"""Experiment: one K-step bench block as a HIP graph replay against the same launches enqueued eagerly by three C calls per episode."""
import os, sys, time
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
env.reset(seed=torch.arange(B, dtype=torch.int32))
env.set_exclusive_device(True)
metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device=dev)
lib, h, s, acts = env._lib, env._handle, stream_ptr(dev), env._actions.data_ptr()
done = torch.cuda.Event()
def wait():
    done.record()
    while not done.query():
        pass
    torch.cuda.synchronize()
for K in (20, 50, 100, 500):
    g = env.capture_random_rollout(K, policy_seed=1, include_reset=True, episode_length=EP, seed_stride=1000003, metrics=metrics)
    def graph_block():
        g.replay(); wait()
    def eager_block():
        left = K
        while left > 0:
            n = min(EP, left)
            lib.frz_wildfire_reset_reseed(h, 1000003, s)
            lib.frz_wildfire_rollout_random_policy_metrics(h, 1, 0, n, acts, _capi.FRZ_RNG_PHILOX, metrics.data_ptr(), s)
            left -= n
        wait()
    for name, f in (('graph', graph_block), ('eager', eager_block)):
        for _ in range(5): f()
        ts = []
        for _ in range(40):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        print(f'K={K:4d} {name}: median block {np.median(ts) * 1e6:8.1f} us = {np.median(ts) / K * 1e6:6.2f} us per step', flush=True)
