"""The reference-shaped loop with deferred steps (utils/env.py): host cost per step with the launches stubbed out, wall per step for several
chunk schedules.   usage: python tools/dbg/deferred_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import numpy as np, torch, configs
from free_range_zoo_amd.envs import wildfire_v0
torch.set_num_threads(8)
B, EPISODE = 65536, 50
dev = torch.device('cuda')
seeds = [torch.arange(B, dtype=torch.int32, device=dev) + 1000003 * e for e in range(8)]


def make(exclusive):
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=EPISODE, device=dev, rng='philox')
    if exclusive:
        assert env.set_exclusive_device(True)
    return env


def loop(env, episodes, check=True):
    for e in range(episodes):
        env.reset(seed=seeds[e])
        for _ in range(EPISODE):
            env.step({agent: env.action_space(agent).sample_nested() for agent in env.agents})
        if check:
            torch.all(env.finished)


def rate(env, label, reps=7, check=True):
    loop(env, 2, check)
    times = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loop(env, 8, check)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    print(f'{label}: {1e6 * np.median(times) / 400:.2f} us per step  ({B * 400 / np.median(times) / 1e9:.2f} G env-steps/s)', flush=True)


env = make(True)
rate(env, 'deferred, default schedule (2..16, doubling)')
for lo, hi in ((2, 8), (2, 32), (4, 16), (4, 32), (3, 24), (5, 50), (50, 50)):
    type(env)._DEFER_MIN, type(env)._DEFER_MAX = lo, hi
    rate(env, f'deferred, chunks {lo}..{hi}')
type(env)._DEFER_MIN, type(env)._DEFER_MAX = 2, 16
# host cost only: the launches stubbed out (the env then never moves: finished stays False, nothing reads it)
env._launch_deferred = lambda n, first, seed: None
rate(env, 'deferred, launches stubbed out, finished still read (host + reset + one read per episode)')
rate(env, 'deferred, launches stubbed out, no finished read', check=False)
t0 = time.perf_counter()
for _ in range(2000):
    a = {agent: env.action_space(agent).sample_nested() for agent in env.agents}
t1 = time.perf_counter()
print(f'dict of samples alone: {1e6 * (t1 - t0) / 2000:.2f} us', flush=True)
env.reset(seed=seeds[0])
t0 = time.perf_counter()
for _ in range(2000):
    env.step({agent: env.action_space(agent).sample_nested() for agent in env.agents})
t1 = time.perf_counter()
print(f'samples + step (stubbed launches): {1e6 * (t1 - t0) / 2000:.2f} us', flush=True)
del env
env = make(False)
rate(env, 'one launch per step')
