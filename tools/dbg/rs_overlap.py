"""Experiment: two independent rideshare envs of B/2 stepping on two streams (kernels of different kinds overlap) against one env of B."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd.envs import rideshare_v0

N = 50
def make(B):
    return rideshare_v0.parallel_env(configuration=configs.rideshare_busy(), parallel_envs=B, max_steps=N, device=torch.device('cuda'), exact_shapes=False)

def run(envs, streams, stagger):
    for e in envs:
        e.reset(seed=torch.arange(e.parallel_envs, dtype=torch.int32))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(N + stagger):
        for i, (e, s) in enumerate(zip(envs, streams)):
            tt = t - i * stagger
            if 0 <= tt < N:
                with torch.cuda.stream(s):
                    e.step_random_policy(1, tt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6

one = make(65536)
s0 = torch.cuda.Stream()
for _ in range(3):
    a = run([one], [s0], 0)
print(f'one env of 65536: {a:.1f} us per step')
del one
for parts in (2, 4):
    envs = [make(65536 // parts) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for _ in range(3):
        b = run(envs, streams, 0)
    print(f'{parts} envs of {65536 // parts} on {parts} streams: {b:.1f} us per step of all')
    del envs
