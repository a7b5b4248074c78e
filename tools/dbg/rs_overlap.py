"""Experiment: two independent rideshare envs of B/2 stepping on two streams, the second delayed by about half a step so that one env's
emit launch overlaps the other's env launch, against one env of B."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd.envs import rideshare_v0

N = 50
def make(B):
    return rideshare_v0.parallel_env(configuration=configs.rideshare_busy(), parallel_envs=B, max_steps=N, device=torch.device('cuda'), exact_shapes=False)

def run(envs, streams, delay_cycles):
    for e in envs:
        e.reset(seed=torch.arange(e.parallel_envs, dtype=torch.int32))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, s in enumerate(streams):
        if i and delay_cycles:
            with torch.cuda.stream(s):
                torch.cuda._sleep(int(delay_cycles * i))
    for t in range(N):
        for e, s in zip(envs, streams):
            with torch.cuda.stream(s):
                e.step_random_policy(1, t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6

one = make(65536)
s0 = torch.cuda.Stream()
for _ in range(3):
    a = run([one], [s0], 0)
print(f'one env of 65536: {a:.1f} us per step')
del one
envs = [make(32768) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for delay_us in (0, 20, 40, 60, 80):
    for _ in range(3):
        b = run(envs, streams, delay_us * 2100)  # _sleep spins for that many shader clocks (~2.1 GHz)
    print(f'2 envs of 32768 on 2 streams, second delayed {delay_us} us: {b:.1f} us per step of all')
