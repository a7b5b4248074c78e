import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch, configs
from free_range_zoo_amd.envs import wildfire_v0
B = 65536
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox')
seed_cpu = torch.arange(B, dtype=torch.int32); seed_dev = seed_cpu.cuda()
env.reset(seed=seed_cpu)
def timed(label, fn, reps=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    host = time.perf_counter() - t0; torch.cuda.synchronize(); total = time.perf_counter() - t0
    print(f'{label:44s} host {1e6*host/reps:8.1f} us  host+device {1e6*total/reps:8.1f} us', flush=True)
timed('reset(seed=cpu tensor)', lambda: env.reset(seed=seed_cpu))
timed('reset(seed=device tensor)', lambda: env.reset(seed=seed_dev))
timed('reset(skip_seeding)', lambda: env.reset(options={'skip_seeding': True}))
timed('torch.all(env.finished) + bool()', lambda: bool(torch.all(env.finished)))
timed('env.finished', lambda: env.finished)
def steps():
    for _ in range(50):
        env.step({agent: env.action_space(agent).sample_nested() for agent in env.agents})
env.reset(seed=seed_dev)
timed('50 reference-shaped steps', steps, 20)
