"""Host-side cost of the drop-in Python API per step (cProfile + wall clock); usage: python tools/api_profile.py [B]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd.envs import wildfire_v0

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox',
                               exact_shapes=False)
seed = torch.arange(B, dtype=torch.int32)


def episode(n=50):
    env.reset(seed=seed)
    for t in range(n):
        env.step_random_policy(policy_seed=1, policy_step=t)


def timed(label, fn, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    print(f'{label:40s} host {1e6 * host / reps:9.1f} us/call   host+device {1e6 * total / reps:9.1f} us/call', flush=True)


episode(); episode()
timed('reset', lambda: env.reset(seed=seed), 20)
timed('reset(skip_seeding)', lambda: env.reset(options={'skip_seeding': True}), 20)
env.reset(seed=seed)
timed('step_random_policy', lambda: env.step_random_policy(policy_seed=1, policy_step=3), 500)
acts = env.last_actions.clone() if hasattr(env, 'last_actions') else env._actions.clone()
timed('step(stacked actions)', lambda: env.step(acts), 500)
d = {a: acts[i] for i, a in enumerate(env.agents)}
timed('step(dict actions)', lambda: env.step(d), 500)
timed('episode (reset + 50 steps)', episode, 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    episode()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
