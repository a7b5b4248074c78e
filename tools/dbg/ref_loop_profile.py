"""Host cost of the reference-shaped rollout loop (per agent action_space(agent).sample_nested(), env.step(dict), torch.all(env.finished) once per
episode) at B = 65536: wall per step + cProfile.  usage: python tools/dbg/ref_loop_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch, configs
from free_range_zoo_amd.envs import wildfire_v0
B = 65536
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox')
seed = torch.arange(B, dtype=torch.int32).cuda()


def episode():
    env.reset(seed=seed)
    steps = 0
    while True:
        for _ in range(50):
            env.step({agent: env.action_space(agent).sample_nested() for agent in env.agents})
        steps += 50
        if torch.all(env.finished):
            return steps


episode()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = sum(episode() for _ in range(10))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'reference-shaped loop: {1e6 * dt / n:.2f} us per step, {B * n / dt / 1e9:.2f} G env-steps/s', flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    episode()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
