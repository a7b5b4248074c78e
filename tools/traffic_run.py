"""Workload for the HBM-traffic counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass): one episode of a bench
workload at B = 65 536 through the C-ABI, uniform random policy sampled inside the step launch.
usage: python tools/traffic_run.py [wildfire|wildfire20|cybersecurity|rideshare|wildfire_grid_8x8|wildfire_grid_16x16]
(wildfire20: the 20-step launch of the driver's bench blocks; wildfire_grid_*: the grid-family workloads of bench.py's secondary_workloads)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd.envs import cybersecurity_v0, rideshare_v0, wildfire_v0
domain = sys.argv[1] if len(sys.argv) > 1 else 'wildfire'
steps = 50
GRIDS = {'wildfire_grid_8x8': (8, 8, 12), 'wildfire_grid_16x16': (16, 16, 6)}
if domain.startswith('wildfire') and domain != 'wildfire' and domain not in GRIDS:
    steps, domain = int(domain[len('wildfire'):]), 'wildfire'
module, build = {'wildfire': (wildfire_v0, configs.wildfire_openness), 'cybersecurity': (cybersecurity_v0, configs.cyber_openness),
                 'rideshare': (rideshare_v0, configs.rideshare_busy),
                 **{name: (wildfire_v0, (lambda shape: lambda: configs.wildfire_grid(*shape))(shape)) for name, shape in GRIDS.items()}}[domain]
B = 65536
env = module.parallel_env(configuration=build(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox', exact_shapes=False)
env.reset(seed=torch.arange(B, dtype=torch.int32))
if domain == 'wildfire':  # the launch of a bench block: opening reset, `steps` steps, episode metrics — ONE multi-step launch
    env.set_exclusive_device(True)
    metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device='cuda')
    env.rollout(steps, policy_seed=20260104, reset_first=True, seed_increment=1000003, metrics=metrics)
elif domain == 'cybersecurity':  # the episode's steps as the bench launches them: one multi-step launch
    env.set_exclusive_device(True)
    env.rollout_random_policy(50, policy_seed=20260104, first_step=0)
else:
    for t in range(50):
        env.step_random_policy(20260104, t)
torch.cuda.synchronize()
print('done', flush=True)
