"""`env.rollout(...)`: n steps of the reference's rollout loop enqueued by one call through the C boundary (`frz_<domain>_rollout`,
`frz_rollout_spec` in include/frz.h) — one multi-step launch where the library has one for the shape, otherwise one launch per step,
with identical results.  Three uses on one MI355X:
  1. continuous random-policy rollouts at a fixed batch (auto-reset: an env that finishes restarts inside the step) with episode metrics;
  2. a recorded rollout (`record=True`: every step's rewards, flags and sampled actions) replayed from its action tape;
  3. the same call for rideshare (one launch sequence per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import time
import torch
import configs
from free_range_zoo_amd.envs import rideshare_v0, wildfire_v0

device = torch.device('cuda')
B = 65536


def granted_cores():
    """cores this process may really use (scheduler affinity cut down to the cgroup's CPU quota): torch's intra-op pool is capped at it below — on a
    box that shows 256 host threads and grants 16 cores' worth of time, a CPU tensor op of B elements otherwise wakes 128 spinning workers, the quota of
    the 100 ms period is gone in ~12 ms and every thread of the process — the one enqueueing launches included — is frozen for the rest"""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cores


torch.set_num_threads(min(torch.get_num_threads(), granted_cores()))

# 1. continuous rollouts: every env always mid-episode, returns of the episodes that ended accumulated on the device
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=device, rng='philox')
env.reset(seed=torch.arange(B, dtype=torch.int32))
env.set_exclusive_device(True)  # nothing else on this GPU: a rollout of the small exact shapes is ONE launch
metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device=device)
env.rollout(200, policy_seed=1, auto_reset=True, seed_stride=1000003, metrics=metrics)  # warm
metrics.zero_()
torch.cuda.synchronize()
t0 = time.perf_counter()
for block in range(10):
    env.rollout(200, policy_seed=1, first_step=200 * (block + 1), auto_reset=True, seed_stride=1000003, metrics=metrics)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
returns, env_steps, episodes = metrics[:-2], float(metrics[-2]), float(metrics[-1])
print(f'auto-reset rollouts: {env_steps / wall / 1e9:.2f} G env-steps/s, {episodes:.0f} episodes ended, mean return per agent '
      f'{(returns / max(episodes, 1)).tolist()}')

# 2. a recorded rollout and its replay from the action tape
env.reset(seed=torch.arange(B, dtype=torch.int32))
first = env.rollout(30, policy_seed=7, record=True)
env.reset(seed=torch.arange(B, dtype=torch.int32))
again = env.rollout(30, actions=first['actions'], record=True)
assert torch.equal(first['rewards'], again['rewards']) and torch.equal(first['dones'], again['dones'])
print(f"recorded rollout: rewards {tuple(first['rewards'].shape)}, dones {tuple(first['dones'].shape)}, actions {tuple(first['actions'].shape)}; "
      f"the replay from the tape gives the same rewards and flags")
env.check()

# 2b. ... and a trajectory a learner can consume: every step's observations (compact tape: the suppressant column, expanded on demand) and
# state beside the rewards / dones / actions — still ONE launch
env.reset(seed=torch.arange(B, dtype=torch.int32))
traj = env.rollout(30, policy_seed=7, record=True, record_observations='compact', record_state=True)
step12 = env.recorded_observations(traj, 12)  # {agent: TensorDict(self [B, 4], others [B, A - 1, k], tasks jagged [B, j, 4])} of step 12
agent = env.agents[0]
print(f"recorded trajectory: step 12 of {agent}: self {tuple(step12[agent]['self'].shape)}, others {tuple(step12[agent]['others'].shape)}, "
      f"{int(step12[agent]['tasks'].values().shape[0])} task rows in the batch; lit fires in the state tape: "
      f"{int((env.recorded_state(traj, 12).fires > 0).sum())}")

# 2c. the reference's own loop, unchanged, on a GPU this process owns: the steps are counted and run in multi-step launches
for timed in (False, True):  # (one untimed pass first: the action-space objects and the step chunks are built on first use)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for episode in range(4):
        env.reset(seed=torch.arange(B, dtype=torch.int32, device=device) + episode)
        while not torch.all(env.finished):
            for _ in range(50):
                env.step({name: env.action_space(name).sample_nested() for name in env.agents})
    torch.cuda.synchronize()
    if timed:
        print(f'reference-shaped loop on an exclusive device: {4 * 50 * B / (time.perf_counter() - t0) / 1e9:.2f} G env-steps/s')

# 3. rideshare: the same spec, one launch sequence per step
ride = rideshare_v0.parallel_env(configuration=configs.rideshare_busy(), parallel_envs=4096, max_steps=32, device=device)
ride.reset(seed=torch.arange(4096, dtype=torch.int32))
out = ride.rollout(32, policy_seed=3, reset_first=True, record=True)
print(f"rideshare rollout: summed reward {float(out['rewards'].sum()):.1f}, truncated at the end: {bool(out['dones'][-1, 1].all())}")
ride.check()
