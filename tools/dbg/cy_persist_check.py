"""Cybersecurity multi-step launch against single-step launches: state/outputs equality and timing."""
import os, sys
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import cybersecurity_v0
from free_range_zoo_amd.utils.env import stream_ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
MAXS = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device('cuda', 0)
def make():
    e = cybersecurity_v0.parallel_env(configuration=configs.cyber_openness(), parallel_envs=B, max_steps=MAXS, device=dev, rng='philox', exact_shapes=False)
    e.reset(seed=torch.arange(B, dtype=torch.int32))
    return e
one, many = make(), make()
many.set_exclusive_device(True)
print('launches', many._lib.frz_cybersecurity_rollout_launches(many._handle, N, _capi.FRZ_RNG_PHILOX))
for t in range(N):
    one.step_random_policy(7, t)
many.rollout_random_policy(N, policy_seed=7, first_step=0)
torch.cuda.synchronize()
one.check(); many.check()
bad = []
names = [n for n in vars(one) if n.startswith('_') and isinstance(getattr(one, n), torch.Tensor) and getattr(one, n).is_cuda and n not in ('_arena', '_act_map_values')]
names += ['num_moves', 'agent_task_count', 'environment_task_count']
for name in names:
    a, b = getattr(one, name), getattr(many, name)
    if a.shape == b.shape and not torch.equal(a, b):
        bad.append(name)
for a in range(len(one.agents)):
    n = int(one._act_map_offsets[a, -1])
    if not torch.equal(one._act_map_values[a, :n], many._act_map_values[a, :n]):
        bad.append(f'act values {a}')
print('B', B, 'steps', N, 'max_steps', MAXS, 'MISMATCH ' + str(bad) if bad else 'identical', 'moves', int(one.num_moves.max()), 'checked', len(names))
lib, h, s, acts = many._lib, many._handle, stream_ptr(dev), many._actions.data_ptr()
for label, f in (('multi-step launch', lambda: lib.frz_cybersecurity_rollout_random_policy(h, 7, 0, N, acts, _capi.FRZ_RNG_PHILOX, s)),
                 ('one launch per step', lambda: [lib.frz_cybersecurity_step_random_policy(h, 7, t, acts, _capi.FRZ_RNG_PHILOX, None, None, s) for t in range(N)])):
    best = 1e9
    for rep in range(5):
        lib.frz_cybersecurity_reset(h, s)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / N * 1e3)
    print(f'  {label}: {best:.2f} us per step')
