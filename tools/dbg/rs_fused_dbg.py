import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd.envs import rideshare_v0
def check(env, what, t0):
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flags = int(env._error_flags.item()); env._error_flags.zero_()
    off = env._task_offsets; counts = env._passenger_count.long()
    ok = bool(torch.equal(off[1:] - off[:-1], counts))
    bad = int((off[1:] - off[:-1] != counts).nonzero()[0]) if not ok else -1
    print(f'   {what}: {1e3*dt:.1f} ms flags={flags} offsets_ok={ok} first_bad_env={bad}', flush=True)
for A, B in ((3, 65536), (8, 140000), (3, 300000), (8, 65536)):
    env = rideshare_v0.parallel_env(configuration=configs.rideshare_busy(A=A, steps=6, per_step=1, seed=4), parallel_envs=B, max_steps=8, device=torch.device('cuda'), exact_shapes=False)
    print(f'A={A} B={B}')
    for i in range(3):
        t0 = time.perf_counter(); env.reset(seed=torch.arange(B, dtype=torch.int32)); check(env, f'reset {i}', t0)
    for i in range(3):
        t0 = time.perf_counter(); env.update_observations(); check(env, f'rebuild {i}', t0)
    for t in range(4):
        t0 = time.perf_counter(); env.step_random_policy(1, t); check(env, f'step {t}', t0)
