"""Where the eager per-step time goes (B = 65536, cfg2): n single-step launches enqueued by ONE C call (no Python between them) — the wall time
per step against the kernel's own duration = the HIP runtime's eager dispatch floor for dependent kernels; then the same through Python."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import numpy as np, torch, configs
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.env import stream_ptr
B, N = 65536, 50
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox', exact_shapes=False)
seed = torch.arange(B, dtype=torch.int32)
lib, h, s = env._lib, env._handle, stream_ptr(env.device)
out = (ctypes.c_float * N)()
for rep in range(3):
    env.reset(seed=seed); torch.cuda.synchronize()
    t0 = time.perf_counter()
    _capi.check(lib.frz_wildfire_timed_rollout(h, 1, 0, N, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, s, out), 'timed')
    wall = time.perf_counter() - t0
    print(f'C loop of {N} eager launches (with timing events): wall {1e6*wall/N:.2f} us/step, kernel mean {1e3*np.mean([out[i] for i in range(N)]):.2f} us', flush=True)
for rep in range(3):
    env.reset(seed=seed); torch.cuda.synchronize()
    t0 = time.perf_counter()
    _capi.check(lib.frz_wildfire_rollout_random_policy(h, 1, 0, N, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, s), 'rollout')
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    print(f'C loop of {N} eager launches (no events): host {1e6*host/N:.2f} us/step, wall {1e6*wall/N:.2f} us/step', flush=True)
