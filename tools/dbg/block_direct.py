"""One K-step bench block (ONE launch: opening reset + K steps + metrics) launched as a one-node HIP graph vs directly through the C-ABI:
wall time per block, each bracketed by synchronize, completion detected by polling an event.  usage: python tools/dbg/block_direct.py [K]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import numpy as np, torch, configs
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd import _capi
from free_range_zoo_amd.utils.env import stream_ptr
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 65536
dev = torch.device('cuda')
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=dev, rng='philox', exact_shapes=False)
env.set_exclusive_device(True)
env.reset(seed=torch.arange(B, dtype=torch.int32))
metrics = torch.zeros(len(env.agents) + 2, dtype=torch.float64, device=dev)
graph = env.capture_random_rollout(K, policy_seed=1, include_reset=True, episode_length=50, seed_stride=1000003, metrics=metrics)
spec = _capi.frz_rollout_spec()
spec.n_steps, spec.rng_mode, spec.policy_seed, spec.flags = K, _capi.FRZ_RNG_PHILOX, 1, _capi.FRZ_ROLLOUT_RESET_FIRST
spec.seed_increment = 1000003
spec.actions_out, spec.metrics = env._actions.data_ptr(), metrics.data_ptr()
stream = stream_ptr(dev)
lib, handle, ref = env._lib, env._handle, ctypes.byref(spec)
ev = torch.cuda.Event()


def timed(fn, reps=300):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return np.median(out[20:]) * 1e6


def direct():
    lib.frz_wildfire_rollout(handle, ref, stream)


for name, fn in (('graph', graph.replay), ('direct', direct), ('graph', graph.replay), ('direct', direct)):
    print(f'K={K} {name}: {timed(fn):.1f} us per block', flush=True)
