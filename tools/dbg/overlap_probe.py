"""Do the grid family's env and lists launches overlap usefully?  Two independent 8x8 / 12-agent envs of B = 65536 stepped (a) one after
the other on one stream, (b) each on its own stream: (b) lets one env's lists launch run beside the other's env launch.
python tools/dbg/overlap_probe.py [HxWxA]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
H, W, A = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '8x8x12').split('x'))
B = 65536
dev = torch.device('cuda')
envs = []
for i in range(2):
    env = wildfire_v0.parallel_env(configuration=configs.wildfire_grid(H, W, A), parallel_envs=B, max_steps=50, device=dev, rng='philox', exact_shapes=False)
    env.reset(seed=torch.arange(B, dtype=torch.int32) + 7 * i)
    envs.append(env)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
lib = envs[0]._lib


def episode(stream_of):
    for env in envs:
        lib.frz_wildfire_reset(env._handle, stream_of(env).cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(50):
        for env in envs:
            lib.frz_wildfire_step_random_policy(env._handle, 1, t, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, None, None, stream_of(env).cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e6


for rep in range(3):
    same = episode(lambda env: streams[0])
    two = episode(lambda env: streams[envs.index(env)])
    print(f'{H}x{W}x{A}: one stream {same:.1f} us per step of both envs ({same / 2:.1f} each), two streams {two:.1f} ({two / 2:.1f} each)', flush=True)
