"""Phase stamps of the step BEFORE THE LAST (an ordinary step: the last one draws nothing for a next step) of a multi-step launch (diagnostic build libfrz_hip_stamps.so), workgroup argv[1] (default 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['FRZ_HIP_LIB'] = os.path.join(ROOT, 'free-range-zoo_amd', 'csrc', 'libfrz_hip_stamps.so')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, configs
from free_range_zoo_amd import _capi
from free_range_zoo_amd.envs import wildfire_v0
from free_range_zoo_amd.utils.env import stream_ptr
B = 65536
WG = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # the workgroup whose stamps are kept
os.environ['FRZ_WF_SKIP'] = str(WG << 16)
env = wildfire_v0.parallel_env(configuration=configs.wildfire_openness(), parallel_envs=B, max_steps=50, device=torch.device('cuda'), rng='philox', exact_shapes=False)
env.reset(seed=torch.arange(B, dtype=torch.int32)); env.set_exclusive_device(True)
lib, h, s = env._lib, env._handle, stream_ptr(env.device)
names = ['entry', 'config staged', 'epoch/totals read', 'phase 1 done', 'past barrier 1', 'phase 2 done', 'past barrier 2', 'phase 3 done', 'phase 4 done',
         'past barrier 5', 'phase 6 done', '(11)', '(12)', 'crew: sums published', 'crew: rewards done', 'crew: counts stored']
rows = []
for rep in range(8):
    lib.frz_wildfire_reset(h, s)
    lib.frz_wildfire_rollout_random_policy(h, 1, 0, 30, env._actions.data_ptr(), _capi.FRZ_RNG_PHILOX, s)
    torch.cuda.synchronize()
    off = env._bufs.mt_state - env._arena.data_ptr() - ((5 * B * 3 * 4 + 255) // 256) * 256
    rows.append(env._arena[off:off + 32 * 8].view(torch.int64).cpu().numpy().astype(np.int64).copy())
st = np.array(rows[2:])
base = st[:, 3:4]  # field: phase 1 done of the last step
print(f'workgroup {WG}: cycles relative to the field role finishing phase 1 of the stamped step (stamps 0-2 are from the launch start)')
for i, n in enumerate(names):
    print(f'  {n:20s} field {int(np.median(st[:, i] - base[:, 0])):8d}   crew {int(np.median(st[:, 16 + i] - base[:, 0])):8d}')
